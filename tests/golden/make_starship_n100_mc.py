"""Oracle records of MONTE-CARLO instances of BASELINE.json configs[2] at its stated size (VERDICT r05 missing 4 / next 1b): Starship
landing flip, SCvx, N = 100, Nsub = 100, reference test parameters and stopping rule (test/examples/starship_flip/tests.jl:77-98,
definition.jl:395-412), initial conditions +-2 % with seed = instance (bench.starship_scvx_record), every instance from ITS OWN oracle
guess (oracle/starship_guess.py: 20 s and 21 s first feasible durations both occur), 30 iterations of the oracle's literal loop
(oracle/scvx_ref.py + oracle/ipm.py).  The device loop is compared with these decision by decision (tests/test_starship_gpu.py).

    python tests/golden/make_starship_n100_mc.py [iterations = 30] [instances = 1,2,3,4,5,7,8,64] [append]     # ~5-10 min per instance, 4 at a time
"""
import multiprocessing as mp
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)

N, NSUB = 100, 100


def run(args):
    i, iters = args
    from make_starship_golden import oracle_batch
    from oracle import scvx_ref
    from oracle.models import MODELS
    from oracle.starship_guess import StarshipConstants, starship_initial_guess
    nom = MODELS["starship"](N).nominal_pp()
    pp = nom * (1 + (0.02 * np.random.default_rng(i).uniform(-1, 1, nom.size) if i else 0.0))

    class K(StarshipConstants):
        pass
    K.r0, K.v0, K.theta0 = pp[0:2], pp[2:4], float(pp[4])
    x, u, p, hs = starship_initial_guess(N, oracle_batch, K)
    hs0 = float(np.load(os.path.join(HERE, "starship_guess_mc.npz"))["hs100"])      # the batch's cost normalisation: the nominal switch altitude
    mdl = MODELS["starship"](N, hs0)
    sp_ = scvx_ref.SCvxParameters(N, NSUB, iters, lam=5e2, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                  eta_lb=1e-8, eta_ub=10.0, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3)
    st, h = scvx_ref.scvx_solve(mdl, sp_, pp=pp, guess=(x, u, p), verbose=False, ipm_opts=dict(max_iter=1000))
    K_ = len(h)
    pad = lambda a, fill: np.concatenate([np.asarray(a, float), np.full(iters - K_, fill)])
    fin = h[-1]["sol"]
    rec = dict(instance=i, pp=pp, guess_x=x, guess_u=u, guess_p=p, guess_hs=hs, status=st, iters=K_,
               eta=pad([r["eta"] for r in h], np.nan), L=pad([r["sub"]["L"] for r in h], np.nan), L_aug=pad([r["sub"]["L_aug"] for r in h], np.nan),
               J_sol=pad([r.get("J_sol", np.nan) for r in h], np.nan), rho=pad([r.get("rho", np.nan) for r in h], np.nan),
               accept=pad([1.0 if r.get("accept", False) else 0.0 for r in h], -1), feas=pad([1.0 if r["sol"].feas else 0.0 for r in h], -1),
               ipm_ok=pad([1.0 if r["sub"]["status"] in ("OPTIMAL", "ALMOST_OPTIMAL") else 0.0 for r in h], -1),
               ipm_iters=pad([r["sub"]["ipm"]["iters"] for r in h], -1), xd=fin.xd, ud=fin.ud, p=fin.p, final_feas=bool(fin.feas))
    print("instance %d: guess t2 = %.0f s, %s after %d iterations, final feas %s, L %.6f" % (i, p[1], st, K_, fin.feas, h[-1]["sub"]["L"]), flush=True)
    return rec


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    inst = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4,5,7,8,64").split(",")]
    with mp.Pool(min(4, len(inst))) as pool:
        recs = pool.map(run, [(i, iters) for i in inst])
    out = dict(N=N, Nsub=NSUB, iters_max=iters, instances=np.array(inst))
    for k in recs[0]:
        if k == "instance":
            continue
        out[k] = np.stack([np.asarray(r[k]) for r in recs])
    path = os.path.join(HERE, "starship_N100_scvx_mc.npz")
    if len(sys.argv) > 3 and sys.argv[3] == "append" and os.path.exists(path):      # add instances to the committed record
        old = dict(np.load(path))
        assert int(old["iters_max"]) == iters
        for k in out:
            if k not in ("N", "Nsub", "iters_max"):
                out[k] = np.concatenate([old[k], out[k]])
    np.savez_compressed(path, **out)
    print("frac_dyn_feasible of the oracle after %d iterations: %.3f" % (iters, float(np.mean(out["final_feas"]))))


if __name__ == "__main__":
    main()
