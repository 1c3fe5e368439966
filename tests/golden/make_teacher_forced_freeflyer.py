"""TEACHER-FORCED subproblem goldens for the free-flyer (BASELINE.json configs[4]'s model on the reference's own grid: N = 50, Nsub = 15,
15 iterations, reference test parameters freeflyer/tests.jl:25-80 (SCvx) and :84-140 (GuSTO, pen = :quad with the cone indicators
of define_conic_constraint!)): for every instance x iteration of the oracle's literal loops on the first Monte-Carlo instances of
bench.py's free-flyer record (initial / terminal positions +-3 mm, seed = instance) the REFERENCE the oracle linearised about,
(eta, lambda) and the optimal value of the oracle's literal conic program -- same layout as make_teacher_forced.py.
tests/test_teacher_forced_gpu.py hands these references to the DEVICE subproblem and compares optimal values to 1e-6.

    OMP_NUM_THREADS=1 python tests/golden/make_teacher_forced_freeflyer.py [instances = 8] [processes = 8]
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
ITERS = 15
N, NSUB = 50, 15


def pp_of(mdl, i):
    rng = np.random.default_rng(i)
    q = mdl.nominal_pp().copy()
    q[0:3] += 0.003 * rng.uniform(-1, 1, 3); q[13:16] += 0.003 * rng.uniform(-1, 1, 3)        # bench.py, freeflyer_gusto_record.pps
    return q


def run(job):
    algo, b = job
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import gusto_ref, scvx_ref
    from oracle.models import MODELS
    mdl = MODELS["freeflyer"](N)
    pp = pp_of(mdl, b)
    if algo == "scvx":
        pars = scvx_ref.SCvxParameters(N, NSUB, ITERS, lam=1e3, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                       eta_lb=1e-6, eta_ub=10.0, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
        st, h = scvx_ref.scvx_solve(mdl, pars, pp=pp)
        recs = [dict(xd=r["ref"].xd, ud=r["ref"].ud, p=r["ref"].p, eta=r["eta"], lam=np.nan, pcost=r["sub"]["L_aug"], L=r["sub"]["L"],
                     ok=r["sub"]["status"] in ("OPTIMAL", "ALMOST_OPTIMAL"), sx=r["sub"]["x"], su=r["sub"]["u"], sp=r["sub"]["p"]) for r in h]
    else:
        gp = gusto_ref.GuSTOParameters(N, NSUB, ITERS, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.5, beta_sh=2.0, beta_gr=2.0,
                                       gamma_fail=5.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=16, eps_abs=0.0,
                                       eps_rel=0.0, feas_tol=1e-3)
        st, h = gusto_ref.gusto_solve(mdl, gp, pp=pp)
        recs = [dict(xd=r["ref"].xd, ud=r["ref"].ud, p=r["ref"].p, eta=r["eta"], lam=r["lam"], pcost=r["sub"]["pcost"], L=r["sub"]["L"],
                     ok=r["sub"]["status"] in ("OPTIMAL", "ALMOST_OPTIMAL"), sx=r["sub"]["x"], su=r["sub"]["u"], sp=r["sub"]["p"]) for r in h]
    return algo, b, pp, st, recs


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    jobs = [(a, b) for b in range(n) for a in ("gusto", "scvx")]
    with mp.Pool(procs) as pool:
        res = pool.map(run, jobs, chunksize=1)
    from oracle.models import MODELS
    mdl = MODELS["freeflyer"](N)
    for algo in ("scvx", "gusto"):
        rs = sorted([r for r in res if r[0] == algo], key=lambda r: r[1])
        npar = len(rs[0][4][0]["p"])
        xd = np.zeros((n, ITERS, N, mdl.nx)); ud = np.zeros((n, ITERS, N, mdl.nu)); p = np.zeros((n, ITERS, npar))
        sx = np.zeros_like(xd); su = np.zeros_like(ud); sp = np.zeros_like(p)
        eta = np.full((n, ITERS), np.nan); lam = np.full((n, ITERS), np.nan); pc = np.full((n, ITERS), np.nan)
        L = np.full((n, ITERS), np.nan); valid = np.zeros((n, ITERS), bool)
        for _, b, pp, st, recs in rs:
            for k, r in enumerate(recs):
                xd[b, k], ud[b, k], p[b, k] = r["xd"], r["ud"], r["p"]
                sx[b, k], su[b, k], sp[b, k] = r["sx"], r["su"], r["sp"]
                eta[b, k], lam[b, k], pc[b, k], L[b, k], valid[b, k] = r["eta"], r["lam"], r["pcost"], r["L"], r["ok"]
        np.savez_compressed(os.path.join(HERE, "teacher_forced_%s_freeflyer_N50.npz" % algo), pp=np.stack([r[2] for r in rs]),
                            ref_xd=xd, ref_ud=ud, ref_p=p, eta=eta, lam=lam, pcost=pc, L=L, valid=valid, sol_p=sp,
                            sol_xd=sx.astype(np.float32), sol_ud=su.astype(np.float32),
                            solved=np.array([r[3].split()[0] == "SCP_SOLVED" for r in rs]), N=N, Nsub=NSUB, iter_max=ITERS)
        print(algo, "subproblems", int(valid.sum()), "of", n * ITERS, "solved", [r[3] for r in rs])


if __name__ == "__main__":
    main()
