"""Generates the committed golden fixtures from the CPU oracle (the reference ships none).

    python tests/golden/make_golden.py

* discretize_<model>.npz : seeded inputs + oracle `discretize!` outputs
* propagate_<model>.npz  : continuous-time propagation (`propagate`) of the same seeded trajectories, res = 101
* ptr_<model>.npz        : per-iteration costs and the final trajectory of the oracle's literal PTR loop
                           (oracle/ptr_ref.py: conic program of src/solvers/ptr.jl solved by oracle/ipm.py)
* cfg_<model>_N<N>_<tag>.npz : the same loop AT THE CONFIG SIZES that are benchmarked (rocket landing N=100, quadrotor
                           N=50; nominal + Monte-Carlo instances x0*(1+0.1 xi), seed = tag): per-iteration history, the
                           final trajectory, and for three subproblems (first, mid-run, converged regime) the reference
                           point and the literal conic program's full solution incl. vd, vs, vic, vtc, P, Pf, eta
                           (`python tests/golden/make_golden.py cfg` regenerates only these; ~10 minutes)
The fixtures pin (a) the oracle against silent regressions (-m "not gpu") and (b) the HIP path (-m gpu).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc  # noqa: E402
from oracle import ptr_ref  # noqa: E402
from oracle.models import MODELS  # noqa: E402

DISC = [("double_integrator", 30, 10), ("quadrotor", 50, 15), ("rocket_landing", 20, 15)]
PTR = [("double_integrator", 30, 10, 6), ("quadrotor", 20, 10, 12), ("rocket_landing", 16, 10, 10)]


CFG = [("rocket_landing", 100, 15, 15, [-1, 0, 1, 2]), ("quadrotor", 50, 15, 15, [-1, 0])]
CFG_SUB_ITERS = (1, 4, 12)   # first / mid-run / converged-regime subproblems stored in full


def mc_pp(mdl, seed):
    """Monte-Carlo per-problem data (same rule as bench.py): nominal for seed < 0."""
    q = mdl.nominal_pp().copy()
    if seed < 0:
        return q
    rng = np.random.default_rng(seed)
    if mdl.name == "quadrotor":
        q[6:9] = q[6:9] * (1 + 0.1 * rng.uniform(-1, 1, 3))
    else:
        q = q * (1 + 0.1 * rng.uniform(-1, 1, q.size))
    return q


def make_cfg():
    for model, N, Nsub, iters, seeds in CFG:
        mdl = MODELS[model]()
        for seed in seeds:
            pp = mc_pp(mdl, seed)
            pars = ptr_ref.PTRParameters(N, Nsub, iters, 1e3, 0.1, 0, 0, 1e-3)
            st, hist = ptr_ref.ptr_solve(model, pars, pp=pp)
            fin = hist[-1]["sol"]
            out = dict(N=N, Nsub=Nsub, iters=iters, status=st, pp=pp, n_hist=len(hist),
                       J=[h["sub"]["J"] for h in hist], J_tr=[h["sub"]["J_tr"] for h in hist],
                       J_vc=[h["sub"]["J_vc"] for h in hist], J_aug=[h["sub"]["J_aug"] for h in hist],
                       feas=[h["sol"].feas for h in hist], ipm_status=[h["sub"]["status"] for h in hist],
                       ipm_iters=[h["sub"]["ipm"]["iters"] for h in hist],
                       xd=fin.xd, ud=fin.ud, p=fin.p, defect=fin.defect)
            for it in CFG_SUB_ITERS:
                if it > len(hist):
                    continue
                h = hist[it - 1]
                sub, ref = h["sub"], h["ref"]
                pre = "s%d_" % it
                out.update({pre + "ref_x": ref.xd, pre + "ref_u": ref.ud, pre + "ref_p": ref.p,
                            pre + "x": sub["x"], pre + "u": sub["u"], pre + "p": sub["p"], pre + "vd": sub["vd"],
                            pre + "vs": sub["vs"] if sub["vs"] is not None else np.zeros((N, 0)),
                            pre + "vic": sub["vic"], pre + "vtc": sub["vtc"], pre + "P": sub["P"], pre + "Pf": sub["Pf"],
                            pre + "etax": sub["etax"], pre + "etau": sub["etau"], pre + "etap": sub["etap"],
                            pre + "cost": np.array([sub["J"], sub["J_tr"], sub["J_vc"], sub["J_aug"]]),
                            pre + "status": sub["status"], pre + "gap": sub["ipm"]["gap"]})
            tag = "nom" if seed < 0 else "mc%d" % seed
            np.savez_compressed(os.path.join(HERE, "cfg_%s_N%d_%s.npz" % (model, N, tag)), **out)
            print(model, N, tag, st, hist[-1]["sub"]["J"], [h["sub"]["status"][:3] for h in hist], flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "cfg":
        make_cfg()
        return
    for model, N, Nsub in DISC:
        mdl = MODELS[model]()
        rng = np.random.default_rng(1234)
        B = 3
        xs, us, ps = [], [], []
        for b in range(B):
            pp = mdl.nominal_pp() * (1 + 0.1 * rng.uniform(-1, 1, mdl.nominal_pp().size))
            x, u, p = mdl.guess(N, pp)
            xs.append(x + 0.05 * rng.standard_normal(x.shape) * (1 + np.abs(x))); us.append(u + 0.05 * rng.standard_normal(u.shape))
            ps.append(p * (1 + 0.1 * rng.uniform(-1, 1, p.shape)))
        xd, ud, p = np.stack(xs), np.stack(us), np.stack(ps).reshape(B, -1)
        scale = ptr_ref.Scaling(*mdl.bbox())
        out = orc.discretize(model, mdl.par(), N, Nsub, xd, ud, p, 1.0 / scale.Sx, 1e-3)
        np.savez_compressed(os.path.join(HERE, "discretize_%s.npz" % model), N=N, Nsub=Nsub, xd=xd, ud=ud, p=p, iSx=1.0 / scale.Sx,
                            **{k: v for k, v in out.items()})
        res = 101
        xc = np.stack([orc.propagate(model, mdl.par(), N, xd[b], ud[b], p[b], res=res)[1] for b in range(B)])
        np.savez_compressed(os.path.join(HERE, "propagate_%s.npz" % model), N=N, res=res, xd=xd, ud=ud, p=p, xc=xc)
    for model, N, Nsub, iters in PTR:
        pars = ptr_ref.PTRParameters(N, Nsub, iters, 1e3, 0.1, 0, 0, 1e-3)
        st, hist = ptr_ref.ptr_solve(model, pars)
        fin = hist[-1]["sol"]
        np.savez_compressed(os.path.join(HERE, "ptr_%s.npz" % model), N=N, Nsub=Nsub, iters=iters, status=st,
                            J=[h["sub"]["J"] for h in hist], J_tr=[h["sub"]["J_tr"] for h in hist],
                            J_vc=[h["sub"]["J_vc"] for h in hist], J_aug=[h["sub"]["J_aug"] for h in hist],
                            feas=[h["sol"].feas for h in hist], xd=fin.xd, ud=fin.ud, p=fin.p)
        print(model, st, hist[-1]["sub"]["J"])
    make_cfg()


if __name__ == "__main__":
    main()
