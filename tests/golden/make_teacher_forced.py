"""TEACHER-FORCED subproblem goldens ("next" 1b): for every instance x iteration of the oracle's literal SCvx and
GuSTO loops on the quadrotor Monte-Carlo instances of the bench records (reference test parameters quadrotor/tests.jl:32-75 /
:86-130, N = 30, Nsub = 15, 6 iterations, goal +-10 %, seed = instance index) this stores the REFERENCE the oracle linearised
about (xd, ud, p), the per-iteration scalars (eta; lambda for GuSTO) and the optimal value of the oracle's literal conic program
(`pcost` = L_aug, plus its split).  tests/test_teacher_forced_gpu.py hands exactly these references to the DEVICE subproblem
(discretize! -> linearise -> gather -> conic_ipm_kernel) and compares optimal values to 1e-6: solver parity on every
subproblem the oracle loop ever formulated, independent of which path the device LOOP would have taken.

    python tests/golden/make_teacher_forced.py [instances = 64] [processes = 7]      # ~4 s per instance
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
ITERS = 6
N, NSUB = 30, 15


def run(b):
    os.environ["OMP_NUM_THREADS"] = "1"
    import bench
    from oracle import gusto_ref, scvx_ref
    from oracle.models import MODELS
    mdl = MODELS["quadrotor"]()
    pp = bench.mc_pp(mdl, 1, b)[0]          # seed = instance index
    out = {}
    sp_ = scvx_ref.quadrotor_test_parameters(N, NSUB, ITERS)
    st, h = scvx_ref.scvx_solve("quadrotor", sp_, pp=pp)
    out["scvx"] = (st, [dict(xd=r["ref"].xd, ud=r["ref"].ud, p=r["ref"].p, eta=r["eta"], lam=np.nan, pcost=r["sub"]["L_aug"],
                             L=r["sub"]["L"], L_aug=r["sub"]["L_aug"], ok=r["sub"]["status"] in ("OPTIMAL", "ALMOST_OPTIMAL"),
                             sx=r["sub"]["x"], su=r["sub"]["u"], sp=r["sub"]["p"]) for r in h])
    gp = gusto_ref.quadrotor_test_parameters(N, NSUB, ITERS)
    gp.eps_abs = gp.eps_rel = 0.0
    st, h = gusto_ref.gusto_solve("quadrotor", gp, pp=pp, ipm_opts=dict(normalise_objective=True))
    out["gusto"] = (st, [dict(xd=r["ref"].xd, ud=r["ref"].ud, p=r["ref"].p, eta=r["eta"], lam=r["lam"], pcost=r["sub"]["pcost"],
                              L=r["sub"]["L"], L_aug=r["sub"]["L_aug"], ok=r["sub"]["status"] in ("OPTIMAL", "ALMOST_OPTIMAL"),
                              sx=r["sub"]["x"], su=r["sub"]["u"], sp=r["sub"]["p"]) for r in h])
    return b, pp, out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    with mp.Pool(procs) as pool:
        res = pool.map(run, range(n), chunksize=2)
    res.sort(key=lambda r: r[0])
    from oracle.models import MODELS
    mdl = MODELS["quadrotor"]()
    for algo in ("scvx", "gusto"):
        xd = np.zeros((n, ITERS, N, mdl.nx)); ud = np.zeros((n, ITERS, N, mdl.nu)); p = np.zeros((n, ITERS, mdl.np))
        sx = np.zeros_like(xd); su = np.zeros_like(ud); sp = np.zeros_like(p)
        eta = np.full((n, ITERS), np.nan); lam = np.full((n, ITERS), np.nan); pc = np.full((n, ITERS), np.nan)
        L = np.full((n, ITERS), np.nan); La = np.full((n, ITERS), np.nan); valid = np.zeros((n, ITERS), bool)
        for b, pp, out in res:
            for k, r in enumerate(out[algo][1]):
                xd[b, k], ud[b, k], p[b, k] = r["xd"], r["ud"], r["p"]
                sx[b, k], su[b, k], sp[b, k] = r["sx"], r["su"], r["sp"]
                eta[b, k], lam[b, k], pc[b, k], L[b, k], La[b, k], valid[b, k] = r["eta"], r["lam"], r["pcost"], r["L"], r["L_aug"], r["ok"]
        np.savez_compressed(os.path.join(HERE, "teacher_forced_%s_quadrotor_N30.npz" % algo), pp=np.stack([r[1] for r in res]),
                            ref_xd=xd, ref_ud=ud, ref_p=p, eta=eta, lam=lam, pcost=pc, L=L, L_aug=La, valid=valid,
                            sol_p=sp, sol_xd=sx.astype(np.float32), sol_ud=su.astype(np.float32),
                            solved=np.array([r[2][algo][0].split()[0] == "SCP_SOLVED" for r in res]), N=N, Nsub=NSUB, iter_max=ITERS)
        print(algo, "subproblems", int(valid.sum()), "of", n * ITERS)


if __name__ == "__main__":
    main()
