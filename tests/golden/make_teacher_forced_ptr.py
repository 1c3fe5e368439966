"""TEACHER-FORCED subproblem golden for the HEADLINE workload (PTR, rocket landing, N = 100, Nsub = 15, 15 iterations, w_vc = 1e3,
w_tr = 0.1, Monte-Carlo seed = instance index, bench.py::mc_pp): for every iteration of the oracle's literal PTR loop
(oracle/ptr_ref.py: every subproblem a literal conic program through oracle/ipm.py) on the first instances of the bench batch
the REFERENCE the oracle linearised about and the cost split (J, J_tr, J_vc, J_aug) of its optimum.
tests/test_teacher_forced_gpu.py hands these references to the device's stage-structured path (K2 -> K3 -> K4a,
scp_ptr_subproblem) and compares J_aug and J_vc to 1e-6 on every one -- the complement of tests/test_config_size_gpu.py, which does
this for iterations 1, 4 and 12 of four instances.

    OMP_NUM_THREADS=1 python tests/golden/make_teacher_forced_ptr.py [instances = 16] [processes = 14]     # ~2 min per instance and core
    OMP_NUM_THREADS=1 python tests/golden/make_teacher_forced_ptr.py 8 14 quadrotor 50     # BASELINE.json configs[1]: quadrotor obstacle avoidance, N = 50
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
MODEL = sys.argv[3] if len(sys.argv) > 3 else "rocket_landing"
N, NSUB, ITERS = (int(sys.argv[4]) if len(sys.argv) > 4 else 100), 15, 15


def run_one(idx):
    os.environ["OMP_NUM_THREADS"] = "1"
    import bench
    from oracle import ptr_ref
    from oracle.models import MODELS
    mdl = MODELS[MODEL]()
    pp = bench.mc_pp(mdl, 1, idx)[0]            # seed = instance index
    pars = ptr_ref.PTRParameters(N, NSUB, ITERS, 1e3, 0.1, 0, 0, 1e-3)
    st, hist = ptr_ref.ptr_solve(MODEL, pars, pp=pp)
    return idx, pp, st, [dict(xd=h["ref"].xd, ud=h["ref"].ud, p=h["ref"].p, cost=[h["sub"]["J"], h["sub"]["J_tr"], h["sub"]["J_vc"], h["sub"]["J_aug"]],
                              ok=h["sub"]["status"] in ("OPTIMAL", "ALMOST_OPTIMAL"), opt=h["sub"]["status"] == "OPTIMAL", sp=h["sub"]["p"]) for h in hist]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    with mp.Pool(procs) as pool:
        res = pool.map(run_one, range(n), chunksize=1)
    res.sort(key=lambda r: r[0])
    from oracle.models import MODELS
    mdl = MODELS[MODEL]()
    xd = np.zeros((n, ITERS, N, mdl.nx)); ud = np.zeros((n, ITERS, N, mdl.nu)); p = np.zeros((n, ITERS, mdl.np)); sp = np.zeros_like(p)
    cost = np.full((n, ITERS, 4), np.nan); valid = np.zeros((n, ITERS), bool); opt = np.zeros((n, ITERS), bool)
    for b, pp, st, recs in res:
        for k, r in enumerate(recs):
            xd[b, k], ud[b, k], p[b, k], sp[b, k], cost[b, k], valid[b, k], opt[b, k] = r["xd"], r["ud"], r["p"], r["sp"], r["cost"], r["ok"], r["opt"]
    np.savez_compressed(os.path.join(HERE, "teacher_forced_ptr_%s_N%d.npz" % (MODEL, N)), pp=np.stack([r[1] for r in res]), ref_xd=xd,
                        ref_ud=ud, ref_p=p, cost=cost, valid=valid, optimal=opt, sol_p=sp,
                        solved=np.array([r[2] == "SCP_SOLVED" for r in res]), N=N, Nsub=NSUB, iter_max=ITERS)
    print("subproblems", int(valid.sum()), "of", n * ITERS, "OPTIMAL", int(opt.sum()))


if __name__ == "__main__":
    main()
