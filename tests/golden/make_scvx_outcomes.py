"""Outcomes of the ORACLE's literal SCvx loop (oracle/scvx_ref.py) on the Monte-Carlo instances of bench.py's `scvx_quadrotor`
record (quadrotor, reference test parameters quadrotor/tests.jl:32-75, N = 30, Nsub = 15, 6 iterations, goal position +-10 %,
seed = instance index): status, dynamic feasibility and the per-iteration record of every instance (linearised cost L of the
subproblem, nonlinear cost J of its solution, performance ratio rho, radius, accept / reject).  bench.py and
tests/test_outcomes_gpu.py compare the device loop with these instance by instance and iteration by iteration.

    python tests/golden/make_scvx_outcomes.py [instances = 1024] [processes = 7]
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
ITERS = 6


def run(b):
    os.environ["OMP_NUM_THREADS"] = "1"
    import bench
    from oracle import scvx_ref
    from oracle.models import MODELS
    mdl = MODELS["quadrotor"]()
    pp = bench.mc_pp(mdl, 1, b)[0]          # seed = instance index
    sp_ = scvx_ref.SCvxParameters(30, 15, ITERS, lam=30.0, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                  eta_lb=1e-3, eta_ub=10.0, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
    st, h = scvx_ref.scvx_solve("quadrotor", sp_, pp=pp)
    L = np.full(ITERS, np.nan); J = np.full(ITERS, np.nan); rho = np.full(ITERS, np.nan); eta = np.full(ITERS, np.nan)
    acc = np.full(ITERS, -1, np.int8)
    for k, r in enumerate(h):
        L[k] = r["sub"]["L"]; eta[k] = r["eta"]; J[k] = r.get("J_sol", np.nan); rho[k] = r.get("rho", np.nan)
        if "accept" in r:
            acc[k] = int(r["accept"])
    return b, 0 if st.split()[0] == "SCP_SOLVED" else 1, len(h), bool(h[-1]["sol"].feas), L, J, rho, eta, acc


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    with mp.Pool(procs) as pool:
        res = pool.map(run, range(n), chunksize=4)
    res.sort(key=lambda r: r[0])
    its = np.array([r[2] for r in res], np.int16); L = np.stack([r[4] for r in res]); acc = np.stack([r[8] for r in res])
    np.savez_compressed(os.path.join(HERE, "scvx_outcomes_quadrotor_N30.npz"), status=np.array([r[1] for r in res], np.int8),
                        iterations=its, accepted=(acc == 1).sum(axis=1).astype(np.int16), L_last=L[np.arange(len(res)), its - 1],
                        feas_last=np.array([r[3] for r in res]), L=L, J_sol=np.stack([r[5] for r in res]), rho=np.stack([r[6] for r in res]),
                        eta=np.stack([r[7] for r in res]), accept=acc, N=30, Nsub=15, iter_max=ITERS)
    print("solved %.4f accepted fraction %.4f" % (np.mean([r[1] == 0 for r in res]), (acc == 1).sum() / its.sum()))


if __name__ == "__main__":
    main()
