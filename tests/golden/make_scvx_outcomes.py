"""Outcomes of the ORACLE's literal SCvx loop (oracle/scvx_ref.py) on the Monte-Carlo instances of bench.py's `scvx_quadrotor`
record (quadrotor, reference test parameters quadrotor/tests.jl:32-75, N = 30, Nsub = 15, 6 iterations, goal position +-10 %,
seed = instance index): status, accepted steps, the linearised cost L and the nonlinear cost J of the last iteration,
dynamic feasibility.  bench.py compares the device loop with these instance by instance.

    python tests/golden/make_scvx_outcomes.py [instances = 1024] [processes = 12]
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def run(b):
    os.environ["OMP_NUM_THREADS"] = "1"
    import bench
    from oracle import scvx_ref
    from oracle.models import MODELS
    mdl = MODELS["quadrotor"]()
    pp = bench.mc_pp(mdl, 1, b)[0]          # seed = instance index
    sp_ = scvx_ref.SCvxParameters(30, 15, 6, lam=30.0, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                  eta_lb=1e-3, eta_ub=10.0, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
    st, h = scvx_ref.scvx_solve("quadrotor", sp_, pp=pp)
    acc = [bool(r.get("accept", False)) for r in h]
    return b, 0 if st.split()[0] == "SCP_SOLVED" else 1, len(h), sum(acc[:-1]) + int(acc[-1]), float(h[-1]["sub"]["L"]), bool(h[-1]["sol"].feas)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    with mp.Pool(procs) as pool:
        res = pool.map(run, range(n), chunksize=4)
    res.sort()
    np.savez_compressed(os.path.join(HERE, "scvx_outcomes_quadrotor_N30.npz"), status=np.array([r[1] for r in res], np.int8),
                        iterations=np.array([r[2] for r in res], np.int16), accepted=np.array([r[3] for r in res], np.int16),
                        L_last=np.array([r[4] for r in res]), feas_last=np.array([r[5] for r in res]), N=30, Nsub=15, iter_max=6)
    print("solved %.4f accepted fraction %.4f" % (np.mean([r[1] == 0 for r in res]), np.sum([r[3] for r in res]) / np.sum([r[2] for r in res])))


if __name__ == "__main__":
    main()
