"""Outcomes of the ORACLE's literal GuSTO loop (oracle/gusto_ref.py) on the Monte-Carlo instances of bench.py's
`gusto_quadrotor` record (quadrotor, reference test parameters quadrotor/tests.jl:86-130, N = 30, Nsub = 15, 6 iterations,
goal position +-10 %, seed = instance index): status, iterations, final cost, dynamic feasibility.  bench.py compares the
device loop's statuses with these instance by instance.

    python tests/golden/make_gusto_outcomes.py [instances = 1024] [processes = 12]      # ~2 s per instance
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
    from oracle import gusto_ref
    from oracle.models import MODELS
    mdl = MODELS["quadrotor"]()
    pp = bench.mc_pp(mdl, 1, b)[0]          # seed = instance index
    op = gusto_ref.quadrotor_test_parameters(30, 15, 6)
    op.eps_abs = op.eps_rel = 0.0
    st, oh = gusto_ref.gusto_solve("quadrotor", op, pp=pp)
    last = oh[-1]
    return b, 0 if st.split()[0] == "SCP_SOLVED" else 1, len(oh), float(last.get("J_aug", np.nan)), float(last.get("lam", np.nan))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    with mp.Pool(procs) as pool:
        res = pool.map(run, range(n), chunksize=4)
    res.sort()
    status = np.array([r[1] for r in res], np.int8); iters = np.array([r[2] for r in res], np.int16)
    np.savez_compressed(os.path.join(HERE, "gusto_outcomes_quadrotor_N30.npz"), status=status, iterations=iters,
                        J_aug=np.array([r[3] for r in res]), lam=np.array([r[4] for r in res]), N=30, Nsub=15, iter_max=6)
    print("solved fraction %.4f of %d" % ((status == 0).mean(), n))


if __name__ == "__main__":
    main()
