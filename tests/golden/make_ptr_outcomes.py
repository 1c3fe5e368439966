"""Outcomes of the ORACLE's literal PTR loop (oracle/ptr_ref.py + oracle/ipm.py) on the FIRST instances of the headline bench
batch (rocket landing, N = 100, Nsub = 15, 15 iterations, Monte-Carlo seed = problem index, bench.py::mc_pp): status, dynamic
feasibility, augmented cost and virtual-control cost of the last subproblem, final time.  bench.py compares the device
batch with these instance by instance (`oracle_outcomes` of the headline line).

    python tests/golden/make_ptr_outcomes.py [instances = 256] [processes = 14]       # ~1 min per instance and core
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def run_one(idx):
    os.environ["OMP_NUM_THREADS"] = "1"      # (export it in the parent too: numpy reads it at import)
    import bench
    from oracle import ptr_ref
    from oracle.models import MODELS
    mdl = MODELS["rocket_landing"]()
    pp = bench.mc_pp(mdl, 1, idx)[0]            # seed = instance index
    pars = ptr_ref.PTRParameters(100, 15, 15, 1e3, 0.1, 0, 0, 1e-3)
    st, hist = ptr_ref.ptr_solve("rocket_landing", pars, pp=pp)
    fin = hist[-1]["sol"]
    return (idx, 0 if st == "SCP_SOLVED" else 1, bool(fin.feas), float(hist[-1]["sub"]["J_aug"]), float(hist[-1]["sub"]["J_vc"]),
            float(fin.p[0]), all(h["sub"]["status"] == "OPTIMAL" for h in hist))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    with mp.Pool(procs) as pool:
        res = pool.map(run_one, range(n), chunksize=2)
    res.sort()
    np.savez_compressed(os.path.join(HERE, "ptr_outcomes_rocket_landing_N100.npz"), status=np.array([r[1] for r in res], np.int8),
                        feas=np.array([r[2] for r in res]), J_aug=np.array([r[3] for r in res]), J_vc=np.array([r[4] for r in res]),
                        tf=np.array([r[5] for r in res]), ipm_all_optimal=np.array([r[6] for r in res]), N=100, Nsub=15, iter_max=15)
    print("solved %.4f feasible %.4f all subproblems OPTIMAL %.4f" % (np.mean([r[1] == 0 for r in res]), np.mean([r[2] for r in res]),
                                                                      np.mean([r[6] for r in res])))


if __name__ == "__main__":
    main()
