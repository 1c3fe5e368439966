"""Golden outcomes of the ORACLE's literal PTR loop (oracle/ptr_ref.py + oracle/ipm.py) on instances of the BENCH batch
(rocket landing, N = 100, Nsub = 15, 15 iterations, Monte-Carlo seed = problem index, bench.py::mc_pp): the 13 instances on
which the round-2 device solver ended SCP_FAILED and a sample of those it left dynamically infeasible.

    python tests/golden/make_bench_outcomes.py          # ~1 min per instance and core

The literal loop SOLVES all of them (every subproblem OPTIMAL): the failures were the structured solver's, not the
problems'; the instances that stay infeasible (virtual control J_vc > 0 after 15 iterations) do so in the literal loop too."""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
FAILED_R02 = [87, 1288, 1415, 1510, 1522, 1852, 2376, 2722, 2809, 3017, 3701, 3952, 3967]
INFEASIBLE_SAMPLE = [9, 22, 30, 66, 82, 113]


def run_one(idx):
    os.environ["OMP_NUM_THREADS"] = "1"
    import bench
    from oracle import ptr_ref
    from oracle.models import MODELS
    mdl = MODELS["rocket_landing"]()
    pp = bench.mc_pp(mdl, idx + 1, 0)[idx]
    pars = ptr_ref.PTRParameters(100, 15, 15, 1e3, 0.1, 0, 0, 1e-3)
    st, hist = ptr_ref.ptr_solve("rocket_landing", pars, pp=pp)
    fin = hist[-1]["sol"]
    return dict(idx=idx, status=st, n_hist=len(hist), feas=bool(fin.feas), J_vc=[h["sub"]["J_vc"] for h in hist],
                J_aug=[h["sub"]["J_aug"] for h in hist], ipm_ok=all(h["sub"]["status"] == "OPTIMAL" for h in hist), pp=pp,
                xd=fin.xd, p=fin.p)


def main():
    idx = FAILED_R02 + INFEASIBLE_SAMPLE
    with ProcessPoolExecutor(max_workers=min(len(idx), os.cpu_count() or 1)) as ex:
        res = list(ex.map(run_one, idx))
    np.savez_compressed(os.path.join(HERE, "bench_outcomes_rocket_landing_N100.npz"), idx=np.array(idx),
                        failed_r02=np.array(FAILED_R02), solved=np.array([r["status"] == "SCP_SOLVED" for r in res]),
                        ipm_all_optimal=np.array([r["ipm_ok"] for r in res]), feas=np.array([r["feas"] for r in res]),
                        J_vc=np.array([r["J_vc"] for r in res]), J_aug=np.array([r["J_aug"] for r in res]),
                        pp=np.array([r["pp"] for r in res]), xd=np.array([r["xd"] for r in res]), p=np.array([r["p"] for r in res]))
    for r in res:
        print(r["idx"], r["status"], r["feas"], r["J_vc"][-1])


if __name__ == "__main__":
    main()
