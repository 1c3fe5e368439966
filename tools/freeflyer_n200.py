"""BASELINE.json configs[4] at its stated size on ONE GPU's share of the batch: free-flyer 6-DoF, GuSTO (reference test
parameters, freeflyer/tests.jl:84-140), N = 200, the full iter_max = 15 loop + correct_convex! projection.
    python tools/freeflyer_n200.py [batch = 512] [out.json] [iterations = 15]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

pkg = graft.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 15
t0 = time.perf_counter()
sol, hist, dt = bench.freeflyer_gusto_full_run(pkg, 200, 15, B, iters)
wall = time.perf_counter() - t0
its = sol.iterations
last = hist["L"][np.maximum(its, 1) - 1, np.arange(B)]
rec = dict(workload="freeflyer GuSTO (quadratic penalty, reference test parameters) N=200 Nsub=15, Monte-Carlo batch %d (positions +-3 mm), "
                    "correct_convex! projection + up to %d iterations, PCIe inclusive" % (B, iters),
           solve_seconds=dt, create_and_solve_seconds=wall, scp_iterations_per_s=float(its.sum()) / dt,
           seconds_per_loop_iteration=dt / max(1, int(its.max())),
           frac_solved=float(np.mean([s == "SCP_SOLVED" for s in sol.status])), frac_dyn_feasible=float(sol.feas.mean()),
           iterations_min_med_max=[int(its.min()), int(np.median(its)), int(its.max())],
           accepted_fraction=float(hist["accepted"][:iters].sum() / max(1, its.sum())),
           cost_median=float(np.median(last)), cost_min_max=[float(last.min()), float(last.max())],
           solver_status_counts=[int(v) for v in np.bincount(hist["solver_status"][:iters][hist["solver_iters"][:iters] > 0].astype(int), minlength=4)],
           ipm_iterations_mean=float(hist["solver_iters"][:iters][hist["solver_iters"][:iters] > 0].mean()),
           lam_last_max=float(hist["lam"][np.maximum(its, 1) - 1, np.arange(B)].max()))
s = json.dumps(rec)
print(s)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(s + "\n")
