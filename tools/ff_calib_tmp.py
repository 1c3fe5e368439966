import sys, time
sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import bench
for B, iters in ((256, 2), (1024, 2)):
    t0=time.time()
    r = bench.freeflyer_gusto_record(pkg, N=200, Nsub=15, B=B, iters=iters, full_B=16, full_iters=3)
    print("B", B, "wall", time.time()-t0, {k: r[k] for k in ("scp_iterations_per_s","seconds","template_and_symbolic_seconds","frac_subproblems_safe","ipm_iterations_mean","kernel_seconds","conic_launches")}, r["roofline"]["frac"], r["full_run"], flush=True)
