"""Run the CPU baseline (oracle/cpu_ptr.cpp, same algorithm as the device) on the bench's Monte-Carlo batch and save
per-instance statistics (worst IPM status, dynamic feasibility, IPM iterations) for comparison with the device's."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import cpu_ptr
from oracle.models import MODELS
import bench

model = sys.argv[1] if len(sys.argv) > 1 else "rocket_landing"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
out = sys.argv[4] if len(sys.argv) > 4 else "/tmp/cpu_batch_%s_%d.npz" % (model, B)
mdl = MODELS[model]()
pp = bench.mc_pp(mdl, B, 0)
t0 = time.time()
r = cpu_ptr.solve_batch(model, N, 15, 15, pp, threads=0, want_hist=True)
print("B=%d %.1f s wall, %d threads -> %.1f SCP it/s" % (B, r["seconds"], cpu_ptr.max_threads(), B * 15 / r["seconds"]))
st = r["stats"]
print("worst ipm status counts", np.bincount(st[:, 1].astype(int), minlength=4), "frac feas", st[:, 2].mean(), "mean ipm iters/solve", st[:, 0].mean() / 15)
np.savez_compressed(out, stats=st, hist=r["hist"], xd=r["xd"], ud=r["ud"], p=r["p"], pp=pp)
