"""Phase breakdown (wall-clock ticks inside the kernel) of the last structured-IPM launch, averaged over problems."""
import sys, ctypes
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
import bench
model = sys.argv[1] if len(sys.argv) > 1 else "rocket_landing"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
niter = int(sys.argv[3]) if len(sys.argv) > 3 else 3
_, N, Nsub, iters, _ = bench.WORKLOADS[model]
traj = pkg.TrajectoryProblem(model)
pp = bench.mc_pp(traj.mdl, B, 0)
pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=niter, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0)
pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
sol, h = pkg.PTR.solve(pbm, pp)
import os
names = ["G", "GT", "factor", "rhs+fwd", "bwd", "arrow", "finish", "total"]
if "fprof" in os.environ.get("SCP_MI355X_LIB", ""):
    names = ["f:Ysoc+Sz+C0", "f:chol Sz", "factor(total)", "f:Y", "f:cf+Snu", "f:chol Snu", "f:X", "total"]
if "profo" in os.environ.get("SCP_MI355X_LIB", ""):
    names = ["residual passes+snapshots", "nt_update", "combined rhs", "refinement residuals", "refinement updates", "step-length pass", "step trial+update", "total"]
acc = np.zeros(8)
nb = min(B, 64)
for b in range(nb):
    t = (ctypes.c_longlong * 8)()
    pkg._lib.lib().scp_debug_get_ipm_profile(pbm.handle, b * (B // nb), t)
    acc += np.array(list(t), float)
acc /= nb
its = h.solver_iters[-1].mean()
print("%s B=%d, last launch: mean IPM iterations %.1f, total %.1f ms" % (model, B, its, acc[7] / 1e5))
other = acc[7] - acc[:7].sum()
for n, v in list(zip(names[:7], acc[:7])) + [("other" if "profo" in os.environ.get("SCP_MI355X_LIB", "") else "other(light passes)", other)]:
    print("  %-26s %8.2f ms  %5.1f %%   %.3f ms/iter" % (n, v / 1e5, 100 * v / acc[7], v / 1e5 / its))
ksec, kcnt = pkg.PTR.kernel_timing(pbm)
print("kernel seconds", ksec, kcnt)
pbm.close()
