"""Per-PTR-iteration statistics of the structured IPM (iterations, exit status) on the bench workload."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
import bench
model = sys.argv[1] if len(sys.argv) > 1 else "rocket_landing"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
_, N, Nsub, iters, _ = bench.WORKLOADS[model]
traj = pkg.TrajectoryProblem(model)
pp = bench.mc_pp(traj.mdl, B, 0)
pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0)
pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
t0 = time.time()
sol, h = pkg.PTR.solve(pbm, pp)
print("%s B=%d: %.2fs" % (model, B, time.time() - t0))
for k in range(h.solver_iters.shape[0]):
    it = h.solver_iters[k]
    st = np.bincount(h.solver_status[k].astype(int), minlength=4)
    print("PTR it %2d: ipm iters mean %.1f min %d med %d p90 %d max %d | status opt/almost/itlim/num %s | J_vc med %.2e dev med %.2e"
          % (k + 1, it.mean(), it.min(), np.median(it), np.percentile(it, 90), it.max(), st,
             np.median(h.J_vc[k]), np.median(h.deviation[k])))
pbm.close()
