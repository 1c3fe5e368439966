import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
sys.argv = sys.argv
import bench
model, N, Nsub, iters, B = "rocket_landing", 100, 15, 15, 256
traj = pkg.TrajectoryProblem(model)
pp = bench.mc_pp(traj.mdl, B, 0)
for ref_gap in [1e30, 1e-1, 1e-2, 1e-3]:
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0, solver_opts=dict(ref_gap=ref_gap))
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    t0 = time.time()
    sol, h = pkg.PTR.solve(pbm, pp)
    dt = time.time() - t0
    print("ref_gap %g: %.2fs solved %.4f feas %.4f mean ipm it %.1f status counts %s" % (ref_gap, dt, np.mean([s == "SCP_SOLVED" for s in sol.status]), sol.feas.mean(), h.solver_iters[h.active].mean(), np.unique(h.solver_status[h.active], return_counts=True)))
    pbm.close()
