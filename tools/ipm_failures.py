"""List the subproblems whose structured-IPM exit status is not (ALMOST_)OPTIMAL in a Monte-Carlo PTR run."""
import sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
import bench
model = sys.argv[1] if len(sys.argv) > 1 else "rocket_landing"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
opts = {}
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    opts[k] = float(v) if "." in v or "e" in v else int(v)
_, N, Nsub, iters, _ = bench.WORKLOADS[model]
traj = pkg.TrajectoryProblem(model)
pp = bench.mc_pp(traj.mdl, B, 0)
pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0, solver_opts=opts)
pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
sol, h = pkg.PTR.solve(pbm, pp)
bad = np.argwhere(h.solver_status > 1)
print("%s B=%d opts %s: solved %.4f feas %.4f, %d failing subproblems, mean ipm iters %.1f" % (
    model, B, opts, np.mean([s == "SCP_SOLVED" for s in sol.status]), sol.feas.mean(), len(bad), h.solver_iters[h.active].mean()))
for it, b in bad[:40]:
    print("  PTR it %2d problem %4d: status %d iters %3d gap %.2e pres %.2e dres %.2e  J_vc %.2e" % (
        it + 1, b, h.solver_status[it, b], h.solver_iters[it, b], h.gap[it, b], h.pres[it, b], h.dres[it, b], h.J_vc[it, b]))
pbm.close()
