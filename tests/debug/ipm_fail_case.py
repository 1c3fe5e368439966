"""Reproduce one failing Monte-Carlo instance (device IPM hits the iteration limit) and solve the SAME subproblem with the
numpy mirror."""
import sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
g.load_oracle()
import bench
from oracle import ptr_ref, ipm_struct
from oracle.models import MODELS
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 209
kfail = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = 100
mdl = MODELS["rocket_landing"]()
opars = ptr_ref.PTRParameters(N, 15, 15, 1e3, 0.1, 0, 0, 1e-3)
scale = ptr_ref.Scaling(*mdl.bbox())
traj = pkg.TrajectoryProblem("rocket_landing")
pp = bench.mc_pp(traj.mdl, 1, seed)
pars = pkg.PTR.Parameters(N=N, Nsub=15, iter_max=kfail - 1, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0)
pbm = pkg.PTR.create(pars, traj, batch_capacity=1)
sol, h = pkg.PTR.solve(pbm, pp)
print("device PTR its 1..%d: ipm iters %s status %s" % (kfail - 1, h.solver_iters[:, 0], h.solver_status[:, 0]))
pbm.close()
ref = ptr_ref.discretize(mdl, opars, scale, sol.xd[0], sol.ud[0], sol.p[0])
for opts in [dict(), dict(maxit=300), dict(nref=2), dict(reg=1e-9)]:
    pars1 = pkg.PTR.Parameters(N=N, Nsub=15, iter_max=1, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0, solver_opts=opts)
    pb = pkg.PTR.create(pars1, traj, batch_capacity=1)
    gs = pkg.PTR.solve_subproblem_(pb, ref.xd[None], ref.ud[None], ref.p[None], pp)
    i = gs["info"][0]
    print("device %-16s status %d iters %3d best_it %3d gap %.1e pres %.1e dres %.1e J_aug %.8e J_vc %.3e" % (
        opts, gs["status"][0], gs["iters"][0], int(i[7]), i[2], i[3], i[4], gs["J_aug"][0], gs["J_vc"][0]))
    pb.close()
P = ipm_struct.build_stage_problem(mdl, opars, scale, ref, pp[0])
tr = []
a = ipm_struct.solve(P, trace=tr, max_iter=150)
print("mirror: %s %d/%s gap %.1e pres %.1e dres %.1e pcost %.8e" % (a["status"], a["iters"], a.get("iters_total"), a["gap"], a["pres"], a["dres"], a["pcost"]))
for t in tr[::5]:
    print("   %3d gap %.2e pres %.2e dres %.2e" % (t["it"], t["gap"], t["pres"], t["dres"]))
