"""IPM iteration counts of the device solver vs the numpy mirror on the SAME converged-regime subproblem (nominal rocket
landing), for a few solver-option variants."""
import sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
g.load_oracle()
from oracle import ptr_ref, ipm_struct
from oracle.models import MODELS
N = 100
mdl = MODELS["rocket_landing"]()
opars = ptr_ref.PTRParameters(N, 15, 15, 1e3, 0.1, 0, 0, 1e-3)
scale = ptr_ref.Scaling(*mdl.bbox())
pp = mdl.nominal_pp()
traj = pkg.TrajectoryProblem("rocket_landing")
# converged reference from the device PTR loop itself
pars = pkg.PTR.Parameters(N=N, Nsub=15, iter_max=8, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0)
pbm = pkg.PTR.create(pars, traj, batch_capacity=1)
sol, h = pkg.PTR.solve(pbm, pp[None])
print("device PTR ipm iters per iteration:", h.solver_iters[:, 0])
pbm.close()
ref = ptr_ref.discretize(mdl, opars, scale, sol.xd[0], sol.ud[0], sol.p[0])
P = ipm_struct.build_stage_problem(mdl, opars, scale, ref, pp)
tr = []
a = ipm_struct.solve(P, trace=tr)
print("mirror on the device's converged reference: %s %d/%s gap %.1e dres %.1e" % (a["status"], a["iters"], a.get("iters_total"), a["gap"], a["dres"]))
for opts in [dict(), dict(nref=2), dict(nref=0), dict(reg=1e-12), dict(reg=1e-8), dict(ref_gap=1e30), dict(stall=6), dict(nref=2, ref_gap=1e30)]:
    pars = pkg.PTR.Parameters(N=N, Nsub=15, iter_max=1, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0, solver_opts=opts)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=1)
    gsub = pkg.PTR.solve_subproblem_(pbm, ref.xd[None], ref.ud[None], ref.p[None], pp[None])
    info = gsub["info"][0]
    print("device %-28s status %d iters %3d best_it %3d gap %.1e pres %.1e dres %.1e J_aug %.10e" % (
        opts, gsub["status"][0], gsub["iters"][0], int(info[7]), info[2], info[3], info[4], gsub["J_aug"][0]))
    pbm.close()
