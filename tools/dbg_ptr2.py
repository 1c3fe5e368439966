import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
from oracle import ptr_ref
from oracle.models import MODELS
model = sys.argv[1]; N = int(sys.argv[2]); iters = int(sys.argv[3])
traj = pkg.TrajectoryProblem(model)
pars = pkg.PTR.Parameters(N=N, Nsub=15, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0)
pbm = pkg.PTR.create(pars, traj, batch_capacity=1)
sol, h = pkg.PTR.solve(pbm, None)
o = orc.discretize(model, orc.default_params(model), N, 15, sol.xd, sol.ud, sol.p, pbm.scale.iSx, 1e-3)
print("gpu defect max scaled", np.abs(sol.defect[0] / pbm.scale.Sx).max(), "oracle defect max scaled", np.abs(o["defect"][0] / pbm.scale.Sx).max(), "feas", sol.feas, o["feas"])
print("Jvc hist", h.J_vc[:, 0])
mdl = MODELS[model](); opars = ptr_ref.PTRParameters(N, 15, iters, 1e3, 0.1, 0, 0, 1e-3); scale = ptr_ref.Scaling(*mdl.bbox())
ref = ptr_ref.discretize(mdl, opars, scale, sol.xd[0], sol.ud[0], sol.p[0])
sub = ptr_ref.solve_subproblem(mdl, opars, scale, ref, mdl.nominal_pp())
print("oracle subproblem about GPU final: J %.6e Jtr %.3e Jvc %.3e status %s" % (sub["J"], sub["J_tr"], sub["J_vc"], sub["status"]))
gsub = pkg.PTR.solve_subproblem_(pbm, sol.xd, sol.ud, sol.p)
print("gpu subproblem about GPU final:    J %.6e Jtr %.3e Jvc %.3e status %s" % (gsub["J"][0], gsub["J_tr"][0], gsub["J_vc"][0], gsub["status"]))
