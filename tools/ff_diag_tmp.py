import sys
sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from oracle.models import MODELS
N, Nsub = 12, 8
mdl = MODELS["freeflyer"](N)
traj = pkg.TrajectoryProblem("freeflyer")
pars = pkg.GuSTO.Parameters(N=N, Nsub=Nsub, iter_max=2, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.5, beta_sh=2.0,
                            beta_gr=2.0, gamma_fail=5.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=16,
                            eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
pbm = pkg.GuSTO.create(pars, traj, batch_capacity=1)
sol, hist = pkg.GuSTO.solve(pbm, mdl.nominal_pp()[None])
print("status", sol.status, "iters", sol.iterations, "solver_status", hist["solver_status"][:,0], "ipm its", hist["solver_iters"][:,0], "eta", hist["eta"][:,0], "flags", hist["flags"][:,0], "L", hist["L"][:,0], hist["L_st"][:,0], hist["L_tr"][:,0])
import ctypes
st = np.zeros(12, np.int64)
pkg._lib.lib().scp_conic_stats  # exists
pbm.close()
