"""GPU debug aid: SCvx device loop vs the oracle loop, iteration table."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
from oracle import scvx_ref
from oracle.models import MODELS
pkg = graft.load_package()
N, Nsub, iters = 30, 15, 15
op = scvx_ref.quadrotor_test_parameters(N, Nsub, iters)
mdl = MODELS["quadrotor"]()
rng = np.random.default_rng(3)
pps = [mdl.nominal_pp()]
for _ in range(2):
    q = mdl.nominal_pp().copy(); q[6:9] *= 1 + 0.1 * rng.uniform(-1, 1, 3); pps.append(q)
traj = pkg.TrajectoryProblem("quadrotor")
pars = pkg.SCvx.Parameters(N=N, Nsub=Nsub, iter_max=iters, lam=op.lam, rho_0=op.rho_0, rho_1=op.rho_1, rho_2=op.rho_2,
                           beta_sh=op.beta_sh, beta_gr=op.beta_gr, eta_init=op.eta_init, eta_lb=op.eta_lb, eta_ub=op.eta_ub)
pbm = pkg.SCvx.create(pars, traj, batch_capacity=3)
sol, hist = pkg.SCvx.solve(pbm, np.stack(pps))
for b in range(3):
    st, oh = scvx_ref.scvx_solve("quadrotor", op, pp=pps[b])
    print("problem", b, st, sol.status[b], sol.iterations[b], len(oh))
    for k, rec in enumerate(oh):
        print(" k=%2d eta %.4g/%.4g acc %d/%d L % .8e/% .8e Lpen %.3e/%.3e Jsol % .8e/% .8e rho % .4f/% .4f ipm %d/%d st %d" % (
            k + 1, hist["eta"][k, b], rec["eta"], hist["accepted"][k, b], rec.get("accept", -1), hist["L"][k, b], rec["sub"]["L"],
            hist["L_pen"][k, b], rec["sub"]["L_pen"], hist["J_sol"][k, b], rec.get("J_sol", np.nan), hist["rho"][k, b], rec.get("rho", np.nan),
            hist["solver_iters"][k, b], rec["sub"]["ipm"]["iters"], hist["solver_status"][k, b]), flush=True)
