"""debug: device GuSTO history next to the oracle's (run on the GPU box)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import gusto_ref
from oracle.models import MODELS
sys.path.insert(0, "tests")
from test_gusto_gpu import make_pars

N, Nsub, iters = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (30, 15, 15)
op = gusto_ref.quadrotor_test_parameters(N, Nsub, iters)
mdl = MODELS["quadrotor"]()
pp = mdl.nominal_pp()
traj = pkg.TrajectoryProblem("quadrotor")
pbm = pkg.GuSTO.create(make_pars(pkg, op), traj, batch_capacity=1)
sol, hist = pkg.GuSTO.solve(pbm, pp[None])
print("device status", sol.status, sol.iterations)
st, oh = gusto_ref.gusto_solve("quadrotor", op, pp=pp)
print("oracle status", st, len(oh))
keys = ("L", "L_st", "L_tr", "J_aug", "J_st", "rho", "eta", "lam", "flags", "deviation", "solver_status", "solver_iters", "dyn_error", "dyn_nrml")
for k in range(iters):
    print("k=%d dev " % (k + 1) + " ".join("%s=%.6g" % (n, hist[n][k, 0]) for n in keys))
    if k < len(oh):
        r = oh[k]
        print("     orc L=%.6g L_st=%.6g L_tr=%.6g J_aug=%.6g J_st=%.6g rho=%.6g eta=%.6g lam=%.6g acc=%s dev=%.6g st=%s dyn_error=%.6g" % (
            r["sub"]["L"], r["sub"]["L_st"], r["sub"]["L_tr"], r.get("J_aug", np.nan), r.get("J_st", np.nan), r.get("rho", np.nan), r["eta"], r["lam"],
            r.get("accept"), r.get("deviation", np.nan), r["sub"]["status"], r.get("dyn_error", np.nan)))
