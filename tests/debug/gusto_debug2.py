"""debug: GuSTO batch of 70 at N=16 with stopping tolerances -- which problems fail and how (run on the GPU box)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import gusto_ref
sys.path.insert(0, "tests")
from test_gusto_gpu import make_pars

op = gusto_ref.quadrotor_test_parameters(16, 10, 14)
op.eps_abs, op.eps_rel = 1e-4, 1e-3
traj = pkg.TrajectoryProblem("quadrotor")
mdl = traj.mdl
rng = np.random.default_rng(5)
pps = np.stack([mdl.nominal_pp() * (1 + 0.03 * rng.uniform(-1, 1, 12)) for _ in range(70)])
pbm = pkg.GuSTO.create(make_pars(pkg, op), traj, batch_capacity=70)
sol, hist = pkg.GuSTO.solve(pbm, pps)
print("status counts", {s: sol.status.count(s) for s in set(sol.status)})
print("iterations", np.bincount(sol.iterations))
bad = [b for b in range(70) if sol.status[b] != "SCP_SOLVED"]
for b in bad[:4]:
    it = sol.iterations[b]
    print("b=%d iters=%d" % (b, it))
    for k in range(it):
        print("   k=%d lam=%.3g eta=%.3g L=%.6g L_st=%.3g L_tr=%.3g J_aug=%.6g rho=%.3g flags=%d st=%d ipm_it=%d dev=%.3g" % (
            k + 1, hist["lam"][k, b], hist["eta"][k, b], hist["L"][k, b], hist["L_st"][k, b], hist["L_tr"][k, b], hist["J_aug"][k, b],
            hist["rho"][k, b], hist["flags"][k, b], hist["solver_status"][k, b], hist["solver_iters"][k, b], hist["deviation"][k, b]))
    st, oh = gusto_ref.gusto_solve("quadrotor", op, pp=pps[b])
    print("   oracle:", st, len(oh), [(r["lam"], r["sub"]["status"], r["sub"]["ipm"]["iters"]) for r in oh])
