"""Which instances of the bench batch (rocket landing, N = 100, Monte-Carlo seed = problem index, bench.py::mc_pp) end
SCP_FAILED or dynamically infeasible after the 15 PTR iterations -- writes tests/golden-style lists for
tests/test_failures_*.py:  python tools/failure_instances.py [B] [out.json]"""
import json
import sys

import numpy as np

sys.path.insert(0, '.')
import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402

pkg = g.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/failure_instances.json"
model, N, Nsub, iters, _ = bench.WORKLOADS["rocket_landing"]
traj = pkg.TrajectoryProblem(model)
pp = bench.mc_pp(traj.mdl, B, 0)
pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
sol, h = pkg.PTR.solve(pbm, pp, device_guess=True)
pbm.close()
failed = [int(b) for b in range(B) if sol.status[b] != "SCP_SOLVED"]
infeas = [int(b) for b in range(B) if sol.status[b] == "SCP_SOLVED" and not sol.feas[b]]
rec = dict(B=B, N=N, Nsub=Nsub, iters=iters, failed=failed, infeasible=infeas,
           failed_at={str(b): dict(iteration=int(sol.iterations[b]), status=int(h.solver_status[sol.iterations[b] - 1, b]),
                                   gap=float(h.gap[sol.iterations[b] - 1, b]), pres=float(h.pres[sol.iterations[b] - 1, b]),
                                   J_vc=[float(v) for v in h.J_vc[:sol.iterations[b], b]]) for b in failed},
           infeasible_J_vc_last={str(b): float(h.J_vc[-1, b]) for b in infeas[:400]},
           infeasible_max_scaled_defect={str(b): float(np.abs(sol.defect[b] / pbm.scale.Sx).max()) for b in infeas[:400]},
           almost_fraction=float((h.solver_status[h.active] == 1).mean()), ipm_iters_mean=float(h.solver_iters[h.active].mean()),
           max_pres=float(h.pres[h.active].max()), max_dres=float(h.dres[h.active].max()))
json.dump(rec, open(out, "w"))
print("failed %d infeasible %d almost %.3f ipm iters %.1f max pres %.2e" % (len(failed), len(infeas), rec["almost_fraction"],
                                                                         rec["ipm_iters_mean"], rec["max_pres"]))
