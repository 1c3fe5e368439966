"""Re-run given instances of the bench batch (seed = index) with alternative structured-IPM options and print what each
option set does to them:  python tools/failure_probe.py 87,1288,...  [more option sets are listed in SETS]"""
import sys

import numpy as np

sys.path.insert(0, '.')
import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402

pkg = g.load_package()
idx = [int(v) for v in sys.argv[1].split(",")]
SETS = [dict(), dict(ref_gap=1e300), dict(ref_gap=1e300, nref=2), dict(reg=1e-10), dict(reg=2e-11), dict(stall=6), dict(split_step=1),
        dict(ref_gap=1e300, reg=1e-10), dict(maxit=200)]
model, N, Nsub, iters, _ = bench.WORKLOADS["rocket_landing"]
traj = pkg.TrajectoryProblem(model)
pp = bench.mc_pp(traj.mdl, max(idx) + 1, 0)[idx]
for opts in SETS:
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3, solver_opts=opts)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=len(idx))
    sol, h = pkg.PTR.solve(pbm, pp, device_guess=True)
    pbm.close()
    nf = sum(s != "SCP_SOLVED" for s in sol.status)
    print("opts %s: failed %d/%d, feasible %d, ipm iters at PTR it 3: %s, status at it 3: %s, gap at it 3: %s" % (
        opts, nf, len(idx), int(sol.feas.sum()), h.solver_iters[2].tolist(), h.solver_status[2].tolist(),
        ["%.0e" % v for v in h.gap[2]]))
