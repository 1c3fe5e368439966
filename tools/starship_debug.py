"""GPU debug aid: Starship PTR loop from the golden guess with several solver option sets."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
g = np.load(os.path.join(ROOT, "tests", "golden", "starship_N31.npz"))
for opts in [dict(), dict(nref=10), dict(reg=1e-9), dict(reg=1e-7), dict(reg=1e-7, nref=10), dict(reg=1e-9, nref=10)]:
    traj = pkg.TrajectoryProblem("starship", hs=float(g["hs"]))
    pars = pkg.PTR.Parameters(N=31, Nsub=100, iter_max=8, wvc=1e3, wtr=0.1, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3, solver_opts=opts)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=1)
    warm = tuple(g[k][None] for k in ("guess_x", "guess_u", "guess_p"))
    sol, hist = pkg.PTR.solve(pbm, traj.mdl.nominal_pp()[None], warm=warm)
    print(opts, sol.status[0], sol.iterations[0], flush=True)
    for k in range(sol.iterations[0]):
        print("  k=%d st %d it %d J_aug %.9e (golden %s) gap %.1e pres %.1e dres %.1e" % (
            k + 1, hist.solver_status[k, 0], hist.solver_iters[k, 0], hist.J_aug[k, 0],
            "%.9e" % g["ptr_J_aug"][k] if k < 6 else "-", hist.gap[k, 0], hist.pres[k, 0], hist.dres[k, 0]), flush=True)
    pbm.close()
