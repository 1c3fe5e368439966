import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
model = sys.argv[1]; N = int(sys.argv[2]); B = int(sys.argv[3]); iters = int(sys.argv[4]) if len(sys.argv) > 4 else 15
traj = pkg.TrajectoryProblem(model)
pars = pkg.PTR.Parameters(N=N, Nsub=15, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0)
pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
pp = []
for b in range(B):
    rng = np.random.default_rng(b)
    q = traj.mdl.nominal_pp()
    pp.append(q * (1 + (0.1 * rng.uniform(-1, 1, q.size) if b > 0 else 0)))
pp = np.stack(pp)
t0 = time.time()
sol, h = pkg.PTR.solve(pbm, pp)
t1 = time.time()
print("wall %.3fs  -> %.1f SCP it/s" % (t1 - t0, B * iters / (t1 - t0)))
sec, cnt = pkg.PTR.kernel_timing(pbm)
print("kernel seconds", sec, cnt)
for it in range(iters):
    print("it %2d J % .6e Jtr %.3e Jvc %.3e feas %s st %s ipm_it %s gap %.1e pres %.1e dres %.1e dev %.2e" % (
        it + 1, h.J[it, 0], h.J_tr[it, 0], h.J_vc[it, 0], h.feas[it, 0], h.solver_status[it, 0], h.solver_iters[it, 0],
        h.gap[it, 0], h.pres[it, 0], h.dres[it, 0], h.deviation[it, 0]))
print("statuses", set(sol.status), "feas frac", sol.feas.mean(), "ipm iters mean per it", h.solver_iters.mean(axis=1))
print("max |iSx defect|", np.abs(sol.defect / pbm.scale.Sx).max(axis=(1, 2))[:8])
print("solver status counts", np.unique(h.solver_status, return_counts=True))
import ctypes
t = (ctypes.c_longlong * 8)()
pkg._lib.lib().scp_debug_get_ipm_profile(pbm.handle, 0, t)
names = ["G", "GT", "factor", "rhs+fwd", "bwd", "aux", "-", "total"]
print("last IPM launch phase ms:", {n: round(v / 1e5, 2) for n, v in zip(names, t)})
