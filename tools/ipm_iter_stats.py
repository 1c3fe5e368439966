"""Per-PTR-iteration statistics of the structured IPM (K3 launch time, iterations, exit status) on the bench workload:
    python tools/ipm_iter_stats.py [model = rocket_landing] [B = 256] [solver opts, e.g. warm=0,reg=5e-11]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
import bench
model = sys.argv[1] if len(sys.argv) > 1 else "rocket_landing"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
opts = {}
if len(sys.argv) > 3 and sys.argv[3]:
    for kv in sys.argv[3].split(","):
        k, v = kv.split("=")
        opts[k] = float(v)
_, N, Nsub, iters, _ = bench.WORKLOADS[model]
traj = pkg.TrajectoryProblem(model)
pp = bench.mc_pp(traj.mdl, B, 0)
pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0, solver_opts=opts)
pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
pkg.PTR.upload(pbm, pp, None, True)
ms = []
for k in range(iters):
    pkg.PTR.kernel_timing(pbm, reset=True)
    pkg.PTR.iterate(pbm)
    ksec, kcnt = pkg.PTR.kernel_timing(pbm, reset=True)
    ms.append(1e3 * ksec[2])
sol, h = pkg.PTR.collect(pbm, B)
print("%s B=%d opts=%s: K3 total %.1f ms" % (model, B, opts, sum(ms)))
for k in range(h.solver_iters.shape[0]):
    it = h.solver_iters[k]
    st = np.bincount(h.solver_status[k].astype(int), minlength=4)
    print("PTR it %2d: K3 %6.1f ms | ipm iters mean %.1f min %d med %d p90 %d p99 %d max %d | status opt/almost/itlim/num %s | J_vc med %.2e dev med %.2e"
          % (k + 1, ms[k], it.mean(), it.min(), np.median(it), np.percentile(it, 90), np.percentile(it, 99), it.max(), st,
             np.median(h.J_vc[k]), np.median(h.deviation[k])))
print("status", {s: sol.status.count(s) for s in set(sol.status)}, "feasible", float(sol.feas.mean()))
last = h.solver_iters[-1]
worst = np.argsort(-last)[:6]
print("slowest problems of the last launch (index: iterations of launches 7.., status of the last):",
      {int(b): (h.solver_iters[6:, b].astype(int).tolist(), int(h.solver_status[-1, b])) for b in worst})
import os
if os.environ.get("K3_ITERS_DUMP"):      # [iter_max][B] IPM iterations per launch and problem + the launch times (input of tools/k3_packing_sim.py)
    np.savez_compressed(os.environ["K3_ITERS_DUMP"], iters=h.solver_iters.astype(np.int16), k3_ms=np.array(ms))
pbm.close()
