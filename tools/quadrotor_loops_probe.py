import sys, json, time
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package()
import bench
import numpy as np
traj = pkg.TrajectoryProblem("quadrotor"); mdl = traj.mdl
pp = bench.mc_pp(mdl, 1024, 0)
gp = pkg.GuSTO.Parameters(N=30, Nsub=15, iter_max=6, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.9, beta_sh=2.0, beta_gr=2.0, gamma_fail=5.0, eta_init=10.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=6, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
pbm = pkg.GuSTO.create(gp, traj, batch_capacity=1024)
for rep in range(2):
    t0 = time.perf_counter(); sol, hist = pkg.GuSTO.solve(pbm, pp); dt = time.perf_counter() - t0
    st = pbm.sub.stats()
    print("gusto 1024: %.2f s, %.0f it/s" % (dt, sol.iterations.sum() / dt), {k: st[k] for k in ("solves", "fallback_solves", "fallback_rescued", "levels")})
pbm.close()
sp = pkg.SCvx.Parameters(N=30, Nsub=15, iter_max=6, lam=30.0, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
pbm = pkg.SCvx.create(sp, traj, batch_capacity=1024)
for rep in range(2):
    t0 = time.perf_counter(); sol, hist = pkg.SCvx.solve(pbm, pp); dt = time.perf_counter() - t0
    print("scvx 1024: %.2f s, %.0f it/s" % (dt, sol.iterations.sum() / dt))
pbm.close()
