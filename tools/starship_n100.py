"""BASELINE.json configs[2] at its stated size: Starship landing flip, N = 100, Nsub = 100, SCvx on one MI355X, from the
reference's own initial guess (bang-bang flip + convex terminal descent, solved as one batch of 100 programs on the
device), with the reference's SCvx test parameters (starship_flip/tests.jl:77-98).  Writes one JSON record.
usage: python tools/starship_n100.py [iter_max] [batch] [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
out = sys.argv[3] if len(sys.argv) > 3 else None
N, Nsub = 100, 100
t0 = time.perf_counter()
mdl0 = pkg.REGISTRY["starship"]()
x, u, p = mdl0.reference_guess(N)
t_guess = time.perf_counter() - t0
traj = pkg.TrajectoryProblem("starship", hs=float(mdl0.hs))
pars = pkg.SCvx.Parameters(N=N, Nsub=Nsub, iter_max=iters, lam=5e2, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0,
                           eta_init=1.0, eta_lb=1e-8, eta_ub=10.0, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3)
t0 = time.perf_counter()
pbm = pkg.SCvx.create(pars, traj, batch_capacity=B)
t_create = time.perf_counter() - t0
pp = np.stack([traj.mdl.nominal_pp()] * B)
guess = tuple(np.stack([a] * B) for a in (x, u, p))
t0 = time.perf_counter()
sol, hist = pkg.SCvx.solve(pbm, pp, guess=guess)
dt = time.perf_counter() - t0
T = pbm.template
rec = dict(workload="starship SCvx N=%d Nsub=%d (reference test parameters), %d instances of the nominal problem, reference guess" % (N, Nsub, B),
           conic_program=dict(n=int(T.n), p=int(T.p), m=int(T.m)), guess_seconds=t_guess, guess_t1_t2=[float(p[0]), float(p[1])],
           create_seconds=t_create, solve_seconds=dt, iterations=int(sol.iterations[0]), status=sol.status[0],
           scp_iterations_per_s=float(sol.iterations.sum()) / dt, dynamically_feasible=bool(sol.feas[0]),
           all_instances_identical=bool(all(np.array_equal(sol.xd[0], sol.xd[b]) for b in range(B))),
           eta=[float(v) for v in hist["eta"][:sol.iterations[0], 0]], accepted=[int(v) for v in hist["accepted"][:sol.iterations[0], 0]],
           L=[float(v) for v in hist["L"][:sol.iterations[0], 0]], J_sol=[float(v) for v in hist["J_sol"][:sol.iterations[0], 0]],
           solver_iters=[int(v) for v in hist["solver_iters"][:sol.iterations[0], 0]],
           solver_status=[int(v) for v in hist["solver_status"][:sol.iterations[0], 0]],
           final_t1_t2=[float(sol.p[0, 0]), float(sol.p[0, 1])], max_scaled_defect=float(np.abs(sol.defect[0] * pbm.scale.iSx[None, :]).max()))
pbm.close()
s = json.dumps(rec)
print(s)
if out:
    open(out, "w").write(s + "\n")
