"""K5 at one workgroup per problem on the config-3 program (Starship SCvx N = 100, n = 7 623 LP): the 30 subproblems of the oracle's
record as one batch -- seconds per launch and IPM iterations, for tuning sweeps (SCP_CONIC_LONG_ITEM, ...).
    python tools/k5_starship_probe.py [repeat = 2]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
XO = {k[4:].lower(): float(v) if '.' in v or 'e' in v else int(v) for k, v in os.environ.items() if k.startswith('K5O_')}     # extra solver options: K5O_REF_TOL=1e-9 ...
if len(sys.argv) > 3 and sys.argv[3] == "freeflyer":      # the config-5 program instead: free-flyer GuSTO N = 200 (n = 10 402), 4 oracle subproblems tiled
    g = np.load(os.path.join(ROOT, "tests", "golden", "freeflyer_gusto_N200.npz"))
    N, Nsub, K = int(g["N"]), int(g["Nsub"]), int(g["eta"].size)
    rep = int(sys.argv[1]); B = int(sys.argv[2])
    traj = pkg.TrajectoryProblem("freeflyer")
    gp = pkg.GuSTO.Parameters(N=N, Nsub=Nsub, iter_max=1, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.5, beta_sh=2.0, beta_gr=2.0, gamma_fail=5.0,
                              eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=16, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
    pbm = pkg.GuSTO.create(gp, traj, batch_capacity=B)
    idx = np.arange(B) % K
    out = []
    for _ in range(rep):
        r = pbm.sub.solve(g["ref_xd"][idx], g["ref_ud"][idx], g["ref_p"][idx], pp=np.tile(g["pp"], (B, 1)), scal=np.stack([g["eta"][idx], g["lam"][idx]], axis=1), **XO)
        rel = np.abs(r["pcost"] - g["L_aug"][idx]) / np.maximum(1.0, np.abs(g["L_aug"][idx]))
        out.append(dict(seconds=r["seconds"], ipm_mean=float(r["iters"].mean()), ipm_max=int(r["iters"].max()), rel_max=float(rel.max()), safe=bool((r["status"] <= 1).all())))
    st = pbm.sub.stats()
    pbm.close()
    print(json.dumps(dict(program="freeflyer N=200", env={k: v for k, v in os.environ.items() if k.startswith("SCP_CONIC")}, batch=B, runs=out, levels=st["levels"], nnzL=st["nnzL"], madds=st["factor_madds"])))
    raise SystemExit
g = np.load(os.path.join(ROOT, "tests", "golden", "starship_N100_scvx_long_t21.npz"))
K = int(g["iters"])
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else K
trs = pkg.TrajectoryProblem("starship", hs=float(g["hs"]))
pars = pkg.SCvx.Parameters(N=int(g["N"]), Nsub=int(g["Nsub"]), iter_max=1, lam=5e2, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0,
                           eta_init=1.0, eta_lb=1e-8, eta_ub=10.0, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3)
pbm = pkg.SCvx.create(pars, trs, batch_capacity=B)
idx = np.arange(B) % K
out = []
for _ in range(rep):
    r = pbm.sub.solve(g["all_ref_xd"][idx], g["all_ref_ud"][idx], g["all_ref_p"][idx], pp=np.tile(trs.mdl.nominal_pp(), (B, 1)), scal=g["eta"][idx][:, None], max_iter=1000, **XO)
    rel = np.abs(r["pcost"] - g["L_aug"][idx]) / np.maximum(1.0, np.abs(g["L_aug"][idx]))
    out.append(dict(seconds=r["seconds"], ipm_mean=float(r["iters"].mean()), ipm_max=int(r["iters"].max()), rel_max=float(rel.max()), safe=bool((r["status"] <= 1).all())))
st = pbm.sub.stats()
pbm.close()
print(json.dumps(dict(env={k: v for k, v in os.environ.items() if k.startswith("SCP_CONIC")}, batch=B, runs=out, levels=st["levels"], nnzL=st["nnzL"], madds=st["factor_madds"])))
