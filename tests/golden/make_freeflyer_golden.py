"""Golden fixture for the 6-DoF free-flyer at the reference's own SCvx test configuration
(test/examples/freeflyer/tests.jl:25-80: N = 50, Nsub = 15, iter_max = 15, lambda = 1e3, rho = (0, 0.1, 0.7), beta = 2,
eta in [1e-6, 10], eta_init = 1, feas_tol = 1e-3) and GuSTO test configuration (:84-140), produced by the ORACLE's literal
SCvx / GuSTO loops (oracle/scvx_ref.py, gusto_ref.py with the cone indicators of define_conic_constraint!) on the
full problem (p = [t_f; delta], np = 301; oracle/models.py Freeflyer(N)).

    python tests/golden/make_freeflyer_golden.py

The reference's test asserts only `sol.status == SCP_SOLVED`; the oracle loop reproduces that and the fixture records
what it converged to, for the device-side subproblem of this model (not built yet: np depends on N)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import gusto_ref, scvx_ref  # noqa: E402
from oracle.models import MODELS  # noqa: E402


def main():
    N = 50
    mdl = MODELS["freeflyer"](N)
    pars = scvx_ref.SCvxParameters(N, 15, 15, lam=1e3, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                   eta_lb=1e-6, eta_ub=10.0, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
    st, hist = scvx_ref.scvx_solve(mdl, pars, verbose=True)
    fin = hist[-1]["sol"]
    np.savez_compressed(os.path.join(HERE, "freeflyer_scvx_N50.npz"), N=N, Nsub=15, status=st, iters=len(hist),
                        eta=[h["eta"] for h in hist], L=[h["sub"]["L"] for h in hist],
                        J_sol=[h.get("J_sol", np.nan) for h in hist], accept=[bool(h.get("accept", False)) for h in hist],
                        feas=[h["sol"].feas for h in hist], xd=fin.xd, ud=fin.ud, p=fin.p, pp=mdl.nominal_pp())
    print(st, len(hist))
    # GuSTO, freeflyer/tests.jl:84-140 (lambda_init = 1e4, rho = (0.1, 0.5), eta_init = 1, mu = 0.8 from iteration 16, pen = :quad)
    gp = gusto_ref.GuSTOParameters(N, 15, 15, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.5, beta_sh=2.0, beta_gr=2.0,
                                   gamma_fail=5.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=16, eps_abs=0.0,
                                   eps_rel=0.0, feas_tol=1e-3)
    st, hist = gusto_ref.gusto_solve(mdl, gp, verbose=True)
    fin = hist[-1]["sol"]
    np.savez_compressed(os.path.join(HERE, "freeflyer_gusto_N50.npz"), N=N, Nsub=15, status=st, iters=len(hist),
                        eta=[h["eta"] for h in hist], lam=[h["lam"] for h in hist], L=[h["sub"]["L"] for h in hist],
                        J_aug=[h.get("J_aug", np.nan) for h in hist], accept=[bool(h.get("accept", False)) for h in hist],
                        feas=[h["sol"].feas for h in hist], xd=fin.xd, ud=fin.ud, p=fin.p, pp=mdl.nominal_pp())
    print(st, len(hist))


if __name__ == "__main__":
    main()
