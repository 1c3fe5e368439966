"""Golden fixture for BASELINE.json configs[2] AT ITS STATED SIZE (Starship landing flip, SCvx, N = 100, Nsub = 100): the
reference's initial guess (product code starship_guess.py with the ORACLE's interior-point solver behind it) and the first
three iterations of the oracle's literal SCvx loop from it (reference test parameters starship_flip/tests.jl:77-98).

    python tests/golden/make_starship_n100_golden.py        # ~10 min: n = 7 623 LPs through oracle/ipm.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import __graft_entry__ as graft  # noqa: E402
from oracle import scvx_ref  # noqa: E402
from oracle.models import MODELS  # noqa: E402

graft.load_package()
from make_starship_golden import oracle_batch  # noqa: E402
from oracle.starship_guess import starship_initial_guess  # noqa: E402


def main():
    N, Nsub, iters = 100, 100, 3
    x, u, p, hs = starship_initial_guess(N, oracle_batch)
    mdl = MODELS["starship"](N, hs)
    sp_ = scvx_ref.SCvxParameters(N, Nsub, iters, lam=5e2, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                  eta_lb=1e-8, eta_ub=10.0, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3)
    st, h = scvx_ref.scvx_solve(mdl, sp_, guess=(x, u, p), verbose=True)
    fin = h[-1]["sol"]
    np.savez_compressed(os.path.join(HERE, "starship_N100_scvx3.npz"), N=N, Nsub=Nsub, hs=hs, guess_x=x, guess_u=u, guess_p=p,
                        status=st, iters=len(h), eta=[r["eta"] for r in h], L=[r["sub"]["L"] for r in h],
                        L_aug=[r["sub"]["L_aug"] for r in h], J_sol=[r.get("J_sol", np.nan) for r in h],
                        accept=[bool(r.get("accept", False)) for r in h], feas=[r["sol"].feas for r in h],
                        ipm_status=[r["sub"]["status"] for r in h], xd=fin.xd, ud=fin.ud, p=fin.p)
    print(st, len(h))


if __name__ == "__main__":
    main()
