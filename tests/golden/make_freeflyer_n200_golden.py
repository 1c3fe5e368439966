"""BASELINE.json configs[4] at its stated size: the oracle's literal GuSTO loop (quadratic penalty, reference test parameters
freeflyer/tests.jl:84-140) on the free-flyer at N = 200, Nsub = 15 -- its first iterations from the reference's guess, with the
reference trajectories of every subproblem so that the product's templates + solver can be checked on the same programs.

    OMP_NUM_THREADS=8 python tests/golden/make_freeflyer_n200_golden.py [iters = 4]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import gusto_ref  # noqa: E402
from oracle.models import MODELS  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    N = 200
    mdl = MODELS["freeflyer"](N)
    gp = gusto_ref.GuSTOParameters(N, 15, iters, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.5, beta_sh=2.0, beta_gr=2.0,
                                   gamma_fail=5.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=16, eps_abs=0.0,
                                   eps_rel=0.0, feas_tol=1e-3)
    t0 = time.time()
    st, h = gusto_ref.gusto_solve(mdl, gp, verbose=True)
    print(st, len(h), "%.0f s" % (time.time() - t0))
    fin = h[-1]["sol"]
    np.savez_compressed(os.path.join(HERE, "freeflyer_gusto_N200.npz"), N=N, Nsub=15, status=st, iters=len(h),
                        eta=[r["eta"] for r in h], lam=[r["lam"] for r in h], L=[r["sub"]["L"] for r in h],
                        L_aug=[r["sub"]["L_aug"] for r in h], L_st=[r["sub"]["L_st"] for r in h], L_tr=[r["sub"]["L_tr"] for r in h],
                        J_aug=[r.get("J_aug", np.nan) for r in h], accept=[bool(r.get("accept", False)) for r in h],
                        feas=[r["sol"].feas for r in h], ipm_status=[r["sub"]["status"] for r in h],
                        ref_xd=[r["ref"].xd for r in h], ref_ud=[r["ref"].ud for r in h], ref_p=[r["ref"].p for r in h],
                        xd=fin.xd, ud=fin.ud, p=fin.p, pp=mdl.nominal_pp())


if __name__ == "__main__":
    main()
