"""The oracle's literal SCvx loop on BASELINE.json configs[2] at its stated size (Starship landing flip, N = 100, Nsub = 100,
reference test parameters and stopping rule) for up to 30 iterations from the guess stored in starship_N100_scvx3.npz -- the
cost / trust-region / accept record the device loop's nominal instance is compared with (bench.py, starship_scvx_record).

    OMP_NUM_THREADS=4 python tests/golden/make_starship_n100_long.py [iters]       # ~2-3 min per iteration (oracle/ipm.py)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from oracle import scvx_ref  # noqa: E402
from oracle.models import MODELS  # noqa: E402


REF_ITERS = [0, 10, 25, 29]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    g = np.load(os.path.join(HERE, "starship_N100_scvx3.npz"))
    N, Nsub, hs = int(g["N"]), int(g["Nsub"]), float(g["hs"])
    mdl = MODELS["starship"](N, hs)
    sp_ = scvx_ref.SCvxParameters(N, Nsub, iters, lam=5e2, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                  eta_lb=1e-8, eta_ub=10.0, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3)
    st, h = scvx_ref.scvx_solve(mdl, sp_, guess=(g["guess_x"], g["guess_u"], g["guess_p"]), verbose=True)
    fin = h[-1]["sol"]
    np.savez_compressed(os.path.join(HERE, "starship_N100_scvx_long.npz"), N=N, Nsub=Nsub, hs=hs, status=st, iters=len(h),
                        eta=[r["eta"] for r in h], L=[r["sub"]["L"] for r in h], L_aug=[r["sub"]["L_aug"] for r in h],
                        J_sol=[r.get("J_sol", np.nan) for r in h], accept=[bool(r.get("accept", False)) for r in h],
                        feas=[r["sol"].feas for r in h], ipm_status=[r["sub"]["status"] for r in h], xd=fin.xd, ud=fin.ud, p=fin.p,
                        # the reference trajectories of a few subproblems (first, mid-run, the most degenerate, last): the
                        # product's templates + solver are checked on exactly these programs (tests/test_template_cpu.py)
                        ref_iters=REF_ITERS, ref_xd=[h[k]["ref"].xd for k in REF_ITERS], ref_ud=[h[k]["ref"].ud for k in REF_ITERS],
                        ref_p=[h[k]["ref"].p for k in REF_ITERS],
                        # round 5: the reference of EVERY iteration (teacher-forced device subproblems, tests/test_starship_gpu.py)
                        all_ref_xd=[r["ref"].xd for r in h], all_ref_ud=[r["ref"].ud for r in h], all_ref_p=[r["ref"].p for r in h],
                        rho=[r.get("rho", np.nan) for r in h], J_ref=[r.get("J_ref", np.nan) for r in h],
                        ipm_iters=[r["sub"]["ipm"]["iters"] for r in h])
    print(st, len(h))


if __name__ == "__main__":
    main()
