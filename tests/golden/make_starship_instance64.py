"""The config-3 subproblems on which the device solver once ended NUMERICAL_ERROR: Monte-Carlo instance 64 of
the Starship SCvx batch at N = 100 (bench.starship_scvx_record: ICs +-2 %, seed 64), iterations 1-3 of the ORACLE's literal loop from
the oracle's own guess -- the reference trajectory, trust-region radius, optimal value and status of every subproblem (iteration 2, eta = 2,
is the degenerate LP).  tests/test_starship_gpu.py solves them teacher-forced on the device.

    python tests/golden/make_starship_instance64.py [instance = 64] [iterations = 3]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from make_starship_golden import oracle_batch  # noqa: E402
from oracle import scvx_ref  # noqa: E402
from oracle.models import MODELS  # noqa: E402
from oracle.starship_guess import StarshipConstants, starship_initial_guess  # noqa: E402

i = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = 100
nom = MODELS["starship"](N).nominal_pp()
pp = nom * (1 + (0.02 * np.random.default_rng(i).uniform(-1, 1, nom.size) if i else 0.0))


class K(StarshipConstants):
    pass


K.r0, K.v0, K.theta0 = pp[0:2], pp[2:4], float(pp[4])
x, u, p, hs = starship_initial_guess(N, oracle_batch, K)
hs0 = float(np.load(os.path.join(HERE, "starship_guess_mc.npz"))["hs100"])
mdl = MODELS["starship"](N, hs0)
sp_ = scvx_ref.SCvxParameters(N, 100, iters, lam=5e2, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                              eta_lb=1e-8, eta_ub=10.0, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3)
st, h = scvx_ref.scvx_solve(mdl, sp_, pp=pp, guess=(x, u, p), verbose=True, ipm_opts=dict(max_iter=1000))
np.savez_compressed(os.path.join(HERE, "starship_N100_scvx_i%d.npz" % i), instance=i, N=N, Nsub=100, hs=hs0, pp=pp, guess_p=p,
                    eta=[r["eta"] for r in h], L=[r["sub"]["L"] for r in h], L_aug=[r["sub"]["L_aug"] for r in h],
                    ipm_status=[r["sub"]["status"] for r in h], ipm_iters=[r["sub"]["ipm"]["iters"] for r in h],
                    accept=[bool(r.get("accept", False)) for r in h],
                    ref_xd=[r["ref"].xd for r in h], ref_ud=[r["ref"].ud for r in h], ref_p=[r["ref"].p for r in h])
print(st, [r["sub"]["status"] for r in h])
