"""One Monte-Carlo instance of the config-3 record (bench.starship_scvx_record: ICs +-2 %, seed = index) through the ORACLE: the
oracle's guess (oracle/starship_guess.py + oracle/ipm.py) and the first iterations of the oracle's literal SCvx loop at N = 100 --
what a device SCP_FAILED / stalled instance is compared with (VERDICT r04 "next" 1c).

    OMP_NUM_THREADS=4 python tools/starship_instance_oracle.py <instance> [iterations = 3]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_starship_golden import oracle_batch  # noqa: E402
from oracle import scvx_ref  # noqa: E402
from oracle.models import MODELS  # noqa: E402
from oracle.starship_guess import StarshipConstants, starship_initial_guess  # noqa: E402


def main():
    i = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    N = 100
    nom = MODELS["starship"](N).nominal_pp()
    pp = nom * (1 + (0.02 * np.random.default_rng(i).uniform(-1, 1, nom.size) if i else 0.0))

    class K(StarshipConstants):
        pass
    K.r0, K.v0, K.theta0 = pp[0:2], pp[2:4], float(pp[4])
    x, u, p, hs = starship_initial_guess(N, oracle_batch, K)
    print("instance", i, "oracle guess t1, t2 =", p[0], p[1])
    # the bench record normalises the cost of the whole batch with the NOMINAL instance's switch altitude (bench.py)
    hs0 = float(np.load(os.path.join(ROOT, "tests", "golden", "starship_guess_mc.npz"))["hs100"])
    mdl = MODELS["starship"](N, hs0)
    sp_ = scvx_ref.SCvxParameters(N, 100, iters, lam=5e2, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                  eta_lb=1e-8, eta_ub=10.0, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3)
    st, h = scvx_ref.scvx_solve(mdl, sp_, pp=pp, guess=(x, u, p), verbose=True, ipm_opts=dict(max_iter=1000))
    print(st, [r["sub"]["status"] for r in h], [r["sub"]["ipm"]["iters"] for r in h])


if __name__ == "__main__":
    main()
