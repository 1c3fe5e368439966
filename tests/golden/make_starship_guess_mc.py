"""The reference's Starship guess (oracle/starship_guess.py with the ORACLE's interior-point solver, oracle/ipm.py) on the five
Monte-Carlo instances of tests/test_starship_gpu.py::test_reference_guess_entirely_on_the_device_per_instance (nominal + four
initial conditions perturbed by 2 %, seed = instance) at N = 31, and on the nominal instance at N = 100: t1, the FIRST FEASIBLE
descent duration t2, the switch state and the guess itself -- what the device guess kernels (csrc/starship_guess.hpp) must pick.

    python tests/golden/make_starship_guess_mc.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from make_starship_golden import oracle_batch  # noqa: E402
from oracle.models import MODELS  # noqa: E402
from oracle.starship_guess import StarshipConstants, starship_initial_guess  # noqa: E402


def guess(N, pp):
    class K(StarshipConstants):
        pass
    K.r0, K.v0, K.theta0 = np.asarray(pp[0:2], float), np.asarray(pp[2:4], float), float(pp[4])
    return starship_initial_guess(N, oracle_batch, K)


def main():
    nom = MODELS["starship"](31).nominal_pp()
    pp = np.stack([nom * (1 + (0.02 * np.random.default_rng(i).uniform(-1, 1, nom.size) if i else 0.0)) for i in range(5)])
    out = [guess(31, q) for q in pp]
    x100, u100, p100, hs100 = guess(100, nom)
    np.savez_compressed(os.path.join(HERE, "starship_guess_mc.npz"), pp=pp, x=np.stack([o[0] for o in out]), u=np.stack([o[1] for o in out]),
                        p=np.stack([o[2] for o in out]), hs=np.array([o[3] for o in out]), x100=x100, u100=u100, p100=p100, hs100=hs100)
    print("t2 at N = 31:", [o[2][1] for o in out], " at N = 100:", p100[1])


if __name__ == "__main__":
    main()
