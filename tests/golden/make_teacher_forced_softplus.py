"""The oracle's literal GuSTO loops with `pen = :softplus` (oracle/gusto_ref.py + oracle/ipm.py::solve_exp) of
tests/test_gusto_gpu.py::test_gusto_softplus_loop_matches_oracle -- quadrotor, N = 16, Nsub = 10, 12 iterations, hom in {500, 50},
the nominal instance and the goal + 2 % instance -- with the REFERENCE, (eta, lambda), optimal value and solution of every
iteration, for the teacher-forced device test (tests/test_teacher_forced_gpu.py).

    python tests/golden/make_teacher_forced_softplus.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import gusto_ref  # noqa: E402
from oracle.models import MODELS  # noqa: E402

N, NSUB, ITERS = 16, 10, 12


def main():
    mdl = MODELS["quadrotor"]()
    pp2 = mdl.nominal_pp().copy(); pp2[6:9] *= 1.02
    cases = [(hom, pp) for hom in (500.0, 50.0) for pp in (mdl.nominal_pp(), pp2)]
    out = dict(hom=[], pp=[], ref_xd=[], ref_ud=[], ref_p=[], sol_xd=[], sol_ud=[], sol_p=[], eta=[], lam=[], L_aug=[], pcost=[],
               J_aug=[], rho=[], accept=[], status=[])
    for hom, pp in cases:
        op = gusto_ref.quadrotor_test_parameters(N, NSUB, ITERS)
        op.pen, op.hom = "softplus", hom
        st, oh = gusto_ref.gusto_solve("quadrotor", op, pp=pp)
        assert st == "SCP_SOLVED" and len(oh) == ITERS
        out["hom"].append(hom); out["pp"].append(pp); out["status"].append(st)
        for key, f in (("ref_xd", lambda r: r["ref"].xd), ("ref_ud", lambda r: r["ref"].ud), ("ref_p", lambda r: r["ref"].p),
                       ("sol_xd", lambda r: r["sub"]["x"]), ("sol_ud", lambda r: r["sub"]["u"]), ("sol_p", lambda r: r["sub"]["p"]),
                       ("eta", lambda r: r["eta"]), ("lam", lambda r: r["lam"]), ("L_aug", lambda r: r["sub"]["L_aug"]),
                       ("pcost", lambda r: r["sub"]["pcost"]), ("J_aug", lambda r: r["J_aug"]), ("rho", lambda r: r.get("rho", np.nan)),
                       ("accept", lambda r: int(r.get("accept", -1)))):
            out[key].append(np.array([f(r) for r in oh]))
        print(hom, st, oh[-1]["J_aug"])
    np.savez_compressed(os.path.join(HERE, "teacher_forced_gusto_softplus_quadrotor_N16.npz"), N=N, Nsub=NSUB, iter_max=ITERS,
                        **{k: np.array(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
