"""CPU twin of tests/test_teacher_forced_gpu.py: the product's conic TEMPLATES (scptoolbox.jl_amd/subproblem.py), filled with the
oracle's per-iteration references of tests/golden/teacher_forced_*_quadrotor_N30.npz and solved by the HOST build of the product's
conic solver (oracle/conic_host.py), against the optimal values of the oracle's literal programs: 1e-6 relative on every
subproblem of a sample of instances (the full 64 x 6 run on the device)."""
import os

import numpy as np
import pytest

from oracle import conic_host, ptr_ref
from oracle.models import MODELS
from template_util import make_src, template_matrices

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SAMPLE = (0, 7, 21, 40, 63)      # instance 21 escalates lambda in the GuSTO loop (VERDICT r04 cites it)


def _setup(pkg):
    mdl = MODELS["quadrotor"]()
    mr = pkg.subproblem.ModelRows(pkg.REGISTRY["quadrotor"]())
    scale = ptr_ref.Scaling(*mdl.bbox())
    pars = ptr_ref.PTRParameters(30, 15, 3, 1e3, 0.1, 0, 0, 1e-3)
    return mdl, mr, scale, pars


@pytest.mark.parametrize("algo", ["scvx", "gusto"])
def test_host_solver_on_the_oracles_references(pkg, orc, algo):
    g = np.load(os.path.join(GOLD, "teacher_forced_%s_quadrotor_N30.npz" % algo))
    mdl, mr, scale, pars = _setup(pkg)
    T = pkg.subproblem.build_scvx(mr, 30, scale, 30.0) if algo == "scvx" else pkg.subproblem.build_gusto(mr, 30, scale)
    worst = 0.0
    for b in SAMPLE:
        for k in np.flatnonzero(g["valid"][b]):
            ref = ptr_ref.discretize(mdl, pars, scale, g["ref_xd"][b, k], g["ref_ud"][b, k], g["ref_p"][b, k])
            scal = float(g["eta"][b, k]) if algo == "scvx" else [float(g["eta"][b, k]), float(g["lam"][b, k])]
            v, G, A, P = template_matrices(T, make_src(T, mdl, ref, g["pp"][b], scal))
            r = conic_host._solve(v["c"], G, v["h"], T.l, T.q, A, v["b"], P=P)
            assert r["status"] in (0, 1), (b, k, r["status"])
            rel = abs(r["pcost"] + T.cost_const - g["pcost"][b, k]) / max(1.0, abs(g["pcost"][b, k]))
            worst = max(worst, rel)
            assert rel <= 1e-6, (algo, b, int(k), r["pcost"] + T.cost_const, float(g["pcost"][b, k]))
    print(algo, "worst relative difference of the optimal value", worst)


@pytest.mark.parametrize("algo", ["scvx", "gusto"])
def test_host_solver_on_the_oracles_freeflyer_references(pkg, orc, algo):
    """The free-flyer at N = 50 (tests/golden/teacher_forced_*_freeflyer_N50.npz): a sample of the oracle loops' subproblems through the
    product's templates and the host build of the product's solver, optimal value 1e-6.  The GuSTO sample holds the subproblems this
    comparison was added for: the first one of instances 0 and 2 (ALMOST_OPTIMAL, 7e-5 / 2e-5 off with ECOS's dynamic-regularisation
    constant 2e-7: a replaced pivot put 5e6 into the factor and the next pivots overflowed) and the third one of every instance
    (OPTIMAL by ECOS's criteria on both sides, but with a static regularisation of 1e-8 the refinement stalled on the equality block,
    the primal residual sat at 4e-9 and, times multipliers of 13, that was 6.4e-6 of the optimal value) -- conic_symbolic.hpp auto_reg."""
    from template_util import OracleRows
    g = np.load(os.path.join(GOLD, "teacher_forced_%s_freeflyer_N50.npz" % algo))
    N = int(g["N"])
    mdl = MODELS["freeflyer"](N)
    scale = ptr_ref.Scaling(*mdl.bbox())
    pars = ptr_ref.PTRParameters(N, int(g["Nsub"]), 3, 1e3, 0.1, 0, 0, 1e-3)
    mr = OracleRows(mdl, N)
    T = pkg.subproblem.build_scvx(mr, N, scale, 1e3) if algo == "scvx" else pkg.subproblem.build_gusto(mr, N, scale)
    worst = 0.0
    for b, k in ((0, 0), (0, 2), (2, 0), (2, 2), (5, 2), (3, 7), (7, 13)):
        if not g["valid"][b, k]:
            continue
        ref = ptr_ref.discretize(mdl, pars, scale, g["ref_xd"][b, k], g["ref_ud"][b, k], g["ref_p"][b, k])
        scal = float(g["eta"][b, k]) if algo == "scvx" else [float(g["eta"][b, k]), float(g["lam"][b, k])]
        v, G, A, P = template_matrices(T, make_src(T, mdl, ref, g["pp"][b], scal, Fcols=[0]))
        r = conic_host._solve(v["c"], G, v["h"], T.l, T.q, A, v["b"], P=P)
        assert r["status"] in (0, 1), (b, k, r["status"])
        rel = abs(r["pcost"] + T.cost_const - g["pcost"][b, k]) / max(1.0, abs(g["pcost"][b, k]))
        worst = max(worst, rel)
        assert rel <= 1e-6, (algo, b, k, r["pcost"] + T.cost_const, float(g["pcost"][b, k]))
    print(algo, "free-flyer: worst relative difference of the optimal value", worst)
