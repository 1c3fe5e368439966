"""Pins for the oracle's conic solver (oracle/ipm.py, restating the ECOS algorithm class) by
solver-independent certificates: agreement with scipy's HiGHS on LPs, closed-form SOCP / QP optima,
KKT residuals and duality gap of a PTR subproblem."""
import numpy as np
import scipy.sparse as sp
from scipy.optimize import linprog


def test_lp_matches_highs():
    from oracle import ipm
    rng = np.random.default_rng(0)
    for trial in range(3):
        n, m = 30, 10
        A = rng.standard_normal((m, n)); x0 = rng.uniform(0.1, 1, n); b = A @ x0; c = rng.uniform(0.1, 1, n)
        res = linprog(c, A_eq=A, b_eq=b, bounds=[(0, None)] * n, method="highs")
        r = ipm.solve(c, -sp.eye(n), np.zeros(n), n, [], A=A, b=b)
        assert r["status"] == ipm.OPTIMAL
        assert abs(r["pcost"] - res.fun) <= 1e-7 * max(1, abs(res.fun))


def test_socp_and_qp_closed_forms():
    from oracle import ipm
    rng = np.random.default_rng(1)
    c = rng.standard_normal(5)
    G = sp.vstack([sp.csc_matrix((1, 5)), -sp.eye(5)]); h = np.concatenate([[1.0], np.zeros(5)])
    r = ipm.solve(c, G, h, 0, [6])                      # min c'x s.t. ||x|| <= 1  ->  x = -c/||c||
    assert r["status"] == ipm.OPTIMAL and np.abs(r["x"] + c / np.linalg.norm(c)).max() < 1e-7
    a = np.array([2.0, -1.0, 0.5])                      # projection of a onto {x >= 0, ||x|| <= 1}
    G = sp.vstack([-sp.eye(3), sp.csc_matrix((1, 3)), -sp.eye(3)]); h = np.concatenate([np.zeros(3), [1.0], np.zeros(3)])
    r = ipm.solve(-a, G, h, 3, [4], P=sp.eye(3), abstol=1e-12, reltol=1e-12, feastol=1e-10)
    xp = np.maximum(a, 0); xp = xp / max(1, np.linalg.norm(xp))
    assert r["status"] == ipm.OPTIMAL and np.abs(r["x"] - xp).max() < 1e-5


def test_ptr_subproblem_certificate():
    """The literal PTR conic program (oracle/ptr_ref.py) is solved to a KKT certificate."""
    from oracle import ptr_ref
    from oracle.models import MODELS
    mdl = MODELS["quadrotor"]()
    pars = ptr_ref.PTRParameters(10, 8, 3, 1e3, 0.1, 0, 0, 1e-3)
    scale = ptr_ref.Scaling(*mdl.bbox())
    x, u, p = mdl.guess(10, mdl.nominal_pp())
    ref = ptr_ref.discretize(mdl, pars, scale, x, u, p)
    sub = ptr_ref.solve_subproblem(mdl, pars, scale, ref, mdl.nominal_pp())
    r = sub["ipm"]
    assert sub["status"] == "OPTIMAL"
    assert r["pres"] <= 1e-8 and r["dres"] <= 1e-8 and (r["gap"] <= 1e-8 or r["relgap"] <= 1e-8)
    assert abs(r["pcost"] - r["dcost"]) <= 1e-6 * max(1.0, abs(r["pcost"]))
    # cost split adds up and the epigraph variables sit on their norms (reduction used by the product)
    assert abs(sub["J"] + sub["J_tr"] + sub["J_vc"] - sub["J_aug"]) < 1e-9
    eta = np.abs((sub["x"] - ref.xd) / scale.Sx).max(axis=1)
    assert np.abs(sub["etax"] - eta).max() < 1e-5  # epigraph variables sit on their norms up to the IPM gap


def test_reduced_structured_solver_equals_literal_program():
    """The reduction used by the HIP solver (epigraph / virtual-control variables eliminated, oracle/ipm_struct.py)
    has the same optimum as the reference's literal conic program."""
    from oracle import ipm_struct, ptr_ref
    from oracle.models import MODELS
    for model, N in (("quadrotor", 10), ("double_integrator", 12)):
        mdl = MODELS[model]()
        pars = ptr_ref.PTRParameters(N, 8, 3, 1e3, 0.1, 0, 0, 1e-3)
        scale = ptr_ref.Scaling(*mdl.bbox())
        pp = mdl.nominal_pp()
        x, u, p = mdl.guess(N, pp)
        ref = ptr_ref.discretize(mdl, pars, scale, x, u, p)
        for it in range(2):
            sub = ptr_ref.solve_subproblem(mdl, pars, scale, ref, pp)
            P = ipm_struct.build_stage_problem(mdl, pars, scale, ref, pp)
            r = ipm_struct.solve(P)
            assert r["status"] in ("OPTIMAL", "ALMOST_OPTIMAL")
            assert abs(r["pcost"] + P.cost_const - sub["J_aug"]) <= 2e-6 * max(1.0, abs(sub["J_aug"]))
            xa, ua, pa = ipm_struct.unpack(P, r["z"], r["p"])
            assert np.abs((ua - sub["u"]) / scale.Su).max() < 2e-5
            ref = ptr_ref.discretize(mdl, pars, scale, sub["x"], sub["u"], sub["p"])
