"""Oracle-side SCvx (oracle/scvx_ref.py, literal restatement of src/solvers/scvx.jl) -- groundwork for SURVEY.md
section 8(f)1.  No golden data exists in the reference, so the loop is pinned on the algorithm's own invariants on the
reference's SCvx test configuration (test/examples/quadrotor/tests.jl:25-78)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def run():
    from oracle import scvx_ref
    pars = scvx_ref.quadrotor_test_parameters(N=20, Nsub=10, iter_max=12)
    st, hist = scvx_ref.scvx_solve("quadrotor", pars)
    return scvx_ref, pars, st, hist


def test_scvx_converges_like_the_reference_test_expects(run):
    scvx_ref, pars, st, hist = run
    assert st == "SCP_SOLVED"                       # the only assertion of the reference's own test
    last = hist[-1]
    assert last["sol"].feas                         # dynamically feasible
    assert last["sub"]["L_pen"] < 1e-6              # virtual control gone
    assert 0.0 < last["sol"].p[0] <= 2.5 + 1e-9     # time dilation within its bounds


def test_trust_region_update_rule_and_acceptance(run):
    scvx_ref, pars, st, hist = run
    eta = pars.eta_init
    ref_id = id(hist[0]["ref"])
    for h in hist:
        assert h["eta"] == eta
        if "rho" not in h:
            break
        acc, eta_next, tag = scvx_ref.update_rule(pars, h["rho"], h["eta"])
        assert (acc, eta_next, tag) == (h["accept"], h["eta_next"], h["tr_update"])
        assert pars.eta_lb <= eta_next <= pars.eta_ub
        assert h["accept"] == (h["rho"] >= pars.rho_0)
        # predicted improvement = J_ref (nonlinear) - original cost of the solution (scvx.jl:726-729, 972-973)
        assert abs(h["pre_improv"] - (h["J_ref"] - h["sub"]["L"])) < 1e-12
        assert abs(h["act_improv"] - (h["J_ref"] - h["J_sol"])) < 1e-12
        eta = eta_next
    # the trust-region bound of every solved subproblem holds: dx_lq + du_lq + dp_lq <= eta
    for h in hist:
        s = h["sub"]
        assert (s["dx_lq"] + s["du_lq"] + s["dp_lq"] <= h["eta"] * (1 + 1e-6) + 1e-7).all()


def test_scvx_and_ptr_reach_comparable_cost(run):
    scvx_ref, pars, st, hist = run
    from oracle import ptr_ref
    ppars = ptr_ref.PTRParameters(pars.N, pars.Nsub, 12, 1e3, 0.1, 0, 0, 1e-3)
    st2, h2 = ptr_ref.ptr_solve("quadrotor", ppars)
    assert st2 == "SCP_SOLVED"
    L_scvx, J_ptr = hist[-1]["sub"]["L"], h2[-1]["sub"]["J"]
    assert abs(L_scvx - J_ptr) <= 2e-2 * abs(J_ptr), (L_scvx, J_ptr)


def test_correct_convex_is_identity_on_a_feasible_guess_and_projects_an_infeasible_one():
    from oracle import scvx_ref, ptr_ref
    from oracle.models import MODELS
    mdl = MODELS["quadrotor"]()
    pars = scvx_ref.quadrotor_test_parameters(N=8, Nsub=4, iter_max=1)
    scale = ptr_ref.Scaling(*mdl.bbox())
    x, u, p = mdl.guess(8, mdl.nominal_pp())
    x2, u2, p2 = scvx_ref.correct_convex(mdl, pars, scale, x, u, p)
    assert np.abs(x2 - x).max() < 1e-6 and np.abs(u2 - u).max() < 1e-5 and np.abs(p2 - p).max() < 1e-6
    u_bad = u.copy(); u_bad[:, 3] = 30.0            # sigma above its upper bound 23.2
    x3, u3, p3 = scvx_ref.correct_convex(mdl, pars, scale, x, u_bad, p)
    # L1-closest feasible input: sigma clipped to 23.2, and a3 raised to sigma cos(60 deg) = 11.6 (pointing constraint)
    assert np.abs(u3[:, 3] - 23.2).max() < 1e-5 and np.abs(u3[:, 2] - 11.6).max() < 1e-5
    assert np.abs(u3[:, :2]).max() < 1e-5 and np.abs(x3 - x).max() < 1e-5
