"""Pins of the exponential-cone solvers added in round 4 -- the ORACLE's (oracle/ipm.py::solve_exp, the checker of GuSTO's
`pen = :softplus`, src/solvers/gusto.jl:996-1031) and the PRODUCT's (csrc/conic_ipm.hpp, host build oracle/conic_host) -- on
mathematics that does not come from either: closed-form optima and scipy's smooth minimiser.  Cone convention (MOI's
ExponentialCone, what ECOS receives): s = h - G x = (x, y, w) with y exp(x / y) <= w, y > 0."""
import numpy as np
import pytest
import scipy.optimize as so
import scipy.sparse as sp

from oracle import conic_host, ipm


def both(c, G, h, l, q, ne, A=None, b=None, P=None):
    """the same program through the oracle's solver (exponential cones = trailing rows) and the product's host build (q = -3)"""
    o = ipm.solve(np.asarray(c, float), sp.csc_matrix(G), np.asarray(h, float), l, list(q), A=None if A is None else sp.csc_matrix(A),
                  b=None if b is None else np.asarray(b, float), P=None if P is None else sp.csc_matrix(P))
    assert o["status"] in ("OPTIMAL", "ALMOST_OPTIMAL"), o["status"]
    p = conic_host.solve(np.asarray(c, float), sp.csc_matrix(G), np.asarray(h, float), l, list(q) + [-3] * ne,
                         None if A is None else sp.csc_matrix(A), None if b is None else np.asarray(b, float),
                         P=None if P is None else sp.triu(sp.csc_matrix(P), format="csc"))
    assert p["status"] in (0, 1), p["status"]
    return o, p


def exp_rows(n, x_row, y_row, w_row):
    """three rows of [G | h] for one cone: each of x_row, y_row, w_row = (coefficients on the variables, constant), s = const + coef'v"""
    G = np.zeros((3, n)); h = np.zeros(3)
    for r, (coef, const) in enumerate((x_row, y_row, w_row)):
        G[r] = -np.asarray(coef, float); h[r] = const
    return G, h


@pytest.mark.parametrize("a", [-3.0, -0.2, 0.0])
def test_epigraph_of_exp_is_exp(a):
    """min w  s.t.  (a, 1, w) in K_exp  ->  w = exp(a); dual multiplier structure aside, the optimal value is closed form"""
    G, h = exp_rows(1, ([0.0], a), ([0.0], 1.0), ([1.0], 0.0))
    o, p = both([1.0], G, h, 0, [], 1)
    assert o["x"][0] == pytest.approx(np.exp(a), rel=1e-7) and p["x"][0] == pytest.approx(np.exp(a), rel=1e-7)


@pytest.mark.parametrize("a", [1.5, 4.0])
def test_a_cone_that_has_to_be_entered_against_its_curvature_is_not_solved_and_says_so(a):
    """A LIMIT of both solvers, pinned so that it is known: they are infeasible-start methods WITHOUT ECOS's homogeneous self-dual
    embedding.  min w s.t. (a, 1, w) in K_exp with a constant a >= 1.5 in the x row can only become feasible by driving the cone's
    third component up the exponential; the complementarity runs ahead of the residual (mu = 5e-3 with a quarter of the initial
    residual left) and every later step is cut to nothing.  What is asserted: neither solver reports success.  GuSTO's softplus
    cones (-w, 1, u), (hom f - w, 1, v) are not of this kind -- the penalty variable w sits in the x row and moves linearly (next
    tests; tests/test_template_cpu.py; the device loop in tests/test_gusto_gpu.py)."""
    G, h = exp_rows(1, ([0.0], a), ([0.0], 1.0), ([1.0], 0.0))
    o = ipm.solve(np.array([1.0]), sp.csc_matrix(G), h, 0, [])
    p = conic_host.solve(np.array([1.0]), sp.csc_matrix(G), h, 0, [-3])
    assert o["status"] == "ITERATION_LIMIT" and p["status"] == 2


@pytest.mark.parametrize("t", [-0.9, -0.5, -0.1])
def test_softplus_through_two_cones_has_the_logit_optimum(t):
    """GuSTO's softplus construction (gusto.jl:996-1031): w >= log(1 + exp(f))  <=>  exp(-w) + exp(f - w) <= 1 with
    (-w, 1, u), (f - w, 1, v) in K_exp, u + v <= 1.  min t f + w over (f, w, u, v): t + sigma(f) = 0 -> f = logit(-t),
    value t f + log(1 + exp(f)) -- the binary entropy of -t."""
    # variables v = (f, w, u, v)
    G1, h1 = exp_rows(4, ([0, -1, 0, 0], 0.0), ([0, 0, 0, 0], 1.0), ([0, 0, 1, 0], 0.0))
    G2, h2 = exp_rows(4, ([1, -1, 0, 0], 0.0), ([0, 0, 0, 0], 1.0), ([0, 0, 0, 1], 0.0))
    Gl, hl = np.array([[0.0, 0.0, 1.0, 1.0]]), np.array([1.0])                      # u + v <= 1
    G, h = np.vstack([Gl, G1, G2]), np.concatenate([hl, h1, h2])
    o, p = both([t, 1.0, 0.0, 0.0], G, h, 1, [], 2)
    f_star = np.log(-t / (1 + t))
    val = t * f_star + np.log1p(np.exp(f_star))
    for r in (o, p):
        assert r["x"][0] == pytest.approx(f_star, abs=2e-6) and float(np.dot([t, 1, 0, 0], r["x"])) == pytest.approx(val, abs=1e-7)
    assert val == pytest.approx(-(-t * np.log(-t) + (1 + t) * np.log(1 + t)), abs=1e-12)           # = H(-t), the binary entropy


def test_regularised_logistic_fit_equals_scipy():
    """min 0.5 |x|^2 + sum_i lam log(1 + exp(hom (a_i'x + b_i))) / hom + SOC-bounded x: the penalty form of a GuSTO subproblem in
    miniature (quadratic cost, softplus penalties through exponential cones, a second-order cone), against scipy's L-BFGS-B / SLSQP
    on the smooth problem."""
    rng = np.random.default_rng(7)
    n, k, lam, hom = 4, 9, 3.0, 5.0
    Am, bv = rng.standard_normal((k, n)), 0.5 * rng.standard_normal(k)
    # variables (x[n], w[k], u[k], v[k]); cost 0.5 |x|^2 + lam / hom sum w
    nv = n + 3 * k
    c = np.concatenate([np.zeros(n), lam / hom * np.ones(k), np.zeros(2 * k)])
    P = sp.diags(np.concatenate([np.ones(n), np.zeros(3 * k)]))
    rows_l, h_l = [], []
    for i in range(k):                                                                # u_i + v_i <= 1
        r = np.zeros(nv); r[n + k + i] = 1.0; r[n + 2 * k + i] = 1.0
        rows_l.append(r); h_l.append(1.0)
    # |x| <= 2 as a second-order cone (t = 2 constant)
    Gs = np.zeros((1 + n, nv)); hs = np.zeros(1 + n); hs[0] = 2.0
    for j in range(n):
        Gs[1 + j, j] = -1.0
    Ge, he = [], []
    for i in range(k):
        cw = np.zeros(nv); cw[n + i] = -1.0
        cu = np.zeros(nv); cu[n + k + i] = 1.0
        cv = np.zeros(nv); cv[n + 2 * k + i] = 1.0
        cf = np.zeros(nv); cf[:n] = hom * Am[i]; cf[n + i] = -1.0
        g1, h1 = exp_rows(nv, (cw, 0.0), (np.zeros(nv), 1.0), (cu, 0.0))
        g2, h2 = exp_rows(nv, (cf, hom * bv[i]), (np.zeros(nv), 1.0), (cv, 0.0))
        Ge += [g1, g2]; he += [h1, h2]
    G = np.vstack([np.array(rows_l), Gs] + Ge); h = np.concatenate([np.array(h_l), hs] + he)
    o, p = both(c, G, h, k, [1 + n], 2 * k, P=P)

    def smooth(x):
        z = hom * (Am @ x + bv)
        return 0.5 * x @ x + lam / hom * np.logaddexp(0.0, z).sum()

    def grad(x):
        z = hom * (Am @ x + bv)
        return x + lam * Am.T @ (1.0 / (1.0 + np.exp(-z)))
    r = so.minimize(smooth, np.zeros(n), jac=grad, method="SLSQP", constraints=[dict(type="ineq", fun=lambda x: 4.0 - x @ x, jac=lambda x: -2 * x)],
                    options=dict(ftol=1e-14, maxiter=500))
    assert r.success
    for s_ in (o, p):
        xs = s_["x"][:n]
        assert smooth(xs) == pytest.approx(r.fun, abs=1e-7) and np.abs(xs - r.x).max() < 2e-5
        assert np.linalg.norm(xs) <= 2.0 + 1e-8


def test_infeasible_exponential_program_is_reported():
    """(x, 1, w) in K_exp with w <= 0.5 and x >= 0: exp(x) <= w is impossible -- both solvers must not return a 'solution'"""
    G1, h1 = exp_rows(2, ([1, 0], 0.0), ([0, 0], 1.0), ([0, 1], 0.0))
    Gl = np.array([[0.0, 1.0], [-1.0, 0.0]]); hl = np.array([0.5, 0.0])             # w <= 0.5, -x <= 0
    G, h = np.vstack([Gl, G1]), np.concatenate([hl, h1])
    o = ipm.solve(np.array([0.0, 1.0]), sp.csc_matrix(G), h, 2, [])
    p = conic_host.solve(np.array([0.0, 1.0]), sp.csc_matrix(G), h, 2, [-3])
    assert o["status"] not in ("OPTIMAL", "ALMOST_OPTIMAL") and p["status"] not in (0, 1)
