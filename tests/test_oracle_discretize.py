"""Pins for the CPU oracle of `discretize!` (oracle/scp_oracle.c).

The reference has no golden vectors for this path, so the oracle
is pinned on mathematics:
  * Jacobians A,B,F of every model against central finite differences of f;
  * LTI closed forms: A_k = expm(A dt) and the FOH input integrals;
  * the reference's own *independent* FOH discretiser for the double integrator
    (test/examples/double_integrator/parameters.jl:64-78);
  * the exact identity x_{k+1} = A x_k + B- u_k + B+ u_{k+1} + F p + r + defect.
"""
import numpy as np
import pytest
from scipy.linalg import expm

MODELS = ["double_integrator", "quadrotor", "rocket_landing"]


def _rand_point(orc, model, rng):
    nx, nu, np_ = orc.MODEL_DIMS[model]
    x = rng.standard_normal(nx)
    u = rng.standard_normal(nu)
    p = 0.5 + rng.uniform(size=np_)
    return x, u, p


@pytest.mark.parametrize("model", MODELS)
def test_jacobians_match_finite_differences(orc, model):
    rng = np.random.default_rng(1)
    par = orc.default_params(model)
    for _ in range(5):
        x, u, p = _rand_point(orc, model, rng)
        f, A, B, F = orc.model_eval(model, par, 0.3, 2, x, u, p)
        h = 1e-6

        def fd(arg, i):
            d = np.zeros_like(arg)
            d[i] = h
            args = {"x": x, "u": u, "p": p}
            key = "x" if arg is x else ("u" if arg is u else "p")
            ap = dict(args); am = dict(args)
            ap[key] = arg + d; am[key] = arg - d
            return (orc.model_eval(model, par, 0.3, 2, ap["x"], ap["u"], ap["p"])[0]
                    - orc.model_eval(model, par, 0.3, 2, am["x"], am["u"], am["p"])[0]) / (2 * h)

        for i in range(x.size):
            np.testing.assert_allclose(A[:, i], fd(x, i), rtol=1e-6, atol=1e-7)
        for i in range(u.size):
            np.testing.assert_allclose(B[:, i], fd(u, i), rtol=1e-6, atol=1e-7)
        for i in range(p.size):
            np.testing.assert_allclose(F[:, i], fd(p, i), rtol=1e-6, atol=1e-7)


def _foh_closed_form(Ac, Bc, dt):
    """Exact FOH discretisation of xdot = Ac x + Bc u(t), u piecewise affine."""
    from scipy.integrate import quad_vec
    Ad = expm(Ac * dt)
    Bm = quad_vec(lambda s: expm(Ac * (dt - s)) @ Bc * (dt - s) / dt, 0, dt, epsabs=1e-14, epsrel=1e-14)[0]
    Bp = quad_vec(lambda s: expm(Ac * (dt - s)) @ Bc * s / dt, 0, dt, epsabs=1e-14, epsrel=1e-14)[0]
    return Ad, Bm, Bp


@pytest.mark.parametrize("model", ["quadrotor", "rocket_landing"])
def test_lti_closed_form(orc, model):
    """With p fixed both models are LTI in (x,u): A_k, B-_k, B+_k must equal the
    matrix-exponential closed form to RK4 truncation accuracy."""
    rng = np.random.default_rng(2)
    nx, nu, np_ = orc.MODEL_DIMS[model]
    par = orc.default_params(model)
    N, Nsub = 6, 15
    xd = rng.standard_normal((1, N, nx))
    ud = rng.standard_normal((1, N, nu))
    p = np.array([[1.7]])
    out = orc.discretize(model, par, N, Nsub, xd, ud, p, np.ones(nx), 1e-3)
    _, Ac, Bc, _ = orc.model_eval(model, par, 0.0, 1, xd[0, 0], ud[0, 0], p[0])
    Ad, Bm, Bp = _foh_closed_form(Ac, Bc, 1.0 / (N - 1))
    for k in range(N - 1):
        np.testing.assert_allclose(out["A"][0, k].T, Ad, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(out["Bm"][0, k].T, Bm, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(out["Bp"][0, k].T, Bp, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(out["E"][0, k].T @ np.eye(nx),
                                   _foh_closed_form(Ac, np.eye(nx), 1.0 / (N - 1))[1]
                                   + _foh_closed_form(Ac, np.eye(nx), 1.0 / (N - 1))[2], rtol=1e-9, atol=1e-11)


def _rk4(f, x0, grid):
    """classic RK4 as in src/utils/helper.jl:411-424."""
    x = x0.copy()
    for a, b in zip(grid[:-1], grid[1:]):
        h = b - a
        k1 = f(a, x); k2 = f(a + h / 2, x + h / 2 * k1); k3 = f(a + h / 2, x + h / 2 * k2); k4 = f(a + h, x + h * k3)
        x = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    return x


def test_double_integrator_against_reference_independent_discretiser(orc):
    """test/examples/double_integrator/parameters.jl:64-78 computes A, Bm, Bp, w of
    the same FOH discretisation by a different route (RK4 of exp(A(dt-t))B(dt-t)/dt
    on a 1000-point grid).  The oracle's discretize! must reproduce it."""
    T, N, g = 10.0, 50, 0.1
    A = np.array([[0.0, 1.0], [0.0, 0.0]]); Bv = np.array([0.0, 1.0])
    dt = T / (N - 1)
    grid = np.linspace(0, dt, 1000)
    Bm = _rk4(lambda t, x: expm(A * (dt - t)) @ Bv * (dt - t) / dt, np.zeros(2), grid)
    Bp = _rk4(lambda t, x: expm(A * (dt - t)) @ Bv * t / dt, np.zeros(2), grid)
    w = _rk4(lambda t, x: expm(A * (dt - t)) @ np.array([0.0, -g]), np.zeros(2), grid)
    Ad = expm(A * dt)
    rng = np.random.default_rng(3)
    xd = rng.standard_normal((1, N, 2)); ud = rng.standard_normal((1, N, 1))
    out = orc.discretize("double_integrator", np.array([g, T]), N, 10, xd, ud, np.zeros((1, 0)), np.ones(2), 1e-3)
    for k in (0, 7, N - 2):
        np.testing.assert_allclose(out["A"][0, k].T, Ad, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(out["Bm"][0, k][0], Bm, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(out["Bp"][0, k][0], Bp, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(out["r"][0, k], w, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("model", MODELS)
def test_linearisation_identity_and_defect(orc, model):
    """x_{k+1} - defect_k == A x_k + B- u_k + B+ u_{k+1} + F p + r exactly at the
    reference point (the linearisation is exact at the point it is taken about),
    and feas == all(||iSx*defect||_inf <= feas_tol)  (discretization.jl:205-210)."""
    rng = np.random.default_rng(4)
    nx, nu, np_ = orc.MODEL_DIMS[model]
    par = orc.default_params(model)
    N, Nsub, B = 7, 9, 3
    xd = rng.standard_normal((B, N, nx)); ud = rng.standard_normal((B, N, nu))
    p = 0.5 + rng.uniform(size=(B, np_))
    iSx = 1.0 / (1.0 + rng.uniform(size=nx))
    out = orc.discretize(model, par, N, Nsub, xd, ud, p, iSx, 0.5)
    for b in range(B):
        ok = True
        for k in range(N - 1):
            lin = (out["A"][b, k].T @ xd[b, k] + out["Bm"][b, k].T @ ud[b, k] + out["Bp"][b, k].T @ ud[b, k + 1]
                   + out["F"][b, k].T @ p[b] + out["r"][b, k])
            np.testing.assert_allclose(xd[b, k + 1] - out["defect"][b, k], lin, rtol=1e-9, atol=1e-9)
            ok &= np.max(np.abs(iSx * out["defect"][b, k])) <= 0.5
        assert bool(out["feas"][b]) == ok


def test_propagate_closed_form_double_integrator(orc):
    """`propagate` pin: for the double integrator with node-wise linear input the state is a cubic in time, which
    RK4 integrates exactly on any grid -> compare with the analytic solution (and with the sample times)."""
    import numpy as np
    N, res = 6, 41
    par = orc.default_params("double_integrator")   # [g, T]
    g, T = par
    rng = np.random.default_rng(5)
    ud = rng.uniform(-2, 2, size=(N, 1))
    xd = np.zeros((N, 2)); xd[0] = [0.3, -0.2]
    tc, xc = orc.propagate("double_integrator", par, N, xd, ud, np.zeros(0), res=res)
    assert abs(tc[0]) == 0 and abs(tc[-1] - 1) < 1e-15 and xc.shape == (res, 2)
    # analytic: v' = T (u(t) - g), r' = T v with u piecewise linear on LinRange(0,1,N)
    grid = np.linspace(0, 1, N)
    def exact(t):
        v, r, t0 = xd[0, 1], xd[0, 0], 0.0
        for k in range(N - 1):
            t1 = min(t, grid[k + 1])
            if t1 <= t0:
                break
            h = t1 - t0
            du = (ud[k + 1, 0] - ud[k, 0]) / (grid[k + 1] - grid[k])
            u0 = ud[k, 0] + du * (t0 - grid[k])
            # integrate over [t0, t1]
            r = r + T * (v * h + T * ((u0 - g) * h ** 2 / 2 + du * h ** 3 / 6))
            v = v + T * ((u0 - g) * h + du * h ** 2 / 2)
            t0 = t1
        return np.array([r, v])
    ref = np.stack([exact(t) for t in tc])
    assert np.abs(xc - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
