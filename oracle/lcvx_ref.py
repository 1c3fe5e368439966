"""CPU ORACLE (test infrastructure, NOT product code): the reference's lossless-convexification examples, restated.

These are the only KNOWN ANSWERS the reference's own tests hold for anything on the conic seam (no golden vectors
anywhere else): `test/examples/double_integrator/tests.jl:25-45` solves the double integrator with friction twice -- analytically by
Pontryagin's maximum principle with a shooting search (`solve_mp`, definition.jl:137-217, `mp_input` :219-248, `mp_sim` :264-294) and
numerically as ONE conic program through `ConicProgram` -> ECOS (`solve_lcvx`, definition.jl:38-118) -- and plots one over the other.
Restated here:

  * `DoubleIntegratorParameters(choice)`  -- parameters.jl:50-87 (FOH discretisation by rk4 on a 1000-point grid, helper.jl:411-501);
  * `solve_mp(mdl)`                        -- the shooting solution (the known answer);
  * `lcvx_program(mdl)`                    -- the conic program of `solve_lcvx` in the solver's standard form
        min c'x  s.t.  A x = b,  G x + s = h,  s in R+^l x Q^{q...}
    with the reference's cones lowered the way JuMP's bridges hand them to a second-order-cone solver (src/parser/cone.jl:36-47,
    :150-166): ZERO -> equality rows; NONPOS -> R+ rows; L1 (t, x) -> |x_i| <= y_i, sum y <= t (scalar x: two R+ rows);
    GEOM (t, x1, x2 = 1) -> (x1 + 1, 2 t, x1 - 1) in Q^3 (geometric-mean bridge, t^2 <= x1 on t >= 0 ... and any t <= 0).

Parity status: pinned to the reference's OWN known answer -- the maximum-principle solution, which needs no solver.  The conic
solve (oracle/ipm.py here, `socp_solve_batch` on the device) is compared with it in tests/test_lcvx_cpu.py / test_lcvx_gpu.py.
"""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp


def rk4(f, x0, tspan, full=False):
    """helper.jl:350-359,451-501 + rk4_core_step :411-424 (no integration actions)"""
    x = np.array(x0, float)
    X = [x.copy()]
    for t, tp in zip(tspan[:-1], tspan[1:]):
        h = tp - t
        k1 = f(t, x)
        k2 = f(t + h / 2, x + h / 2 * k1)
        k3 = f(t + h / 2, x + h / 2 * k2)
        k4 = f(t + h, x + h * k3)
        x = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        if full:
            X.append(x.copy())
    return np.array(X).T if full else x


class DoubleIntegratorParameters:
    """parameters.jl:50-87"""

    def __init__(self, choice, T=10.0):
        assert choice in (1, 2)
        self.choice = choice
        self.T = float(T)
        self.N = 50
        A = np.array([[0.0, 1.0], [0.0, 0.0]])
        B = np.array([0.0, 1.0])
        self.g = 0.1 if choice == 1 else 0.6
        self.s = 47.0 if choice == 1 else 30.0
        g = self.g
        self.f = lambda t, x, u: np.array([x[1], u - g])
        self.n, self.m = 2, 1
        dt = self.T / (self.N - 1)
        tg = np.linspace(0.0, dt, 1000)
        self.Bm = rk4(lambda t, x: sla.expm(A * (dt - t)) @ B * (dt - t) / dt, np.zeros(2), tg)
        self.Bp = rk4(lambda t, x: sla.expm(A * (dt - t)) @ B * t / dt, np.zeros(2), tg)
        self.w = rk4(lambda t, x: sla.expm(A * (dt - t)) @ np.array([0.0, -g]), np.zeros(2), tg)
        self.A = sla.expm(A * dt)
        self.dt = dt


def mp_input(p):
    """definition.jl:219-248"""
    if p > 4:
        return 2.0
    if 2 <= p <= 4:
        return p / 2
    if 0 <= p < 2:
        return 1.0
    if -2 <= p < 0:
        return -1.0
    if -4 <= p < -2:
        return p / 2
    return -2.0


def mp_sim(f, T, s, c, ts):
    """definition.jl:264-294"""
    p = lambda t: c * (t - ts)
    t_crit = [ts + a / c for a in (4, 2, 0, -2, -4)]
    t_crit = [tau for tau in t_crit if 0 <= tau <= T]
    t_crit = [0.0] + t_crit + [T]
    grids = [np.linspace(t_crit[i], t_crit[i + 1], 100) for i in range(len(t_crit) - 1)]
    xs = []
    for i, tg in enumerate(grids):
        x0 = np.zeros(2) if i == 0 else xs[i - 1][:, -1]
        xs.append(rk4(lambda t, x: f(t, x, mp_input(p(t))), x0, tg, full=True))
    t = np.concatenate(grids)
    x = np.concatenate(xs, axis=1)
    err = float(np.linalg.norm(x[:, -1] - np.array([s, 0.0])))
    return dict(c=c, ts=ts, err=err, t=t, x=x)


def solve_mp(mdl, N_grid=25, tol_err=1e-2, max_iter=10):
    """definition.jl:137-217: shooting on (c, ts) by an iterated grid search; returns t, x, u and the searched (c, ts, err)"""
    run = lambda c, ts: mp_sim(mdl.f, mdl.T, mdl.s, c, ts)
    c_range, ts_range = ((-3.0, -1.0), (4.5, 5.5)) if mdl.choice == 1 else ((-1.5, -0.5), (6.5, 7.5))
    cg = np.linspace(c_range[0], c_range[1], N_grid)
    tg = np.linspace(ts_range[0], ts_range[1], N_grid)
    c_grid = np.ones((N_grid, 1)) * cg[None, :]          # c varies along the columns
    ts_grid = tg[:, None] * np.ones((1, N_grid))         # ts along the rows
    it = 1
    while True:
        err = np.array([[run(c_grid[i, j], ts_grid[i, j])["err"] for j in range(N_grid)] for i in range(N_grid)])
        pad = err[1:-1, 1:-1]
        k = int(np.argmin(pad.T.reshape(-1)))             # Julia's column-major argmin of pad[:]
        i, j = k % (N_grid - 2), k // (N_grid - 2)
        if pad[i, j] <= tol_err:
            c, ts = c_grid[1:-1, 1:-1][i, j], ts_grid[1:-1, 1:-1][i, j]
            break
        i += 1; j += 1
        cg = np.linspace(c_grid[i, j - 1], c_grid[i, j + 1], N_grid)
        tg = np.linspace(ts_grid[i - 1, j], ts_grid[i + 1, j], N_grid)
        c_grid = np.ones((N_grid, 1)) * cg[None, :]
        ts_grid = tg[:, None] * np.ones((1, N_grid))
        it += 1
        if it > max_iter:
            raise RuntimeError("failed to find a solution")
    out = run(c, ts)
    u = np.array([mp_input(c * (t - ts)) for t in out["t"]])
    return dict(t=out["t"], x=out["x"], u=u, c=float(c), ts=float(ts), err=out["err"], iterations=it)


def lcvx_program(mdl):
    """definition.jl:38-118 in standard form.  Variable order: x[2, N] (column-major: x_1k, x_2k per node), u[N], sigma[N], sigma2[N].
    Returns dict(c, A, b, G, h, l, q, n, idx) with idx = slices of the four blocks; cost = sum(sigma2) * dt."""
    N, dt = mdl.N, mdl.dt
    ix = lambda k: 2 * k
    iu = lambda k: 2 * N + k
    isg = lambda k: 3 * N + k
    is2 = lambda k: 4 * N + k
    n = 5 * N
    c = np.zeros(n)
    c[4 * N:] = dt
    # ZERO cones: initial condition, final condition, dynamics
    rows, cols, vals, b = [], [], [], []
    def eq(entries, rhs):
        r = len(b)
        for j, v in entries:
            rows.append(r); cols.append(j); vals.append(v)
        b.append(rhs)
    eq([(ix(0), 1.0)], 0.0); eq([(ix(0) + 1, 1.0)], 0.0)
    eq([(ix(N - 1), 1.0)], mdl.s); eq([(ix(N - 1) + 1, 1.0)], 0.0)
    for k in range(N - 1):
        for i in range(2):      # x_{k+1} - (A x_k + Bm u_k + Bp u_{k+1} + w) = 0
            eq([(ix(k + 1) + i, 1.0), (ix(k), -mdl.A[i, 0]), (ix(k) + 1, -mdl.A[i, 1]), (iu(k), -mdl.Bm[i]), (iu(k + 1), -mdl.Bp[i])], mdl.w[i])
    A = sp.csc_matrix((vals, (rows, cols)), shape=(len(b), n))
    # R+ rows (G x + s = h, s >= 0  <=>  G x <= h): sigma <= 2, 1 <= sigma, +-u <= sigma
    grow, gcol, gval, h = [], [], [], []
    def le(entries, rhs):
        r = len(h)
        for j, v in entries:
            grow.append(r); gcol.append(j); gval.append(v)
        h.append(rhs)
    for k in range(N):
        le([(isg(k), 1.0)], 2.0)
        le([(isg(k), -1.0)], -1.0)
        le([(iu(k), 1.0), (isg(k), -1.0)], 0.0)
        le([(iu(k), -1.0), (isg(k), -1.0)], 0.0)
    l = len(h)
    # GEOM (sigma, sigma2, 1): s = (sigma2 + 1, 2 sigma, sigma2 - 1) in Q^3  ->  rows of -G x + h
    q = []
    for k in range(N):
        le([(is2(k), -1.0)], 1.0)
        le([(isg(k), -2.0)], 0.0)
        le([(is2(k), -1.0)], -1.0)
        q.append(3)
    G = sp.csc_matrix((gval, (grow, gcol)), shape=(len(h), n))
    return dict(c=c, A=A, b=np.array(b), G=G, h=np.array(h), l=l, q=q, n=n,
                idx=dict(x=slice(0, 2 * N), u=slice(2 * N, 3 * N), sigma=slice(3 * N, 4 * N), sigma2=slice(4 * N, 5 * N)))


def compare_with_mp(mdl, xsol, mp):
    """LCvx solution (standard-form vector) against the maximum-principle trajectory: states and input at the LCvx grid nodes (the MP
    trajectory interpolated linearly on its fine grid), and the two costs int u^2 dt."""
    N = mdl.N
    t = np.linspace(0.0, mdl.T, N)
    x = xsol[: 2 * N].reshape(N, 2).T
    u = xsol[2 * N: 3 * N]
    s2 = xsol[4 * N: 5 * N]
    # the MP grid repeats the switch times: make it strictly increasing for the interpolation
    tm, keep = np.unique(mp["t"], return_index=True)
    xm = np.stack([np.interp(t, tm, mp["x"][i, keep]) for i in range(2)])
    um = np.interp(t, tm, mp["u"][keep])
    cost_lcvx = float(np.sum(s2) * mdl.dt)
    trapz = getattr(np, "trapezoid", None) or np.trapz
    cost_mp = float(trapz(mp["u"] ** 2, mp["t"]))
    return dict(pos_err_max=float(np.abs(x[0] - xm[0]).max()), vel_err_max=float(np.abs(x[1] - xm[1]).max()),
                u_err_max=float(np.abs(u - um).max()), u_err_rms=float(np.sqrt(np.mean((u - um) ** 2))),
                cost_lcvx=cost_lcvx, cost_mp=cost_mp, cost_rel_diff=abs(cost_lcvx - cost_mp) / cost_mp)
