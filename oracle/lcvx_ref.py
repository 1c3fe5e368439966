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


# ------------------------------------------------------------------------------------------------------------------------------
# Lossless-convexification 3-DoF rocket landing (test/examples/rocket_landing: parameters.jl:75-150, definition.jl:33-150, tests.jl:23-35):
# the minimum-fuel powered-descent program for a fixed time of flight and the golden-section search over the time of flight.
# ------------------------------------------------------------------------------------------------------------------------------
def skew(v):
    """helper.jl:65-70"""
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


class Rocket:
    """parameters.jl:75-150"""

    def __init__(self):
        ex, ey, ez = np.eye(3)
        self.g = -3.7114 * ez
        th = 30 * np.pi / 180
        T_sid = 24.6229 * 3600
        self.om = (2 * np.pi / T_sid) * (ex * np.cos(th) + ey * 0 + ez * np.sin(th))
        self.m_dry, self.m_wet, self.Isp = 1505.0, 1905.0, 225.0
        n_eng, self.phi = 6, 27 * np.pi / 180
        T_max = 3.1e3
        T_1, T_2 = 0.3 * T_max, 0.8 * T_max
        self.rho_min = n_eng * T_1 * np.cos(self.phi)
        self.rho_max = n_eng * T_2 * np.cos(self.phi)
        self.gam_gs, self.gam_p = 86 * np.pi / 180, 40 * np.pi / 180
        self.v_max = 500 * 1e3 / 3600
        self.r0 = (2 * ex + 0 * ey + 1.5 * ez) * 1e3
        self.v0 = 80 * ex + 30 * ey - 75 * ez
        self.dt = 1.0
        ge = 9.807
        self.alpha = 1 / (self.Isp * ge * np.cos(self.phi))
        W = skew(self.om)
        self.A_c = np.block([[np.zeros((3, 3)), np.eye(3), np.zeros((3, 1))], [-(W @ W), -2 * W, np.zeros((3, 1))], [np.zeros((1, 7))]])
        self.B_c = np.block([[np.zeros((3, 4))], [np.eye(3), np.zeros((3, 1))], [np.zeros((1, 3)), -self.alpha * np.ones((1, 1))]])
        self.p_c = np.concatenate([np.zeros(3), self.g, [0.0]])


def c2d(A, B, p, dt):
    """helper.jl:248-265: zero-order-hold discretisation through one matrix exponential"""
    n, m = A.shape[0], B.shape[1]
    M = np.zeros((n + m + 1, n + m + 1))
    M[:n, :n] = A; M[:n, n:n + m] = B; M[:n, n + m] = p
    E = sla.expm(M * dt)
    return E[:n, :n], E[:n, n:n + m], E[:n, n + m]


def golden(f, a, b, tol=1e-3):
    """helper.jl:291-331"""
    phi = (1 + np.sqrt(5)) / 2
    n = int(np.ceil(np.log((b - a) / tol) / np.log(phi) + 1))
    rho = phi - 1
    d = rho * b + (1 - rho) * a
    yd = f(d)
    for _ in range(n - 1):
        c = rho * a + (1 - rho) * b
        yc = f(c)
        if yc < yd:
            b, d, yd = d, c, yc
        else:
            a, b = b, c
    return b, f(b)


def pdg_program(rocket, tf):
    """`solve_pdg_fft` (definition.jl:33-150) in standard form, in the reference's SCALED variables (definition.jl:56-86).  Variable order:
    r_s[3, N], v_s[3, N], z_s[N], u_s[3, N-1], xi_s[N-1] (column-major blocks).  The quadratic thrust lower bound
    xi >= mu_min (1 - dz + dz^2 / 2) enters as the rotated second-order cone JuMP's quadratic bridge produces:
    dz^2 <= 2 tau, tau = xi / mu_min - 1 + dz  <=>  (tau + 1, tau - 1, sqrt(2) dz) in Q^3."""
    R = rocket
    N = int(np.floor(tf / R.dt)) + 1 + int(tf % R.dt != 0)
    dt = tf / (N - 1)
    t = np.arange(N) * dt
    A, B, p = c2d(R.A_c, R.B_c, R.p_c, dt)
    ir = lambda i, k: 3 * k + i
    iv = lambda i, k: 3 * N + 3 * k + i
    iz = lambda k: 6 * N + k
    iu = lambda i, k: 7 * N + 3 * k + i
    ixi = lambda k: 7 * N + 3 * (N - 1) + k
    n = 7 * N + 4 * (N - 1)
    S_r = np.maximum(1.0, np.abs(R.r0)); S_v = np.maximum(1.0, np.abs(R.v0))
    s_z = (np.log(R.m_dry) + np.log(R.m_wet)) / 2; S_z = np.log(R.m_wet) - s_z
    s_u = np.array([0.0, 0.0, 0.5 * (R.rho_min / R.m_wet * np.cos(R.gam_p) + R.rho_max / R.m_dry)])
    S_u = np.array([R.rho_max / R.m_dry * np.sin(R.gam_p), R.rho_max / R.m_dry * np.sin(R.gam_p), R.rho_max / R.m_dry - s_u[2]])
    s_xi, S_xi = s_u[2], S_u[2]
    # physical quantity = list of (index, coefficient) + constant
    r = lambda i, k: ([(ir(i, k), S_r[i])], 0.0)
    v = lambda i, k: ([(iv(i, k), S_v[i])], 0.0)
    z = lambda k: ([(iz(k), S_z)], s_z)
    u = lambda i, k: ([(iu(i, k), S_u[i])], s_u[i])
    xi = lambda k: ([(ixi(k), S_xi)], s_xi)
    X = lambda k: [r(0, k), r(1, k), r(2, k), v(0, k), v(1, k), v(2, k), z(k)]
    U = lambda k: [u(0, k), u(1, k), u(2, k), xi(k)]

    def lin(terms):            # sum of coef * (physical expr) -> (entries, const)
        ent, c0 = {}, 0.0
        for coef, (e, c) in terms:
            c0 += coef * c
            for j, a in (e.items() if isinstance(e, dict) else e):
                ent[j] = ent.get(j, 0.0) + coef * a
        return ent, c0
    c = np.zeros(n)
    for k in range(N - 1):
        c[ixi(k)] = dt * S_xi
    cost_const = dt * (N - 1) * s_xi
    Ar, Ac, Av, b = [], [], [], []

    def eq(expr):              # expr == 0
        ent, c0 = expr
        ent = dict(ent) if not isinstance(ent, dict) else ent
        rr = len(b)
        for j, a in ent.items():
            if a != 0.0:
                Ar.append(rr); Ac.append(j); Av.append(a)
        b.append(-c0)
    Gr, Gc, Gv, h = [], [], [], []

    def cone_row(expr):        # appends one row of s = h - G x with s = expr
        ent, c0 = expr
        ent = dict(ent) if not isinstance(ent, dict) else ent
        rr = len(h)
        for j, a in ent.items():
            if a != 0.0:
                Gr.append(rr); Gc.append(j); Gv.append(-a)
        h.append(c0)
    for k in range(N - 1):     # dynamics
        Xk, Xn, Uk = X(k), X(k + 1), U(k)
        for i in range(7):
            eq(lin([(1.0, Xn[i])] + [(-A[i, j], Xk[j]) for j in range(7)] + [(-B[i, j], Uk[j]) for j in range(4)] + [(-p[i], ([], 1.0))]))
    for i in range(3):         # boundary conditions
        eq(lin([(1.0, r(i, 0)), (-R.r0[i], ([], 1.0))])); eq(lin([(1.0, v(i, 0)), (-R.v0[i], ([], 1.0))]))
    eq(lin([(1.0, z(0)), (-np.log(R.m_wet), ([], 1.0))]))
    for i in range(3):
        eq(lin([(1.0, r(i, N - 1))])); eq(lin([(1.0, v(i, N - 1))]))
    z0 = lambda k: np.log(R.m_wet - R.alpha * R.rho_max * t[k])
    mu_min = lambda k: R.rho_min * np.exp(-z0(k))
    mu_max = lambda k: R.rho_max * np.exp(-z0(k))
    one = ([], 1.0)
    # ---- R+ rows: expr >= 0 ----
    for k in range(N - 1):
        cone_row(lin([(mu_max(k), one), (-mu_max(k), z(k)), (mu_max(k) * z0(k), one), (-1.0, xi(k))]))       # xi <= mu_max (1 - dz)
        cone_row(lin([(1.0, u(2, k)), (-np.cos(R.gam_p), xi(k))]))                                            # pointing
    for k in range(N):
        cone_row(lin([(1.0, z(k)), (-z0(k), one)]))
        cone_row(lin([(np.log(R.m_wet - R.alpha * R.rho_min * t[k]), one), (-1.0, z(k))]))
        cg, sg = np.cos(R.gam_gs), np.sin(R.gam_gs)
        for Hrow in ([cg, 0, -sg], [-cg, 0, -sg], [0, cg, -sg], [0, -cg, -sg]):                               # glide slope: H r <= 0
            cone_row(lin([(-Hrow[i], r(i, k)) for i in range(3)]))
    cone_row(lin([(1.0, z(N - 1)), (-np.log(R.m_dry), one)]))
    l = len(h)
    q = []
    for k in range(N - 1):     # thrust lower bound (rotated cone), thrust LCvx cone
        tau = lin([(1.0 / mu_min(k), xi(k)), (-1.0, one), (1.0, z(k)), (-z0(k), one)])
        cone_row(lin([(1.0, tau), (1.0, one)])); cone_row(lin([(1.0, tau), (-1.0, one)])); cone_row(lin([(np.sqrt(2.0), z(k)), (-np.sqrt(2.0) * z0(k), one)]))
        q.append(3)
        cone_row(xi(k)); [cone_row(u(i, k)) for i in range(3)]
        q.append(4)
    for k in range(N):         # velocity bound
        cone_row(([], R.v_max)); [cone_row(v(i, k)) for i in range(3)]
        q.append(4)
    Amat = sp.csc_matrix((Av, (Ar, Ac)), shape=(len(b), n))
    Gmat = sp.csc_matrix((Gv, (Gr, Gc)), shape=(len(h), n))
    return dict(c=c, cost_const=cost_const, A=Amat, b=np.array(b), G=Gmat, h=np.array(h), l=l, q=q, n=n, N=N, dt=dt, t=t,
                unscale=dict(S_r=S_r, S_v=S_v, S_z=S_z, s_z=s_z, S_u=S_u, s_u=s_u, S_xi=S_xi, s_xi=s_xi))


def pdg_cost(rocket, tf, solve):
    """cost of `solve_pdg_fft(rocket, tf)` (Inf when the solver does not return OPTIMAL: definition.jl:126-128 -> FailedSolution).
    `solve(P)` -> (status_is_optimal, x)."""
    P = pdg_program(rocket, tf)
    ok, x = solve(P)
    return (float(P["c"] @ x + P["cost_const"]) if ok else np.inf), P, x
