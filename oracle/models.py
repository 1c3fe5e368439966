"""CPU ORACLE (test infrastructure; parity status "parity unpinned", see oracle/__init__.py): numpy restatement of the example problem
definitions -- everything a `TrajectoryProblem` carries besides the dynamics
(which live in oracle/scp_oracle.c): non-convex constraints s/C/D/G, convex
sets X/U as cone rows, boundary conditions, cost, initial guess, scaling boxes.

Conventions: math layout (matrices are [rows, cols]); a cone constraint is a
tuple (kind, Mx_or_Mu, Mp, m0) meaning  z = M*x(or u) + Mp*p + m0  with
  NONPOS: z <= 0        SOC: z[0] >= ||z[1:]||_2
(src/parser/cone.jl:36-47).  `pp` is the per-problem data vector (Monte-Carlo
initial/terminal conditions).
"""
import numpy as _np

from .oracle import default_params  # noqa: F401  (re-export)


def linrange(a, b, n):
    j = _np.arange(n, dtype=_np.float64)
    t = j / (n - 1)
    return (1.0 - t) * a + t * b


def straightline_interpolate(v0, vf, N):
    """src/utils/helper.jl:203-219 -> [N, nv]."""
    t = linrange(0.0, 1.0, N)
    c = (1.0 - t) / 1.0
    return c[:, None] * _np.asarray(v0, float)[None, :] + (1.0 - c)[:, None] * _np.asarray(vf, float)[None, :]


class Quadrotor:
    """test/examples/quadrotor/{parameters,definition}.jl (SCvx/PTR form of the
    running cost and of s, definition.jl:110-123, 255-290)."""
    name = "quadrotor"
    nx, nu, np, ns, nic, ntc = 6, 4, 1, 2, 6, 6
    g = 9.81
    u_min, u_max, tilt_max = 0.6, 23.2, 60 * _np.pi / 180
    tf_min, tf_max, gamma = 0.0, 2.5, 0.0
    obs_H = [_np.diag([2.0, 2.0, 0.0]), _np.diag([1.5, 1.5, 0.0])]   # parameters.jl:114-118
    obs_c = [_np.array([1.0, 2.0, 0.0]), _np.array([2.0, 5.0, 0.0])]

    def par(self):
        """parameter blob of the compiled model (csrc/models/quadrotor.hpp; the C oracle reads its leading entry g)"""
        obs = [v for H, c in zip(self.obs_H, self.obs_c) for v in list(_np.diag(H)) + list(c)]
        return _np.array([self.g, self.u_min, self.u_max, self.tilt_max, self.tf_min, self.tf_max, self.gamma] + obs, dtype=float)

    def nominal_pp(self):
        return _np.array([0, 0, 0, 0, 0, 0, 2.5, 6.0, 0, 0, 0, 0], dtype=float)  # [r0 v0 rf vf]

    def bbox(self):
        lat = self.u_max * _np.sin(self.tilt_max)
        ub = _np.array([[-lat, lat], [-lat, lat], [self.u_min * _np.cos(self.tilt_max), self.u_max],
                       [self.u_min, self.u_max]])
        return _np.tile([[0.0, 1.0]], (6, 1)), ub, _np.array([[self.tf_min, self.tf_max]])

    def guess(self, N, pp):  # definition.jl:60-90
        x = straightline_interpolate(pp[0:6], pp[6:12], N)
        hover = _np.array([0, 0, self.g, self.g])
        return x, straightline_interpolate(hover, hover, N), _np.array([0.5 * (self.tf_min + self.tf_max)])

    # -- cost: phi = gamma*(tdil/tdil_max)^2, Gamma = (1-gamma)*(sigma/|g|)^2 (definition.jl:92-138)
    def cost_terms(self):
        Qu = _np.zeros(self.nu); Qu[3] = (1 - self.gamma) / self.g ** 2
        return dict(Qu=Qu, lu=_np.zeros(self.nu), lx=_np.zeros(self.nx), tx=_np.zeros(self.nx),
                    tp=_np.zeros(self.np), Qp=_np.array([self.gamma / self.tf_max ** 2]))

    def X(self, t, k):
        return []

    def U(self, t, k):  # definition.jl:188-253
        nu = self.nu
        rows = []
        e = _np.zeros((1, nu)); e[0, 3] = -1
        rows.append(("NONPOS", e, _np.zeros((1, 1)), _np.array([self.u_min])))
        e = _np.zeros((1, nu)); e[0, 3] = 1
        rows.append(("NONPOS", e, _np.zeros((1, 1)), _np.array([-self.u_max])))
        M = _np.zeros((4, nu)); M[0, 3] = 1; M[1, 0] = 1; M[2, 1] = 1; M[3, 2] = 1
        rows.append(("SOC", M, _np.zeros((4, 1)), _np.zeros(4)))
        e = _np.zeros((1, nu)); e[0, 3] = _np.cos(self.tilt_max); e[0, 2] = -1
        rows.append(("NONPOS", e, _np.zeros((1, 1)), _np.zeros(1)))
        rows.append(("NONPOS", _np.zeros((1, nu)), _np.ones((1, 1)), _np.array([-self.tf_max])))
        rows.append(("NONPOS", _np.zeros((1, nu)), -_np.ones((1, 1)), _np.array([self.tf_min])))
        return rows

    def s(self, t, k, x, u, p):  # definition.jl:256-268, ellipsoid.jl:99-118
        return _np.array([1 - _np.linalg.norm(H @ (x[0:3] - c)) for H, c in zip(self.obs_H, self.obs_c)])

    def C(self, t, k, x, u, p):
        C = _np.zeros((self.ns, self.nx))
        for i, (H, c) in enumerate(zip(self.obs_H, self.obs_c)):
            r = x[0:3]
            C[i, 0:3] = -(H.T @ H) @ (r - c) / _np.linalg.norm(H @ (r - c))
        return C

    def D(self, t, k, x, u, p):
        return _np.zeros((self.ns, self.nu))

    def G(self, t, k, x, u, p):
        return _np.zeros((self.ns, self.np))

    # -- boundary conditions (definition.jl:294-351)
    def gic(self, x, p, pp):
        return x - pp[0:6]

    def H0(self, x, p, pp):
        return _np.eye(6)

    def K0(self, x, p, pp):
        return _np.zeros((6, 1))

    def gtc(self, x, p, pp):
        return x - pp[6:12]

    def Hf(self, x, p, pp):
        return _np.eye(6)

    def Kf(self, x, p, pp):
        return _np.zeros((6, 1))


class RocketLanding:
    """Builder-defined free-final-time landing problem over
    test/examples/rocket_landing/parameters.jl:77-146, definition.jl:84-130."""
    name = "rocket_landing"
    nx, nu, np, ns, nic, ntc = 7, 4, 1, 2, 7, 6
    m_dry, m_wet = 1505.0, 1905.0
    tf_min, tf_max = 40.0, 120.0
    gamma_gs, gamma_p = 86 * _np.pi / 180, 40 * _np.pi / 180
    v_max = 500 * 1e3 / 3600
    cost_weight = 1.0

    def par(self):
        """parameter blob of the compiled model (csrc/models/rocket_landing.hpp; the C oracle reads the leading [g, omega, alpha])"""
        rmin, rmax = self.thrust_limits()
        return _np.concatenate([default_params("rocket_landing"), [self.m_dry, self.m_wet, rmin, rmax, self.gamma_gs, self.gamma_p,
                                                                   self.v_max, self.tf_min, self.tf_max, self.cost_weight]])

    def thrust_limits(self):
        n_eng, phi, T_max = 6, 27 * _np.pi / 180, 3.1e3
        return n_eng * 0.3 * T_max * _np.cos(phi), n_eng * 0.8 * T_max * _np.cos(phi)

    def nominal_pp(self):
        return _np.array([2000.0, 0.0, 1500.0, 80.0, 30.0, -75.0])

    def bbox(self):
        _, rho_max = self.thrust_limits()
        a_max = rho_max / self.m_dry
        xb = _np.array([[-2500.0, 2500.0], [-2500.0, 2500.0], [0.0, 2500.0], [-self.v_max, self.v_max],
                       [-self.v_max, self.v_max], [-self.v_max, self.v_max],
                       [_np.log(self.m_dry), _np.log(self.m_wet)]])
        ub = _np.array([[-a_max, a_max], [-a_max, a_max], [0.0, a_max], [0.0, a_max]])
        return xb, ub, _np.array([[self.tf_min, self.tf_max]])

    def guess(self, N, pp):
        x0 = _np.concatenate([pp[0:6], [_np.log(self.m_wet)]])
        xf = _np.concatenate([_np.zeros(6), [_np.log(self.m_dry)]])
        hover = _np.array([0, 0, 3.7114, 3.7114])
        return straightline_interpolate(x0, xf, N), straightline_interpolate(hover, hover, N), _np.array([75.0])

    def cost_terms(self):
        tx = _np.zeros(self.nx); tx[6] = -self.cost_weight  # maximise final mass
        return dict(Qu=_np.zeros(self.nu), lu=_np.zeros(self.nu), lx=_np.zeros(self.nx), tx=tx,
                    tp=_np.zeros(self.np), Qp=_np.zeros(self.np))

    def X(self, t, k):
        cg, sg = _np.cos(self.gamma_gs), _np.sin(self.gamma_gs)
        H = _np.zeros((4, self.nx))
        H[:, 0:3] = [[cg, 0, -sg], [-cg, 0, -sg], [0, cg, -sg], [0, -cg, -sg]]  # definition.jl:105-113
        h0 = _np.zeros(4)
        if t >= 1.0:
            # terminal node: the terminal condition pins r_N = 0, the APEX of the glide-slope cone, where all four rows
            # are active with non-unique multipliers (no strictly feasible point satisfies the terminal condition).
            # The rows are redundant there and are replaced by the trivially satisfied 0*x - 1 <= 0 (same row count).
            H = _np.zeros((4, self.nx)); h0 = -_np.ones(4)
        rows = [("NONPOS", H, _np.zeros((4, 1)), h0)]
        M = _np.zeros((4, self.nx)); M[1, 3] = M[2, 4] = M[3, 5] = 1
        rows.append(("SOC", M, _np.zeros((4, 1)), _np.array([self.v_max, 0, 0, 0])))       # :116
        e = _np.zeros((1, self.nx)); e[0, 6] = -1
        rows.append(("NONPOS", e, _np.zeros((1, 1)), _np.array([_np.log(self.m_dry)])))      # :130
        rows.append(("NONPOS", _np.zeros((1, self.nx)), _np.ones((1, 1)), _np.array([-self.tf_max])))
        rows.append(("NONPOS", _np.zeros((1, self.nx)), -_np.ones((1, 1)), _np.array([self.tf_min])))
        return rows

    def U(self, t, k):
        M = _np.zeros((4, self.nu)); M[0, 3] = 1; M[1, 0] = 1; M[2, 1] = 1; M[3, 2] = 1
        rows = [("SOC", M, _np.zeros((4, 1)), _np.zeros(4))]                                # :100
        e = _np.zeros((1, self.nu)); e[0, 3] = _np.cos(self.gamma_p); e[0, 2] = -1
        rows.append(("NONPOS", e, _np.zeros((1, 1)), _np.zeros(1)))                          # :103
        return rows

    def s(self, t, k, x, u, p):
        rmin, rmax = self.thrust_limits()
        return _np.array([rmin * _np.exp(-x[6]) - u[3], u[3] - rmax * _np.exp(-x[6])])

    def C(self, t, k, x, u, p):
        rmin, rmax = self.thrust_limits()
        C = _np.zeros((2, self.nx)); C[0, 6] = -rmin * _np.exp(-x[6]); C[1, 6] = rmax * _np.exp(-x[6])
        return C

    def D(self, t, k, x, u, p):
        D = _np.zeros((2, self.nu)); D[0, 3] = -1; D[1, 3] = 1
        return D

    def G(self, t, k, x, u, p):
        return _np.zeros((2, self.np))

    def gic(self, x, p, pp):
        return x - _np.concatenate([pp[0:6], [_np.log(self.m_wet)]])

    def H0(self, x, p, pp):
        return _np.eye(7)

    def K0(self, x, p, pp):
        return _np.zeros((7, 1))

    def gtc(self, x, p, pp):
        return x[0:6].copy()

    def Hf(self, x, p, pp):
        return _np.eye(7)[0:6]

    def Kf(self, x, p, pp):
        return _np.zeros((6, 1))


class DoubleIntegrator:
    """Builder-defined PTR version of test/examples/double_integrator:
    |u| in [1,2] with the non-convex part 1 - u^2 <= 0 in `s`, cost int u^2."""
    name = "double_integrator"
    nx, nu, np, ns, nic, ntc = 2, 1, 0, 1, 2, 2
    g, T, s_travel = 0.1, 10.0, 47.0

    def par(self):
        return _np.array([self.g, self.T])

    def nominal_pp(self):
        return _np.array([0.0, 0.0, self.s_travel, 0.0])

    def bbox(self):
        return (_np.array([[0.0, self.s_travel], [0.0, 2 * self.s_travel / self.T]]), _np.array([[-2.0, 2.0]]),
                _np.zeros((0, 2)))

    def guess(self, N, pp):
        # accelerate then brake: |u| >= 1 is non-convex, a one-signed guess can never brake
        u = _np.where(_np.arange(N) < N // 2, 1.5, -1.5).reshape(N, 1).astype(float)
        return straightline_interpolate(pp[0:2], pp[2:4], N), u, _np.zeros(0)

    def cost_terms(self):
        return dict(Qu=_np.array([1.0]), lu=_np.zeros(1), lx=_np.zeros(2), tx=_np.zeros(2), tp=_np.zeros(0),
                    Qp=_np.zeros(0))

    def X(self, t, k):
        return []

    def U(self, t, k):
        return [("NONPOS", _np.array([[1.0]]), _np.zeros((1, 0)), _np.array([-2.0])),
                ("NONPOS", _np.array([[-1.0]]), _np.zeros((1, 0)), _np.array([-2.0]))]

    def s(self, t, k, x, u, p):
        return _np.array([1 - u[0] ** 2])

    def C(self, t, k, x, u, p):
        return _np.zeros((1, 2))

    def D(self, t, k, x, u, p):
        return _np.array([[-2 * u[0]]])

    def G(self, t, k, x, u, p):
        return _np.zeros((1, 0))

    def gic(self, x, p, pp):
        return x - pp[0:2]

    def H0(self, x, p, pp):
        return _np.eye(2)

    def K0(self, x, p, pp):
        return _np.zeros((2, 0))

    def gtc(self, x, p, pp):
        return x - pp[2:4]

    def Hf(self, x, p, pp):
        return _np.eye(2)

    def Kf(self, x, p, pp):
        return _np.zeros((2, 0))


class Starship:
    """test/examples/starship_flip/{parameters,definition}.jl: the landing-flip problem (PTR / SCvx form).  The grid
    size N enters s(.) through the phase-switch test (definition.jl:705-712), so the model is built for a given N."""
    name = "starship"
    nx, nu, np, ns, nic, ntc = 8, 3, 10, 23, 7, 6      # ns = _common_s_sz = 7 + 2 nx (definition.jl:714)
    g0, m, ls = 9.81, 120e3, 50.0
    T_min1, T_max1 = 880e3, 2210e3
    delta_max = _np.deg2rad(10.0)
    rate_delay = 0.05
    tf_min, tf_max, tau_s = 0.0, 40.0, 0.5
    gamma_gs, theta_max2 = _np.deg2rad(27.0), _np.deg2rad(15.0)
    r0, v0, theta0 = _np.array([100.0, 600.0]), _np.array([0.0, -85.0]), _np.deg2rad(90.0)
    vf = _np.array([0.0, -0.1])

    def __init__(self, N=31, hs=100.0):
        self.N = N
        self.hs = hs        # the reference's guess generator overwrites traj.hs (definition.jl:181)
        self.T_min3, self.T_max3 = 3 * self.T_min1, 3 * self.T_max1
        self.deltadot_max = 2 * self.delta_max

    def par(self):
        return _np.array([float(self.N), float(self.hs)])

    def nominal_pp(self):
        return _np.concatenate([self.r0, self.v0, [self.theta0]])     # [r0 v0 theta0]

    def bbox(self):  # set_scale!, definition.jl:50-77
        xb = _np.array([[-100.0, 100.0], [0.0, self.r0[1]], [-10.0, 10.0], [self.v0[1], 0.0], [0.0, self.theta0],
                       _np.deg2rad([-10.0, 10.0]), [self.m - 1e3, self.m], [-self.delta_max, self.delta_max]])
        ub = _np.array([[self.T_min1, self.T_max3], [-self.delta_max, self.delta_max], [-self.deltadot_max, self.deltadot_max]])
        pb = _np.vstack([[[0.0, self.tf_max], [0.0, self.tf_max]], xb])
        return xb, ub, pb

    def guess(self, N, pp):
        """straight line between the boundary states (same rule as csrc/models/starship.hpp; the reference's bang-bang +
        LCvx guess, definition.jl:97-445, can be passed as a warm start)"""
        x0 = _np.array([pp[0], pp[1], pp[2], pp[3], pp[4], 0.0, 0.0, 0.0])
        xf = _np.array([0.0, 0.0, self.vf[0], self.vf[1], 0.0, 0.0, -3e3, 0.0])
        t = linrange(0.0, 1.0, N)
        x = (1.0 - t)[:, None] * x0[None, :] + t[:, None] * xf[None, :]
        u = _np.zeros((N, 3))
        u[:, 0] = _np.where(t <= self.tau_s, self.T_min3, self.m * self.g0)
        p = _np.concatenate([[10.0, 10.0], 0.5 * (x0 + xf)])
        return x, u, p

    def cost_terms(self):  # definition.jl:456-476
        tx = _np.zeros(self.nx); tx[6] = -1.0 / 10e3
        tp = _np.zeros(self.np); tp[2 + 1] = -0.3 / self.hs
        return dict(Qu=_np.zeros(self.nu), lu=_np.zeros(self.nu), lx=_np.zeros(self.nx), tx=tx, tp=tp, Qp=_np.zeros(self.np))

    def X(self, t, k):  # definition.jl:642-671
        e = _np.zeros((1, self.nx)); e[0, 3] = 1.0
        tsum = _np.zeros((1, self.np)); tsum[0, 0] = tsum[0, 1] = 1.0
        z = _np.zeros((1, self.nx))
        return [("NONPOS", e, _np.zeros((1, self.np)), _np.zeros(1)),
                ("NONPOS", z, tsum, _np.array([-self.tf_max])),
                ("NONPOS", z, -tsum, _np.array([self.tf_min]))]

    def U(self, t, k):  # definition.jl:673-701
        flip = t <= self.tau_s
        T_max, T_min = (self.T_max3, self.T_min3) if flip else (self.T_max1, self.T_min1)
        rows = []
        zp = _np.zeros((1, self.np))
        e = _np.zeros((1, self.nu)); e[0, 0] = 1.0
        rows.append(("NONPOS", e, zp, _np.array([-T_max])))
        rows.append(("NONPOS", -e, zp, _np.array([T_min])))
        d = _np.zeros((1, self.nu)); d[0, 1] = 1.0
        rows.append(("NONPOS", d, zp, _np.array([-self.delta_max])))      # L1 cone (delta_max, delta): |delta| <= delta_max
        rows.append(("NONPOS", -d, zp, _np.array([-self.delta_max])))
        return rows

    def _phase_switch(self, t):
        dt = 1.0 / (self.N - 1)
        return (self.tau_s - dt) + 1e-3 <= t <= self.tau_s + 1e-3

    def _phase2(self, t):
        return self._phase_switch(t) or t > self.tau_s

    def s(self, t, k, x, u, p):  # definition.jl:723-752
        s = _np.zeros(self.ns)
        dd, de, dedot = x[7], u[1], u[2]
        s[0] = (de - dd) - dedot * self.rate_delay
        s[1] = dedot * self.rate_delay - (de - dd)
        s[2] = dedot - self.deltadot_max
        s[3] = -self.deltadot_max - dedot
        s[4] = _np.linalg.norm(x[0:2]) * _np.cos(self.gamma_gs) - x[1]
        if self._phase_switch(t):
            s[5:13] = p[2:10] - x
            s[13:21] = x - p[2:10]
        if self._phase2(t):
            s[-2] = x[4] - self.theta_max2
            s[-1] = -self.theta_max2 - x[4]
        return s

    def C(self, t, k, x, u, p):  # definition.jl:753-776
        C = _np.zeros((self.ns, self.nx))
        C[0, 7] = -1.0; C[1, 7] = 1.0
        nr = _np.linalg.norm(x[0:2])
        gr = _np.zeros(2) if nr < _np.sqrt(_np.finfo(float).eps) else x[0:2] / nr
        C[4, 0:2] = gr * _np.cos(self.gamma_gs) - _np.array([0.0, 1.0])
        if self._phase_switch(t):
            C[5:13] = -_np.eye(8); C[13:21] = _np.eye(8)
        if self._phase2(t):
            C[-2, 4] = 1.0; C[-1, 4] = -1.0
        return C

    def D(self, t, k, x, u, p):  # definition.jl:777-788
        D = _np.zeros((self.ns, self.nu))
        D[0, 1] = 1.0; D[0, 2] = -self.rate_delay; D[1, 1] = -1.0; D[1, 2] = self.rate_delay
        D[2, 2] = 1.0; D[3, 2] = -1.0
        return D

    def G(self, t, k, x, u, p):  # definition.jl:789-798
        G = _np.zeros((self.ns, self.np))
        if self._phase_switch(t):
            G[5:13, 2:10] = _np.eye(8); G[13:21, 2:10] = -_np.eye(8)
        return G

    def gic(self, x, p, pp):  # definition.jl:814-842
        return x[0:7] - _np.concatenate([pp[0:5], [0.0, 0.0]])

    def H0(self, x, p, pp):
        return _np.eye(7, 8)

    def K0(self, x, p, pp):
        return _np.zeros((7, self.np))

    def gtc(self, x, p, pp):  # definition.jl:843-870
        return x[0:6] - _np.concatenate([[0.0, 0.0], self.vf, [0.0, 0.0]])

    def Hf(self, x, p, pp):
        return _np.eye(6, 8)

    def Kf(self, x, p, pp):
        return _np.zeros((6, self.np))


class _Quat:
    """src/utils/quaternion.jl: q = (v, w) with the scalar LAST in vector form (:33-36, 453-456)."""

    def __init__(self, v, w):
        self.v, self.w = _np.asarray(v, float), float(w)

    @staticmethod
    def axis_angle(alpha, a):                       # :112-124
        a = _np.asarray(a, float) / _np.linalg.norm(a)
        return _Quat(a * _np.sin(alpha / 2), _np.cos(alpha / 2))

    def skew(self, side="L"):                       # :190-198
        S = _np.zeros((4, 4))
        sk = _np.array([[0, -self.v[2], self.v[1]], [self.v[2], 0, -self.v[0]], [-self.v[1], self.v[0], 0]])
        S[0:3, 0:3] = self.w * _np.eye(3) + (1 if side == "L" else -1) * sk
        S[0:3, 3] = self.v; S[3, 0:3] = -self.v; S[3, 3] = self.w
        return S

    def vec(self):
        return _np.concatenate([self.v, [self.w]])

    def __mul__(self, o):                           # :211-214
        r = self.skew() @ o.vec()
        return _Quat(r[0:3], r[3])

    def conj(self):                                 # :257-260
        return _Quat(-self.v, self.w)

    def log(self):                                  # :277-282
        n = _np.linalg.norm(self.v)
        return 2 * _np.arctan2(n, self.w), self.v / n


def _hyperrectangle(offset, width, height, depth, yaw=0.0, pitch=0.0, roll=0.0):
    """Hyperrectangle(offset, width, height, depth; yaw, pitch, roll), src/utils/hyperrectangle.jl:102-150 -> (c, s) with
    the set {r : |(r - c) / s|_inf <= 1}."""
    lo = _np.array([-width / 2, -height / 2, 0.0]); hi = _np.array([width / 2, height / 2, depth])
    c_, s_ = (lambda a: _np.cos(_np.deg2rad(a))), (lambda a: _np.sin(_np.deg2rad(a)))
    Rz = _np.array([[c_(yaw), -s_(yaw), 0], [s_(yaw), c_(yaw), 0], [0, 0, 1]])
    Ry = _np.array([[c_(pitch), 0, s_(pitch)], [0, 1, 0], [-s_(pitch), 0, c_(pitch)]])
    Rx = _np.array([[1, 0, 0], [0, c_(roll), -s_(roll)], [0, s_(roll), c_(roll)]])
    R = Rz @ Ry @ Rx
    lr, ur = R @ lo, R @ hi
    l, u = _np.minimum(lr, ur) + offset, _np.maximum(lr, ur) + offset
    return (u + l) / 2, (u - l) / 2


class Freeflyer:
    """test/examples/freeflyer/{parameters,definition}.jl.

    Freeflyer()   -- what discretize! and the initial guess need (np = 1: the room-SDF slacks never enter the dynamics,
                     oracle/scp_oracle.c).
    Freeflyer(N)  -- the whole trajectory problem (SCvx form, `algo = :scvx`), p = [t_f; delta] with delta[i, k] one slack
                     per room i and node k (np = 1 + 6 N, parameters.jl:121-128); np_dyn = 1 tells ptr_ref.discretize which
                     parameters the dynamics see."""
    name = "freeflyer"
    nx, nu = 13, 6
    np_dyn = 1
    tf_min, tf_max = 60.0, 200.0
    v_max, w_max, T_max, M_max = 0.4, _np.deg2rad(1.0), 20e-3, 1e-4          # parameters.jl:135-138
    gamma, hom, eps_sdf = 0.0, 50.0, 1e-4                                      # parameters.jl:168-170
    obs_c = _np.array([[8.5, -0.15, 5.0], [11.2, 1.84, 5.0], [11.3, 3.8, 4.8]])     # Ellipsoid(I / 0.3, c), :95-101
    obs_h = 1.0 / 0.3
    n_obs, n_iss = 3, 6

    def __init__(self, N=None):
        self.N = N
        self.np = 1 if N is None else 1 + self.n_iss * N
        self.ns, self.nic, self.ntc = self.n_obs + 1, 13, 13
        z = 4.75
        rooms = [_hyperrectangle([6.0, 0.0, z], 1.0, 1.0, 1.5, pitch=90.0),                  # parameters.jl:102-109
                 _hyperrectangle([7.5, 0.0, z], 2.0, 2.0, 4.0, pitch=90.0),
                 _hyperrectangle([11.5, 0.0, z], 1.25, 1.25, 0.5, pitch=90.0),
                 _hyperrectangle([10.75, -1.0, z], 1.5, 1.5, 1.5, yaw=-90.0, pitch=90.0),
                 _hyperrectangle([10.75, 1.0, z], 1.5, 1.5, 1.5, yaw=90.0, pitch=90.0),
                 _hyperrectangle([10.75, 2.5, z], 2.5, 2.5, 4.5, yaw=90.0, pitch=90.0)]
        self.room_c = _np.array([r[0] for r in rooms]); self.room_s = _np.array([r[1] for r in rooms])

    def par(self):
        return default_params("freeflyer")

    def id_delta(self, k):
        """0-based indices into p of delta[:, k] (k 1-based): reshape(p[id_delta], n_iss, :) is column-major."""
        return 1 + self.n_iss * (k - 1) + _np.arange(self.n_iss)

    def nominal_pp(self):                           # parameters.jl:160-167
        q0 = _Quat.axis_angle(_np.deg2rad(-40), [0.0, 1.0, 1.0]).vec()
        qf = _Quat.axis_angle(_np.deg2rad(0), [0.0, 0.0, 1.0]).vec()
        return _np.concatenate([[6.5, -0.2, 5.0], [0.035, 0.035, 0.0], q0, _np.zeros(3), [11.3, 6.0, 4.5], _np.zeros(3), qf,
                                _np.zeros(3)])

    def bbox(self):
        """set_scale! (definition.jl:47-66): r and p advised; v, w, T, M from the LPs over their norm balls; q free."""
        pp = self.nominal_pp()
        r0, rf = pp[0:3], pp[13:16]
        xb = _np.vstack([_np.stack([_np.minimum(r0, rf), _np.maximum(r0, rf)], axis=1), _np.tile([[-self.v_max, self.v_max]], (3, 1)),
                         _np.tile([[0.0, 1.0]], (4, 1)), _np.tile([[-self.w_max, self.w_max]], (3, 1))])
        ub = _np.vstack([_np.tile([[-self.T_max, self.T_max]], (3, 1)), _np.tile([[-self.M_max, self.M_max]], (3, 1))])
        pb = _np.vstack([[[self.tf_min, self.tf_max]], _np.tile([[-100.0, 1.0]], (self.np - 1, 1))])
        return xb, ub, pb

    def guess(self, N, pp):
        """set_guess!, definition.jl:84-186 (line by line)."""
        r0, q0 = _np.asarray(pp[0:3], float), _Quat(pp[6:9], pp[9])
        rf, qf = _np.asarray(pp[13:16], float), _Quat(pp[19:22], pp[22])
        flight_time = 0.5 * (self.tf_min + self.tf_max)
        x = _np.zeros((13, N))
        speed = _np.linalg.norm(rf - r0, 1) / flight_time
        times = straightline_interpolate([0.0], [flight_time], N)[:, 0]
        leg = _np.abs(rf - r0) / speed
        cumul = _np.cumsum(leg)
        for k in range(N):
            tk = times[k]
            # The reference leaves r[:, k], v[:, k] UNINITIALISED (RealMatrix(undef, ...), definition.jl:105) when no leg claims
            # the node -- which happens at the last node whenever the cumulative leg times sum to a hair less than the flight
            # time (17 of 128 instances with +-3 mm spread).  Defined here as the product defines it: the goal position at rest.
            x[0:3, k] = rf
            for i in range(3):
                if tk <= cumul[i]:
                    t0 = cumul[i - 1] if i > 0 else 0.0
                    tf = cumul[i]
                    a = r0.copy(); a[0:i] = rf[0:i]
                    b = a.copy(); b[i] = rf[i]
                    tc = max(t0, min(tf, tk))
                    c = (tf - tc) / (tf - t0)
                    x[0:3, k] = c * a + (1 - c) * b                   # linterp(tk, hcat(r0, rf), [t0, tf])
                    d = b - a
                    x[3:6, k] = speed * d / _np.linalg.norm(d)
                    break
        for k in range(N):
            mix = k / (N - 1)
            tau = max(0.0, min(1.0, mix))
            dq = q0.conj() * qf
            da, dax = dq.log()
            x[6:10, k] = (q0 * _Quat.axis_angle(tau * da, dax)).vec()
        rot_ang, rot_ax = (qf * q0.conj()).log()
        x[10:13, :] = (rot_ang / flight_time * rot_ax)[:, None]
        p = _np.zeros(self.np)
        p[0] = flight_time
        if self.np > 1:                                              # delta[i, k] = 1 - |(r_k - c_i) / s_i|_inf  (:166-172)
            for k in range(1, N + 1):
                p[self.id_delta(k)] = 1.0 - _np.abs((x[0:3, k - 1][None, :] - self.room_c) / self.room_s).max(axis=1)
        return x.T.copy(), _np.zeros((N, 6)), p

    # ---- the trajectory problem (N given) ----
    def cost_terms(self):           # set_cost!, definition.jl:188-222 (algo = :scvx)
        tp = _np.zeros(self.np); tp[1:] = -self.eps_sdf
        Qp = _np.zeros(self.np); Qp[0] = self.gamma / self.tf_max ** 2
        Qu = _np.concatenate([_np.full(3, (1 - self.gamma) / self.T_max ** 2), _np.full(3, (1 - self.gamma) / self.M_max ** 2)])
        return dict(Qu=Qu, lu=_np.zeros(6), lx=_np.zeros(13), tx=_np.zeros(13), tp=tp, Qp=Qp)

    def X(self, t, k):              # problem_set_X!, definition.jl:288-350
        z = _np.zeros((1, self.np))
        rows = []
        M = _np.zeros((4, 13)); M[1, 3] = M[2, 4] = M[3, 5] = 1.0
        rows.append(("SOC", M, _np.zeros((4, self.np)), _np.array([self.v_max, 0, 0, 0])))
        M = _np.zeros((4, 13)); M[1, 10] = M[2, 11] = M[3, 12] = 1.0
        rows.append(("SOC", M, _np.zeros((4, self.np)), _np.array([self.w_max, 0, 0, 0])))
        e = z.copy(); e[0, 0] = 1.0
        rows.append(("NONPOS", _np.zeros((1, 13)), e, _np.array([-self.tf_max])))
        rows.append(("NONPOS", _np.zeros((1, 13)), -e, _np.array([self.tf_min])))
        idd = self.id_delta(k)
        for i in range(self.n_iss):                                  # (1 - delta_ik, (r - c_i) ./ s_i) in LINF
            M = _np.zeros((4, 13)); Mp = _np.zeros((4, self.np)); m0 = _np.zeros(4)
            Mp[0, idd[i]] = -1.0; m0[0] = 1.0
            for j in range(3):
                M[1 + j, j] = 1.0 / self.room_s[i, j]; m0[1 + j] = -self.room_c[i, j] / self.room_s[i, j]
            rows.append(("LINF", M, Mp, m0))
        return rows

    def U(self, t, k):              # problem_set_U!, definition.jl:352-376
        rows = []
        for o, bound in ((0, self.T_max), (3, self.M_max)):
            M = _np.zeros((4, 6)); M[1, o] = M[2, o + 1] = M[3, o + 2] = 1.0
            rows.append(("SOC", M, _np.zeros((4, self.np)), _np.array([bound, 0, 0, 0])))
        return rows

    def _lse(self, delta):          # logsumexp(delta; t = hom) and its gradient, src/utils/helper.jl:623-651
        a = _np.max(self.hom * delta)
        ex = _np.exp(self.hom * delta - a)
        return (a + _np.log(ex.sum())) / self.hom, ex / ex.sum()

    def s(self, t, k, x, u, p):     # definition.jl:381-398
        out = _np.zeros(self.ns)
        for i in range(self.n_obs):
            out[i] = 1.0 - self.obs_h * _np.linalg.norm(x[0:3] - self.obs_c[i])
        out[-1] = -self._lse(p[self.id_delta(k)])[0]
        return out

    def C(self, t, k, x, u, p):     # :399-411, ellipsoid.jl:99-118: -grad ||H (r - c)||
        C = _np.zeros((self.ns, 13))
        for i in range(self.n_obs):
            d = x[0:3] - self.obs_c[i]
            C[i, 0:3] = -self.obs_h * d / _np.linalg.norm(d)
        return C

    def D(self, t, k, x, u, p):
        return _np.zeros((self.ns, 6))

    def G(self, t, k, x, u, p):     # :412-428
        G = _np.zeros((self.ns, self.np))
        G[-1, self.id_delta(k)] = -self._lse(p[self.id_delta(k)])[1]
        return G

    def gic(self, x, p, pp):        # set_bcs!, definition.jl:454-500
        return x - _np.concatenate([pp[0:3], pp[3:6], pp[6:10], pp[10:13]])

    def H0(self, x, p, pp):
        return _np.eye(13)

    def K0(self, x, p, pp):
        return _np.zeros((13, self.np))

    def gtc(self, x, p, pp):
        return x - _np.concatenate([pp[13:16], pp[16:19], pp[19:23], pp[23:26]])

    def Hf(self, x, p, pp):
        return _np.eye(13)

    def Kf(self, x, p, pp):
        return _np.zeros((13, self.np))


MODELS = {m.name: m for m in (DoubleIntegrator, Quadrotor, RocketLanding, Starship, Freeflyer)}
