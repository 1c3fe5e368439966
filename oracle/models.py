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
        return _np.array([self.g])

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
        return default_params("rocket_landing")

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


MODELS = {m.name: m for m in (DoubleIntegrator, Quadrotor, RocketLanding)}
