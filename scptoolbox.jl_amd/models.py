"""Host-side registry of the compiled device models.

The reference lets the user pass Julia closures (`problem_set_dynamics!` etc.,
src/parser/problem.jl:432-450); closures cannot cross the C ABI, so a model is
selected by id and configured by a POD parameter blob.  This
module holds, per model, what the *host* needs: ids, nominal parameter blobs,
scaling boxes and initial-guess generators.  All arithmetic on the hot path
happens in the HIP library.
"""
import numpy as np

MODEL_IDS = {"double_integrator": 0, "quadrotor": 1, "rocket_landing": 2, "starship": 3, "freeflyer": 4}


def linrange(a, b, n):
    """Julia LinRange(a,b,n) (Base `lerpi`)."""
    j = np.arange(n, dtype=np.float64)
    t = j / (n - 1)
    return (1.0 - t) * a + t * b


def straightline_interpolate(v0, vf, N):
    """src/utils/helper.jl:203-219: linterp between two end points on LinRange(0,1,N).

    Returns [N, nv] (C order == Julia [nv, N] column-major)."""
    v0 = np.asarray(v0, dtype=np.float64)
    vf = np.asarray(vf, dtype=np.float64)
    t = linrange(0.0, 1.0, N)
    c = (1.0 - t) / (1.0 - 0.0)  # helper.jl:114 with t_grid = [0, 1]
    return c[:, None] * v0[None, :] + (1.0 - c)[:, None] * vf[None, :]


class NativeModel:
    """Description of one compiled model (the `mdl` of a TrajectoryProblem)."""

    name = None
    OVERRIDES = None     # names a subclass accepts as keyword overrides of its constants (None: unchecked)

    def __init__(self, **overrides):
        if self.OVERRIDES is not None:
            unknown = sorted(set(overrides) - set(self.OVERRIDES))
            if unknown:     # a misspelt constant would otherwise be ignored and the defaults used silently
                raise ValueError("%s: unknown model constant(s) %s; known: %s" % (self.name, unknown, sorted(self.OVERRIDES)))
        self.opts = dict(overrides)

    # -- to be provided by subclasses --
    def par(self):
        raise NotImplementedError

    def scale_advice(self):
        """Returns (x_bbox, u_bbox, p_bbox) arrays [n,2]: the bounding boxes that
        `compute_scaling` (src/solvers/scp.jl:376-517) obtains either from user
        advice or from its min/max LPs over the convex sets X and U."""
        raise NotImplementedError

    def guess(self, N, pp):
        raise NotImplementedError

    def nominal_pp(self):
        raise NotImplementedError


class DoubleIntegratorModel(NativeModel):
    """test/examples/double_integrator/parameters.jl:48-83 as a fixed-duration PTR
    problem (builder-defined): x=[pos,vel], u=[accel], np=0."""
    name = "double_integrator"
    nx, nu, np = 2, 1, 0
    OVERRIDES = ("g", "T", "s")

    def par(self):
        return np.array([self.opts.get("g", 0.1), self.opts.get("T", 10.0)])

    def nominal_pp(self):
        # per-problem data: [x0(2), xf(2)]   (definition.jl:56-64: x0=0, xf=[s,0], s=47)
        return np.array([0.0, 0.0, self.opts.get("s", 47.0), 0.0])

    def scale_advice(self):
        s = self.opts.get("s", 47.0)
        return (np.array([[0.0, s], [0.0, 2 * s / self.opts.get("T", 10.0)]]),
                np.array([[-2.0, 2.0]]), np.zeros((0, 2)))

    def guess(self, N, pp):
        x = straightline_interpolate(pp[0:2], pp[2:4], N)
        # accelerate then brake: |u| >= 1 is non-convex, a one-signed guess can never brake
        u = np.where(np.arange(N) < N // 2, 1.5, -1.5).reshape(N, 1).astype(np.float64)
        return x, u, np.zeros(0)


class QuadrotorModel(NativeModel):
    """test/examples/quadrotor/{parameters,definition}.jl.  Every constant of `QuadrotorProblem` (parameters.jl:96-130) is
    data in the parameter blob `par()`; overrides: g, u_min, u_max, tilt_max (rad), tf_min, tf_max, gamma,
    obstacles = [(diag(H), c), (diag(H), c)]."""
    name = "quadrotor"
    nx, nu, np = 6, 4, 1
    OVERRIDES = ("g", "u_min", "u_max", "tilt_max", "tf_min", "tf_max", "gamma", "obstacles")

    def __init__(self, **overrides):
        super().__init__(**overrides)
        o = self.opts
        self.g = float(o.get("g", 9.81))                                                              # parameters.jl:109
        self.u_min, self.u_max = float(o.get("u_min", 0.6)), float(o.get("u_max", 23.2))              # :110-111
        self.tilt_max = float(o.get("tilt_max", np.deg2rad(60)))                                      # :112
        self.tf_min, self.tf_max, self.gamma = float(o.get("tf_min", 0.0)), float(o.get("tf_max", 2.5)), float(o.get("gamma", 0.0))
        self.obstacles = o.get("obstacles", [([2.0, 2.0, 0.0], [1.0, 2.0, 0.0]), ([1.5, 1.5, 0.0], [2.0, 5.0, 0.0])])   # :114-118
        assert len(self.obstacles) == 2

    def par(self):
        obs = [v for H, c in self.obstacles for v in list(H) + list(c)]
        return np.array([self.g, self.u_min, self.u_max, self.tilt_max, self.tf_min, self.tf_max, self.gamma] + obs, dtype=float)

    def nominal_pp(self):
        # per-problem data [r0 v0 rf vf]  (parameters.jl:120-126)
        return np.array([0, 0, 0, 0, 0, 0, 2.5, 6.0, 0, 0, 0, 0], dtype=np.float64)

    def scale_advice(self):
        # X is absent -> every state LP is unbounded (DUAL_INFEASIBLE) and the box
        # stays [0,1] (scp.jl:393-398,470-477).  U: u_min<=sigma<=u_max, ||a||<=sigma,
        # sigma*cos(tilt)<=a3 (definition.jl:188-253) -> the LP optima are analytic.
        lat = self.u_max * np.sin(self.tilt_max)
        ub = np.array([[-lat, lat], [-lat, lat], [self.u_min * np.cos(self.tilt_max), self.u_max], [self.u_min, self.u_max]])
        xb = np.tile(np.array([[0.0, 1.0]]), (6, 1))
        pb = np.array([[self.tf_min, self.tf_max]])  # advised, definition.jl:48-58
        return xb, ub, pb

    def guess(self, N, pp):
        # definition.jl:60-90
        x = straightline_interpolate(pp[0:6], pp[6:12], N)
        hover = np.array([0.0, 0.0, self.g, self.g])
        u = straightline_interpolate(hover, hover, N)
        p = np.array([0.5 * (self.tf_min + self.tf_max)])
        return x, u, p


class RocketLandingModel(NativeModel):
    """Mars rocket landing (test/examples/rocket_landing/parameters.jl:77-146) as a
    free-final-time PTR problem (builder-defined).  Every constant is data in the parameter blob
    `par()`; overrides: m_dry, m_wet, rho_min, rho_max, gamma_gs (rad), gamma_p (rad), v_max, tf_min, tf_max, cost_weight,
    g (3-vector), omega (3-vector), alpha."""
    name = "rocket_landing"
    nx, nu, np = 7, 4, 1
    OVERRIDES = ("m_dry", "m_wet", "rho_min", "rho_max", "gamma_gs", "gamma_p", "v_max", "tf_min", "tf_max", "cost_weight", "g", "omega",
                 "alpha")

    def __init__(self, **overrides):
        super().__init__(**overrides)
        o = self.opts
        self.m_dry, self.m_wet = float(o.get("m_dry", 1505.0)), float(o.get("m_wet", 1905.0))          # parameters.jl:86-87
        n_eng, phi, T_max = 6, 27 * np.pi / 180, 3.1e3                                                  # :88-93
        self.rho_min = float(o.get("rho_min", n_eng * 0.3 * T_max * np.cos(phi)))
        self.rho_max = float(o.get("rho_max", n_eng * 0.8 * T_max * np.cos(phi)))
        self.gamma_gs, self.gamma_p = float(o.get("gamma_gs", 86 * np.pi / 180)), float(o.get("gamma_p", 40 * np.pi / 180))
        self.v_max = float(o.get("v_max", 500 * 1e3 / 3600))
        self.tf_min, self.tf_max = float(o.get("tf_min", 40.0)), float(o.get("tf_max", 120.0))
        self.cost_weight = float(o.get("cost_weight", 1.0))
        th = 30 * np.pi / 180
        T_sid = 24.6229 * 3600
        self.g = np.asarray(o.get("g", [0.0, 0.0, -3.7114]), float)
        self.omega = np.asarray(o.get("omega", (2 * np.pi / T_sid) * np.array([np.cos(th), 0.0, np.sin(th)])), float)
        Isp, ge = 225.0, 9.807
        self.alpha = float(o.get("alpha", 1 / (Isp * ge * np.cos(phi))))

    def par(self):
        return np.concatenate([self.g, self.omega, [self.alpha, self.m_dry, self.m_wet, self.rho_min, self.rho_max, self.gamma_gs,
                                                    self.gamma_p, self.v_max, self.tf_min, self.tf_max, self.cost_weight]])

    def nominal_pp(self):
        # per-problem data [r0(3) v0(3)]   (parameters.jl:102-103)
        return np.array([2000.0, 0.0, 1500.0, 80.0, 30.0, -75.0])

    def thrust_limits(self):
        return self.rho_min, self.rho_max

    def scale_advice(self):
        xb = np.array([[-2500.0, 2500.0], [-2500.0, 2500.0], [0.0, 2500.0],
                       [-self.v_max, self.v_max], [-self.v_max, self.v_max], [-self.v_max, self.v_max],
                       [np.log(self.m_dry), np.log(self.m_wet)]])
        a_max = self.rho_max / self.m_dry
        ub = np.array([[-a_max, a_max], [-a_max, a_max], [0.0, a_max], [0.0, a_max]])
        pb = np.array([[self.tf_min, self.tf_max]])
        return xb, ub, pb

    def guess(self, N, pp):
        x0 = np.concatenate([pp[0:6], [np.log(self.m_wet)]])
        xf = np.concatenate([np.zeros(6), [np.log(self.m_dry)]])
        x = straightline_interpolate(x0, xf, N)
        g = -self.g[2]
        hover = np.array([0.0, 0.0, g, g])
        u = straightline_interpolate(hover, hover, N)
        return x, u, np.array([75.0])


class StarshipModel(NativeModel):
    """Starship landing flip (test/examples/starship_flip/{parameters,definition}.jl): x = [r(2) v(2) theta omega m delta_d],
    u = [T delta delta_dot], p = [t1 t2 xs(8)].  State-dependent Jacobians; no stage-structured fast path: PTR and SCvx run
    through the generic conic path.  The model's non-convex constraints need the grid size (phase-switch node,
    definition.jl:705-712): `bind(pars)` is called by SCPProblem before the parameter blob is read."""
    name = "starship"

    # parameters.jl:99-212 -- every value is data in the parameter blob `par()` and can be overridden by keyword
    # (StarshipModel(m=..., T_max1=..., gamma_gs=...)); host scaling, guess and formulation read the same values
    DEFAULTS = dict(g0=9.81, m=120e3, rs=4.5, ls=50.0, lcg=None, lcp=None, J=None, CD=None, T_min1=880e3, T_max1=2210e3,
                    T_min3=None, T_max3=None, alpha_e=-1.0 / (330.0 * 9.81), delta_max=float(np.deg2rad(10.0)),
                    deltadot_max=None, rate_delay=0.05, tf_min=0.0, tf_max=40.0, tau_s=0.5, gamma_gs=float(np.deg2rad(27.0)),
                    theta_max2=float(np.deg2rad(15.0)), vf_x=0.0, vf_y=-0.1, cost_alt=0.3, cost_mass=10e3, v_terminal=85.0)
    OVERRIDES = tuple(DEFAULTS) + ("hs", "N")
    nx, nu, np = 8, 3, 10      # (class attribute `np` shadows numpy below this line inside the class body only)
    hs = 100.0      # altitude normalisation of the terminal cost (parameters.jl:190); `reference_guess` overwrites it

    def __init__(self, **overrides):
        super().__init__(**overrides)
        c = dict(self.DEFAULTS)
        c.update({k: v for k, v in self.opts.items() if k in c})
        derived = dict(lcg=0.4 * c["ls"], lcp=0.45 * c["ls"], J=1.0 / 12.0 * c["m"] * (6.0 * c["rs"] ** 2 + c["ls"] ** 2),
                       CD=c["m"] * c["g0"] / c["v_terminal"] ** 2 * 1.2, T_min3=3.0 * c["T_min1"], T_max3=3.0 * c["T_max1"],
                       deltadot_max=2.0 * c["delta_max"])        # parameters.jl:127-146: quantities defined through others
        for k, v in derived.items():
            if c[k] is None:
                c[k] = v
        for k, v in c.items():
            setattr(self, k, float(v))

    def bind(self, pars):
        self.N = int(pars.N)

    def par(self):
        if getattr(self, "N", None) is None:
            self.N = int(self.opts.get("N", 31))
        return np.array([float(self.N), float(self.opts.get("hs", self.hs)), self.g0, self.m, self.lcg, self.lcp, self.J, self.CD,
                         self.T_min1, self.T_max1, self.T_min3, self.T_max3, self.alpha_e, self.delta_max, self.deltadot_max,
                         self.rate_delay, self.tf_min, self.tf_max, self.tau_s, self.gamma_gs, self.theta_max2, self.vf_x,
                         self.vf_y, self.cost_alt, self.cost_mass])

    def reference_guess(self, N, pp=None, device=0):
        """The reference's own initial guess (bang-bang flip + convex terminal descent, definition.jl:97-445), evaluated by the
        library's kernels (csrc/starship_guess.hpp through scp_guess_batch_host: flip simulation, the 31 candidate descent programs
        as one conic batch, reconstruction).  Returns (x, u, p) and sets `self.hs` to the switch altitude like the reference does
        (:181) -- call it BEFORE `create` so that the cost sees the same normalisation.  (The host formulation that rounds 2-4 kept
        here is the oracle's: oracle/starship_guess.py.)"""
        from . import scp as _scp

        class _P:      # the handle needs a grid only (discretize! / subproblem parameters are not used by the guess kernels)
            pass
        pars = _P()
        pars.N, pars.Nsub, pars.disc_method, pars.feas_tol = int(N), 10, _scp.FOH, 5e-3
        traj = type("T", (), {"mdl": self})()
        pbm = _scp.SCPProblem(pars, traj, batch_capacity=1, device=device)
        try:
            ppa = np.asarray(self.nominal_pp() if pp is None else pp, float)[None]
            x, u, p = _scp.device_guess(pbm, ppa)
            if _scp.device_guess_failures(pbm):
                raise ArithmeticError("could not find a terminal descent time of flight")      # definition.jl:415-419
        finally:
            pbm.close()
        self.hs = float(p[0, 3])              # traj.hs = dot(xs[r], ey), p = [t1; t2; xs]
        return x[0], u[0], p[0]

    def nominal_pp(self):
        # per-problem data [r0(2) v0(2) theta0]  (parameters.jl:181-184)
        return np.array([100.0, 600.0, 0.0, -85.0, np.deg2rad(90.0)])

    def scale_advice(self):
        # set_scale!, definition.jl:50-77 (every component is advised: no scaling LPs are solved)
        r0y, v0y, th0 = 600.0, -85.0, np.deg2rad(90.0)
        xb = np.array([[-100.0, 100.0], [0.0, r0y], [-10.0, 10.0], [v0y, 0.0], [0.0, th0], np.deg2rad([-10.0, 10.0]),
                       [self.m - 1e3, self.m], [-self.delta_max, self.delta_max]])
        ub = np.array([[self.T_min1, 3 * self.T_max1], [-self.delta_max, self.delta_max], [-2 * self.delta_max, 2 * self.delta_max]])
        pb = np.vstack([[[0.0, self.tf_max], [0.0, self.tf_max]], xb])
        return xb, ub, pb

    def guess(self, N, pp):
        x0 = np.array([pp[0], pp[1], pp[2], pp[3], pp[4], 0.0, 0.0, 0.0])
        xf = np.array([0.0, 0.0, self.vf_x, self.vf_y, 0.0, 0.0, -3e3, 0.0])
        t = linrange(0.0, 1.0, N)
        x = (1.0 - t)[:, None] * x0[None, :] + t[:, None] * xf[None, :]
        u = np.zeros((N, 3))
        u[:, 0] = np.where(t <= self.tau_s, self.T_min3, self.m * self.g0)
        return x, u, np.concatenate([[10.0, 10.0], 0.5 * (x0 + xf)])


def quat_mul(a, b):
    """q * p for quaternions [v; w] (scalar last), src/utils/quaternion.jl:190-214."""
    av, aw, bv, bw = a[:3], a[3], b[:3], b[3]
    return np.concatenate([aw * bv + bw * av + np.cross(av, bv), [aw * bw - av @ bv]])


def quat_from_axis_angle(alpha, axis):
    """Quaternion(alpha, a), quaternion.jl:112-124."""
    a = np.asarray(axis, float) / np.linalg.norm(axis)
    return np.concatenate([a * np.sin(alpha / 2), [np.cos(alpha / 2)]])


def quat_log(q):
    """Log(q) -> (angle, axis), quaternion.jl:277-282."""
    nv = np.linalg.norm(q[:3])
    return 2 * np.arctan2(nv, q[3]), q[:3] / nv


def slerp_interpolate(q0, q1, tau):
    """quaternion.jl:483-490."""
    tau = max(0.0, min(1.0, tau))
    q0c = np.concatenate([-q0[:3], [q0[3]]])
    ang, ax = quat_log(quat_mul(q0c, q1))
    return quat_mul(q0, quat_from_axis_angle(tau * ang, ax))


def hyperrectangle(offset, width, height, depth, yaw=0.0, pitch=0.0, roll=0.0):
    """Hyperrectangle(offset, width, height, depth; yaw, pitch, roll) (src/utils/hyperrectangle.jl:102-150) -> (c, s) of the
    scaling x = s .* y + c that maps the box onto {y : |y|_inf <= 1} (:26-54)."""
    lo = np.array([-width / 2, -height / 2, 0.0]); hi = np.array([width / 2, height / 2, depth])
    c_, s_ = (lambda a: np.cos(np.deg2rad(a))), (lambda a: np.sin(np.deg2rad(a)))
    Rz = np.array([[c_(yaw), -s_(yaw), 0], [s_(yaw), c_(yaw), 0], [0, 0, 1]])
    Ry = np.array([[c_(pitch), 0, s_(pitch)], [0, 1, 0], [-s_(pitch), 0, c_(pitch)]])
    Rx = np.array([[1, 0, 0], [0, c_(roll), -s_(roll)], [0, s_(roll), c_(roll)]])
    R = Rz @ Ry @ Rx
    lr, ur = R @ lo, R @ hi
    l, u = np.minimum(lr, ur) + np.asarray(offset, float), np.maximum(lr, ur) + np.asarray(offset, float)
    return (u + l) / 2, (u - l) / 2


class FreeflyerModel(NativeModel):
    """test/examples/freeflyer/{parameters,definition}.jl.  The parameter vector is p = [t_f; delta] with one room-SDF
    slack per room and node (np = 1 + 6 N, parameters.jl:121-128): one GLOBAL parameter and six NODE parameters in the
    library's terms (include/scp_mi355x.h, scp_model_info) -- its length is known once the grid is (`bind(pars)`, called by
    SCPProblem).  Everything of `FreeFlyerProblem` (vehicle, trajectory, environment: parameters.jl:86-192) is data in the
    parameter blob `par()`; overrides: m, J, v_max, w_max, T_max, M_max, tf_min, tf_max, gamma, hom, eps_sdf,
    obstacles = [(h, c)] (3 ellipsoids H = h I), rooms = [(c, s)] (6 boxes)."""
    name = "freeflyer"
    nx, nu = 13, 6
    OVERRIDES = ("N", "m", "J", "v_max", "w_max", "T_max", "M_max", "tf_min", "tf_max", "gamma", "hom", "eps_sdf", "obstacles", "rooms")
    np_glob, np_node = 1, 6
    n_obs, n_iss = 3, 6

    def __init__(self, **overrides):
        super().__init__(**overrides)
        self.N = overrides.get("N")
        o = self.opts
        self.tf_min, self.tf_max = float(o.get("tf_min", 60.0)), float(o.get("tf_max", 200.0))      # parameters.jl:165-166
        self.v_max, self.w_max = float(o.get("v_max", 0.4)), float(o.get("w_max", np.deg2rad(1.0)))  # :135-136
        self.T_max, self.M_max = float(o.get("T_max", 20e-3)), float(o.get("M_max", 1e-4))          # :137-138
        z = 4.75
        self.obstacles = o.get("obstacles", [(1.0 / 0.3, c) for c in ([8.5, -0.15, 5.0], [11.2, 1.84, 5.0], [11.3, 3.8, 4.8])])
        self.rooms = o.get("rooms", [hyperrectangle([6.0, 0.0, z], 1.0, 1.0, 1.5, pitch=90.0),       # :102-109
                                     hyperrectangle([7.5, 0.0, z], 2.0, 2.0, 4.0, pitch=90.0),
                                     hyperrectangle([11.5, 0.0, z], 1.25, 1.25, 0.5, pitch=90.0),
                                     hyperrectangle([10.75, -1.0, z], 1.5, 1.5, 1.5, yaw=-90.0, pitch=90.0),
                                     hyperrectangle([10.75, 1.0, z], 1.5, 1.5, 1.5, yaw=90.0, pitch=90.0),
                                     hyperrectangle([10.75, 2.5, z], 2.5, 2.5, 4.5, yaw=90.0, pitch=90.0)])
        assert len(self.obstacles) == self.n_obs and len(self.rooms) == self.n_iss
        self.room_c = np.array([r[0] for r in self.rooms], float); self.room_s = np.array([r[1] for r in self.rooms], float)

    def bind(self, pars):
        self.N = int(pars.N)

    @property
    def np(self):
        if self.N is None:
            raise RuntimeError("free-flyer: the parameter vector has 1 + 6 N entries; create the SCP problem first (or pass N=...)")
        return self.np_glob + self.np_node * self.N

    def par(self):
        o = self.opts
        head = [o.get("m", 7.2)] + list(o.get("J", (0.1083, 0.1083, 0.1083))) + [                     # parameters.jl:139-141
            self.v_max, self.w_max, self.T_max, self.M_max, self.tf_min, self.tf_max, o.get("gamma", 0.0), o.get("hom", 50.0),
            o.get("eps_sdf", 1e-4)]                                                                    # :168-170
        obs = [v for h, c in self.obstacles for v in [h] + list(c)]
        rooms = [v for c, s_ in zip(self.room_c, self.room_s) for v in list(c) + list(s_)]
        return np.array(head + obs + rooms, dtype=float)

    def nominal_pp(self):
        # per-problem data [r0 v0 q0 w0 rf vf qf wf]  (parameters.jl:160-167)
        q0 = quat_from_axis_angle(np.deg2rad(-40.0), [0.0, 1.0, 1.0])
        qf = quat_from_axis_angle(0.0, [0.0, 0.0, 1.0])
        return np.concatenate([[6.5, -0.2, 5.0], [0.035, 0.035, 0.0], q0, np.zeros(3), [11.3, 6.0, 4.5], np.zeros(3), qf, np.zeros(3)])

    def scale_advice(self):
        # set_scale!, definition.jl:52-66: positions, t_f and the slacks advised; v, w, T, M from their norm balls; q free -> [0, 1]
        pp = self.nominal_pp()
        r0, rf = pp[0:3], pp[13:16]
        xb = np.vstack([np.stack([np.minimum(r0, rf), np.maximum(r0, rf)], axis=1), np.tile([[-self.v_max, self.v_max]], (3, 1)),
                        np.tile([[0.0, 1.0]], (4, 1)), np.tile([[-self.w_max, self.w_max]], (3, 1))])
        ub = np.vstack([np.tile([[-self.T_max, self.T_max]], (3, 1)), np.tile([[-self.M_max, self.M_max]], (3, 1))])
        pb = np.vstack([[[self.tf_min, self.tf_max]], np.tile([[-100.0, 1.0]], (self.np - 1, 1))])
        return xb, ub, pb

    def guess(self, N, pp):
        """definition.jl:84-186: axis-by-axis path at constant speed, SLERP attitude, constant body rate, idle inputs, the
        slacks set to the rooms' signed distances along the guess (:166-172)."""
        r0, q0, rf, qf = pp[0:3], pp[6:10], pp[13:16], pp[19:23]
        T = 0.5 * (self.tf_min + self.tf_max)
        speed = np.abs(rf - r0).sum() / T
        times = linrange(0.0, T, N)
        cum = np.cumsum(np.abs(rf - r0) / speed)
        x = np.zeros((N, 13))
        for k in range(N):
            tk = times[k]
            x[k, 0:3] = rf
            for i in range(3):
                if tk <= cum[i]:
                    t0 = cum[i - 1] if i > 0 else 0.0
                    a = r0.copy(); a[:i] = rf[:i]
                    b = a.copy(); b[i] = rf[i]
                    tc = max(t0, min(cum[i], tk)); c = (cum[i] - tc) / (cum[i] - t0)
                    x[k, 0:3] = c * a + (1 - c) * b
                    d = b - a
                    x[k, 3:6] = speed * d / np.linalg.norm(d)
                    break
            x[k, 6:10] = slerp_interpolate(q0, qf, k / (N - 1))
        ang, ax = quat_log(quat_mul(qf, np.concatenate([-q0[:3], [q0[3]]])))
        x[:, 10:13] = ang / T * ax
        delta = 1.0 - np.abs((x[:, None, 0:3] - self.room_c[None]) / self.room_s[None]).max(axis=2)     # [N, 6]
        return x, np.zeros((N, 6)), np.concatenate([[T], delta.reshape(-1)])


REGISTRY = {m.name: m for m in (DoubleIntegratorModel, QuadrotorModel, RocketLandingModel, StarshipModel, FreeflyerModel)}
