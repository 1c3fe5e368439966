"""CPU ORACLE (test infrastructure, NOT product code -- moved here from the product package in round 5; the product evaluates the
guess with its kernels, csrc/starship_guess.hpp): literal restatement of

the reference's Starship initial guess (test/examples/starship_flip/definition.jl:97-445), host-side pre-processing:

  phase 1  bang-bang gimbal flip at minimum three-engine thrust, simulated with RK4 on 5000 points without aerodynamic
           torques (:120-171), cut where the vertical speed reaches the switch speed; resampled on the first half of the
           SCP grid; the state at the switch node becomes p[xs] and its altitude the cost normalisation `hs` (:181);
  phase 2  terminal descent as a convex program on a double integrator (lossless-convexification style: thrust vector
           inputs, SOC thrust and tilt bounds, FOH-discretised with the reference's own RK4 recipe, :183-231), solved for
           t2 = 10, 11, ... s until feasible (:404-421) -- here ALL candidate durations are one batch with a shared sparsity
           pattern (`solve_batch`), the first feasible one is taken;
  then theta, T, omega, m of phase 2 are reconstructed from the thrust vectors (:423-440).

Row equilibration (round 5).  The reference hands these programs to ECOS with DEFAULT options (`ConicProgram(solver = ECOS,
solver_options = Dict("verbose" => 0))`, definition.jl:291), and ECOS equilibrates its data by default.  The rows here are in
physical units -- thrust bounds of 2.2e6 N next to unit rows -- and WITHOUT equilibration the marginal candidates (the first
feasible durations) defeat an interior-point method that has no equilibration of its own: oracle/ipm.py ended t2 = 20 s at N = 100
in NUMERICAL_ERROR and took 21 s, the product's solver ended 21 ... 23 s in ITERATION_LIMIT and took 24 s.  Every row (every cone)
is therefore divided by its largest coefficient before the solve (`equilibrate`): the same feasible sets, and every candidate is then
DECIDED -- an infeasibility certificate below the first feasible duration, OPTIMAL in 8 ... 16 iterations from it on -- by both
solvers alike: t2 = 21 s at N = 31, 20 s at N = 100.

`solve_batch(c, G, h, l, q, A, b)` solves B conic programs with the pattern of (G, A): c[n], G scipy [m, n] pattern with
values Gx[B, nnz] (CSC order), h[B, m], Ax[B, nnzA], b[B, p]; returns (x[B, n], status[B]) with status <= 1 meaning
(ALMOST_)OPTIMAL.  The product passes the device solver (ConicProgramBatch); the golden-fixture generator passes the
oracle's IPM."""
import numpy as np
import scipy.sparse as sp

from .models import linrange      # (oracle/models.py)


class StarshipConstants:
    """test/examples/starship_flip/parameters.jl:99-212."""
    g0, m, rs, ls = 9.81, 120e3, 4.5, 50.0
    lcg, lcp = 0.4 * 50.0, 0.45 * 50.0
    J = 1.0 / 12.0 * 120e3 * (6 * 4.5 ** 2 + 50.0 ** 2)
    CD = 120e3 * 9.81 / 85.0 ** 2 * 1.2
    T_min1, T_max1 = 880e3, 2210e3
    T_min3, T_max3 = 3 * 880e3, 3 * 2210e3
    alpha_e = -1.0 / (330.0 * 9.81)
    delta_max = np.deg2rad(10.0)
    rate_delay = 0.05
    r0 = np.array([100.0, 600.0]); v0 = np.array([0.0, -85.0]); theta0 = np.deg2rad(90.0)
    theta_s = np.deg2rad(-10.0); vs = np.array([0.0, -10.0]); vf = np.array([0.0, -0.1])
    tau_s, theta_max2 = 0.5, np.deg2rad(15.0)


def _dynamics_no_aero_torque(x, u, K):
    """dynamics(...; no_aero_torques = true) with unit time dilation (definition.jl:498-550)."""
    v, th, om, dd = x[2:4], x[4], x[5], x[7]
    T, de = u[0], u[1]
    ei = np.array([np.cos(th), np.sin(th)]); ej = np.array([-np.sin(th), np.cos(th)])
    Tv = T * (-np.sin(de) * ei + np.cos(de) * ej)
    MT = -K.lcg * T * np.sin(de)
    D = -K.CD * np.linalg.norm(v) * v
    f = np.zeros(8)
    f[0:2] = v
    f[2:4] = (Tv + D) / K.m + np.array([0.0, -K.g0])
    f[4] = om
    f[5] = MT / K.J
    f[6] = K.alpha_e * T
    f[7] = (de - dd) / K.rate_delay
    return f


def _rk4_full(f, x0, t):
    X = np.zeros((len(t), x0.size)); X[0] = x0
    for j in range(1, len(t)):
        h = t[j] - t[j - 1]
        x = X[j - 1]
        k1 = f(t[j - 1], x); k2 = f(t[j - 1] + h / 2, x + h / 2 * k1); k3 = f(t[j - 1] + h / 2, x + h / 2 * k2)
        k4 = f(t[j - 1] + h, x + h * k3)
        X[j] = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    return X


def _descent_lti(dt_norm, tdil, K):
    """FOH discretisation of the double integrator over one normalised interval (definition.jl:183-231, same RK4 recipe on
    LinRange(0, dt, 100) with V = [Phi; int iPhi B s-; int iPhi B s+; int iPhi r])."""
    nx, nu = 4, 2
    A = np.zeros((4, 4)); A[0:2, 2:4] = np.eye(2)
    B = np.zeros((4, 2)); B[2:4] = np.eye(2) / K.m
    r = np.array([0.0, 0.0, 0.0, -K.g0])
    A, B, r = tdil * A, tdil * B, tdil * r

    def derivs(t, V):
        Phi = V[:16].reshape(4, 4, order="F")
        sm, sp_ = (dt_norm - t) / dt_norm, t / dt_norm
        iPhi = np.linalg.solve(Phi, np.eye(4))
        return np.concatenate([(A @ Phi).reshape(-1, order="F"), (iPhi @ B * sm).reshape(-1, order="F"),
                               (iPhi @ B * sp_).reshape(-1, order="F"), iPhi @ r])
    V = np.zeros(16 + 8 + 8 + 4); V[:16] = np.eye(4).reshape(-1, order="F")
    t = linrange(0.0, dt_norm, 100)
    for j in range(1, 100):
        h = t[j] - t[j - 1]
        k1 = derivs(t[j - 1], V); k2 = derivs(t[j - 1] + h / 2, V + h / 2 * k1); k3 = derivs(t[j - 1] + h / 2, V + h / 2 * k2)
        k4 = derivs(t[j - 1] + h, V + h * k3)
        V = V + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    Ak = V[:16].reshape(4, 4, order="F")
    return Ak, Ak @ V[16:24].reshape(4, 2, order="F"), Ak @ V[24:32].reshape(4, 2, order="F"), Ak @ V[32:36]


def equilibrate(G0, Gx, hs, l, q, A0, Ax, bs):
    """every inequality row / second-order cone / equality row of every program divided by its largest |coefficient|"""
    G0 = sp.csc_matrix(G0); A0 = sp.csc_matrix(A0)
    m = G0.shape[0]
    grp = np.arange(m)
    o = l
    for qq in q:
        grp[o:o + qq] = o
        o += qq
    Gx, hs, Ax, bs = Gx.copy(), hs.copy(), Ax.copy(), bs.copy()
    for t in range(Gx.shape[0]):
        mx = np.zeros(m); np.maximum.at(mx, G0.indices, np.abs(Gx[t]))
        gm = np.zeros(m); np.maximum.at(gm, grp, mx)
        e = np.where(gm[grp] > 0, 1.0 / np.where(gm[grp] > 0, gm[grp], 1.0), 1.0)
        Gx[t] *= e[G0.indices]; hs[t] *= e
        ma = np.zeros(A0.shape[0]); np.maximum.at(ma, A0.indices, np.abs(Ax[t]))
        ea = np.where(ma > 0, 1.0 / np.where(ma > 0, ma, 1.0), 1.0)
        Ax[t] *= ea[A0.indices]; bs[t] *= ea
    return Gx, hs, Ax, bs


def starship_initial_guess(N, solve_batch, K=StarshipConstants, t2_candidates=None):
    """Returns (x[N, 8], u[N, 3], p[10], hs)."""
    tau = linrange(0.0, 1.0, N)
    id1 = np.nonzero(tau <= K.tau_s)[0]
    id2 = np.arange(id1[-1], N)
    x_g = np.zeros((N, 8)); u_g = np.zeros((N, 3))
    # ---- phase 1: flip (:116-180) ----
    flip_ac = K.lcg / K.J * K.T_min3 * np.sin(K.delta_max)
    ts = np.sqrt((K.theta0 - K.theta_s) / flip_ac)

    def ctrl(t):
        de = K.delta_max if t <= ts else (-K.delta_max if t <= 2 * ts else 0.0)
        return np.array([K.T_min3, de, 0.0])
    x10 = np.zeros(8); x10[0:2] = K.r0; x10[2:4] = K.v0; x10[4] = K.theta0; x10[7] = K.delta_max
    tf = 2 * ts + 10.0
    t = linrange(0.0, tf, 5000)
    X1 = _rk4_full(lambda tt, x: _dynamics_no_aero_torque(x, ctrl(tt), K), x10, t)
    cross = np.nonzero(X1[:, 3] >= K.vs[1])[0]
    if cross.size == 0:
        raise ArithmeticError("no terminal velocity crossing, increase time of flight (t_theta_cst)")   # :163-167
    k0 = cross[0]
    t, X1 = t[:k0 + 1], X1[:k0 + 1]
    t1 = t[-1]

    def sample(tq):
        return np.array([np.interp(tq, t, X1[:, i]) for i in range(8)])
    for k in id1:
        tq = tau[k] / K.tau_s * t1
        x_g[k] = sample(tq); u_g[k] = ctrl(tq)
    xs = sample(tau[id1[-1]] / K.tau_s * t1)
    hs = float(xs[1])                                       # traj.hs = dot(xs[r], ey)  (:181)
    # ---- phase 2: descent programs, one per candidate duration (:183-421) ----
    tau2 = tau[id2] - tau[id2[0]]
    N2 = len(tau2)
    dtn = tau2[1] - tau2[0]
    Tmax_x = K.T_max1 * np.sin(K.theta_max2)
    Sx, cx = np.ones(4), np.zeros(4)
    Su, cu = np.ones(2), np.zeros(2)

    def upd(S, c, i, lo, hi):
        if lo > hi:
            lo, hi = hi, lo
        if hi - lo > np.sqrt(np.finfo(float).eps):
            S[i], c[i] = hi - lo, lo
    upd(Sx, cx, 0, 0, xs[0]); upd(Sx, cx, 1, 0, xs[1]); upd(Sx, cx, 2, 0, xs[2]); upd(Sx, cx, 3, 0, xs[3])
    upd(Su, cu, 0, -Tmax_x, Tmax_x); upd(Su, cu, 1, K.T_min1, K.T_max1)
    nxv = 4 * N2
    n = nxv + 2 * N2
    ix = lambda k: np.arange(4 * k, 4 * k + 4)
    iu = lambda k: nxv + np.arange(2 * k, 2 * k + 2)
    x0 = np.array([xs[0], xs[1], xs[2], xs[3]]); xf = np.array([0.0, 0.0, K.vf[0], K.vf[1]])
    cands = np.arange(10.0, 41.0, 1.0) if t2_candidates is None else np.asarray(t2_candidates, float)
    progs = []
    for t2 in cands:
        A_, Bm, Bp, r_ = _descent_lti(dtn, t2 / (1 - K.tau_s), K)
        rows, cols, vals, bvec = [], [], [], []
        nr = 0

        def eq(terms, const):
            nonlocal nr
            for idx, M in terms:
                rr, cc = np.nonzero(np.ones_like(M))            # dense blocks: the pattern must not depend on t2
                rows.extend(nr + rr); cols.extend(np.asarray(idx)[cc]); vals.extend(M[rr, cc])
            bvec.extend(-np.asarray(const)); nr += len(const)
        eq([(ix(0), np.diag(Sx))], cx - x0)                     # x_1 = x0
        eq([(ix(N2 - 1), np.diag(Sx))], cx - xf)                # x_N = xf
        for k in range(N2 - 1):                                 # x_{k+1} - (A x_k + B- u_k + B+ u_{k+1} + r) = 0
            eq([(ix(k + 1), np.diag(Sx)), (ix(k), -A_ * Sx[None, :]), (iu(k), -Bm * Su[None, :]), (iu(k + 1), -Bp * Su[None, :])],
               cx - A_ @ cx - Bm @ cu - Bp @ cu - r_)
        Am = sp.csc_matrix((vals, (rows, cols)), shape=(nr, n))
        # cone rows: NONPOS then SOC blocks
        g_rows, g_cols, g_vals, hv = [], [], [], []
        ng = 0
        for k in range(N2):                                     # T_min1 - u_y <= 0 ;  -r_y <= 0
            g_rows.append(ng); g_cols.append(iu(k)[1]); g_vals.append(-Su[1]); hv.append(-(K.T_min1 - cu[1])); ng += 1
            g_rows.append(ng); g_cols.append(ix(k)[1]); g_vals.append(-Sx[1]); hv.append(cx[1]); ng += 1
        l = ng
        q = []
        ct = 1.0 / np.cos(K.theta_max2)
        for k in range(N2):
            # (T_max1, u) in Q^3 : s = [T_max1; u] = h - G x
            hv.extend([K.T_max1, cu[0], cu[1]])
            g_rows.extend([ng + 1, ng + 2]); g_cols.extend([iu(k)[0], iu(k)[1]]); g_vals.extend([-Su[0], -Su[1]])
            ng += 3; q.append(3)
            # (u_y / cos(theta_max2), u) in Q^3
            hv.extend([cu[1] * ct, cu[0], cu[1]])
            g_rows.extend([ng, ng + 1, ng + 2]); g_cols.extend([iu(k)[1], iu(k)[0], iu(k)[1]]); g_vals.extend([-Su[1] * ct, -Su[0], -Su[1]])
            ng += 3; q.append(3)
        Gm = sp.csc_matrix((g_vals, (g_rows, g_cols)), shape=(ng, n))
        progs.append((Am, np.array(bvec), Gm, np.array(hv), l, q))
    A0, _, G0, _, l, q = progs[0]
    A0 = sp.csc_matrix(A0); A0.sort_indices(); G0 = sp.csc_matrix(G0); G0.sort_indices()

    def vals_of(M, P0):
        M = sp.csc_matrix(M); M.sort_indices()
        assert np.array_equal(M.indices, P0.indices) and np.array_equal(M.indptr, P0.indptr)
        return M.data
    Ax = np.stack([vals_of(p_[0], A0) for p_ in progs]); bs = np.stack([p_[1] for p_ in progs])
    Gx = np.stack([vals_of(p_[2], G0) for p_ in progs]); hs_ = np.stack([p_[3] for p_ in progs])
    Gx, hs_, Ax, bs = equilibrate(G0, Gx, hs_, l, q, A0, Ax, bs)
    z, status = solve_batch(np.zeros(n), G0, Gx, hs_, l, q, A0, Ax, bs)
    ok = np.nonzero(np.asarray(status) <= 1)[0]
    if ok.size == 0:
        raise ArithmeticError("could not find a terminal descent time of flight")      # :415-419
    t2 = float(cands[ok[0]]); zz = z[ok[0]]
    X2 = np.stack([Sx * zz[ix(k)] + cx for k in range(N2)]); T2 = np.stack([Su * zz[iu(k)] + cu for k in range(N2)])
    x_g[id2, 0:2] = X2[:, 0:2]; x_g[id2, 2:4] = X2[:, 2:4]
    tdil = t2 / (1 - K.tau_s)
    m20 = x_g[id2[0], 6]
    for k in range(N2):
        j = id2[k]
        x_g[j, 4] = -np.arctan2(T2[k, 0], T2[k, 1])
        u_g[j, 0] = np.linalg.norm(T2[k])
        if k > 0:
            x_g[j - 1, 5] = (x_g[j, 4] - x_g[j - 1, 4]) / ((tau2[k] - tau2[k - 1]) * tdil)
            tt = tau2[:k + 1] * tdil
            ff = K.alpha_e * u_g[id2[:k + 1], 0]
            x_g[j, 6] = m20 + np.sum(0.5 * np.diff(tt) * (ff[1:] + ff[:-1]))
    p = np.concatenate([[t1, t2], xs])
    return x_g, u_g, p, hs
