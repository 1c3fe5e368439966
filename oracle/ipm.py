"""CPU ORACLE (test infrastructure, NOT product code): primal-dual interior-point
solver for the conic programs the reference hands to ECOS.

    min  1/2 x'Px + c'x   s.t.  A x = b,   G x + s = h,   s in K
    K = R+^l  x  Q^{q_1} x ... x Q^{q_N}          (ECOS standard form + quadratic cost)

The reference reaches `libecos` (third party, un-vendored, version unpinned:
Project.toml:11) through JuMP `optimize!`
(src/parser/program.jl:419-424).  ECOS cannot be linked here, so this module
restates its *published algorithm class*: a Mehrotra predictor-corrector
primal-dual path-following method with Nesterov-Todd scaling for the symmetric
cones (Domahidi, Chu, Boyd, "ECOS: An SOCP solver for embedded systems", ECC
2013; the Newton system / NT-scaling algebra follows Vandenberghe, "The CVXOPT
linear and quadratic cone program solvers", 2010).  The quadratic objective is
handled natively (CVXOPT `coneqp` form) instead of through MOI's
quadratic->SOC bridge; the optimum is the same.

Parity status: "parity unpinned" w.r.t. ECOS itself; pinned by solver-independent
certificates in tests/test_oracle_ipm.py (KKT residuals, duality gap,
agreement with scipy's HiGHS on LPs, closed-form SOCP solutions).
Default tolerances and iteration limit are ECOS's defaults (feastol = abstol =
reltol = 1e-8, maxit = 100).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

OPTIMAL, ALMOST_OPTIMAL, ITERATION_LIMIT, NUMERICAL_ERROR = "OPTIMAL", "ALMOST_OPTIMAL", "ITERATION_LIMIT", "NUMERICAL_ERROR"


class Cone:
    def __init__(self, l, q):
        self.l = int(l)
        self.q = [int(v) for v in q]
        self.m = self.l + sum(self.q)
        self.deg = self.l + len(self.q)
        self.offs = []
        o = self.l
        for d in self.q:
            self.offs.append(o)
            o += d

    def e(self):
        v = np.zeros(self.m)
        v[: self.l] = 1.0
        for o in self.offs:
            v[o] = 1.0
        return v

    def prod(self, u, v):
        """Jordan product u o v."""
        w = np.empty(self.m)
        w[: self.l] = u[: self.l] * v[: self.l]
        for o, d in zip(self.offs, self.q):
            w[o] = u[o:o + d] @ v[o:o + d]
            w[o + 1:o + d] = u[o] * v[o + 1:o + d] + v[o] * u[o + 1:o + d]
        return w

    def inv_prod(self, lam, d_):
        """solve lam o u = d."""
        u = np.empty(self.m)
        u[: self.l] = d_[: self.l] / lam[: self.l]
        for o, d in zip(self.offs, self.q):
            l0, l1 = lam[o], lam[o + 1:o + d]
            d0, d1 = d_[o], d_[o + 1:o + d]
            u0 = (l0 * d0 - l1 @ d1) / (l0 * l0 - l1 @ l1)
            u[o] = u0
            u[o + 1:o + d] = (d1 - u0 * l1) / l0
        return u

    def max_step(self, s, ds):
        """largest alpha >= 0 with s + alpha ds in K (inf if unbounded)."""
        a = np.inf
        neg = ds[: self.l] < 0
        if neg.any():
            a = min(a, np.min(-s[: self.l][neg] / ds[: self.l][neg]))
        for o, d in zip(self.offs, self.q):
            s0, s1 = s[o], s[o + 1:o + d]
            d0, d1 = ds[o], ds[o + 1:o + d]
            # (s0 + a d0)^2 - |s1 + a d1|^2 >= 0 and s0 + a d0 >= 0
            qa = d0 * d0 - d1 @ d1
            qb = 2 * (s0 * d0 - s1 @ d1)
            qc = s0 * s0 - s1 @ s1
            roots = []
            if abs(qa) <= 1e-14 * (d0 * d0 + d1 @ d1 + 1e-300):
                if qb < 0:
                    roots.append(-qc / qb)
            else:
                disc = qb * qb - 4 * qa * qc
                if disc >= 0:
                    sq = np.sqrt(disc)
                    # numerically stable pair of roots
                    qq = -0.5 * (qb + (sq if qb >= 0 else -sq))
                    roots.append(qq / qa)
                    if qq != 0:
                        roots.append(qc / qq)
            # f(a) = s0 + a d0 - |s1 + a d1| is concave with f(0) > 0: its only positive
            # root is the exit point; roots of the squared equation with s0 + a d0 < 0 are spurious
            for r in roots:
                if r > 0 and s0 + r * d0 >= -1e-12 * (abs(s0) + abs(r * d0)):
                    a = min(a, r)
        return a

    def nt_scaling(self, s, z):
        """Nesterov-Todd scaling: returns (apply W, apply W^-1, W^2 blocks, lambda)."""
        w_l = np.sqrt(s[: self.l] / z[: self.l])
        lam = np.empty(self.m)
        lam[: self.l] = np.sqrt(s[: self.l] * z[: self.l])
        socs = []
        for o, d in zip(self.offs, self.q):
            sk, zk = s[o:o + d], z[o:o + d]
            sres = np.sqrt(sk[0] ** 2 - sk[1:] @ sk[1:])
            zres = np.sqrt(zk[0] ** 2 - zk[1:] @ zk[1:])
            sb, zb = sk / sres, zk / zres
            gamma = np.sqrt((1 + sb @ zb) / 2)
            wb = np.empty(d)
            wb[0] = (sb[0] + zb[0]) / (2 * gamma)
            wb[1:] = (sb[1:] - zb[1:]) / (2 * gamma)
            eta = np.sqrt(sres / zres)
            Wk = np.empty((d, d))
            Wk[0, 0] = wb[0]
            Wk[0, 1:] = wb[1:]
            Wk[1:, 0] = wb[1:]
            Wk[1:, 1:] = np.eye(d - 1) + np.outer(wb[1:], wb[1:]) / (1 + wb[0])
            # closed-form inverse: W^-1 = 1/eta [w0 -w1'; -w1 I + w1 w1'/(1+w0)] (no linear solve with the
            # ill-conditioned W near the cone boundary)
            Wik = Wk.copy()
            Wik[0, 1:] = -wb[1:]; Wik[1:, 0] = -wb[1:]
            Wk *= eta
            Wik /= eta
            socs.append((Wk, Wik))
            lam[o:o + d] = Wk @ zk
        return w_l, socs, lam

    def apply_W(self, w_l, socs, v, inverse=False):
        out = np.empty(self.m)
        out[: self.l] = v[: self.l] / w_l if inverse else v[: self.l] * w_l
        for (o, d), (Wk, Wik) in zip(zip(self.offs, self.q), socs):
            out[o:o + d] = Wik @ v[o:o + d] if inverse else Wk @ v[o:o + d]
        return out

    def Winv_matrix(self, w_l, socs):
        """block-diagonal W^-1 (sparse)."""
        blocks = [sp.diags(1.0 / w_l)] if self.l > 0 else []
        for _, Wik in socs:
            blocks.append(sp.csc_matrix(Wik))
        return sp.block_diag(blocks, format="csc") if blocks else sp.csc_matrix((0, 0))

    def interior(self, v):
        if self.l and np.min(v[: self.l]) <= 0:
            return False
        for o, d in zip(self.offs, self.q):
            if not v[o] > np.linalg.norm(v[o + 1:o + d]):
                return False
        return True

    def shift_interior(self, v):
        """cvxopt-style: if v not in int K, add (1 + alpha) e."""
        mins = []
        if self.l:
            mins.append(np.min(v[: self.l]))
        for o, d in zip(self.offs, self.q):
            mins.append(v[o] - np.linalg.norm(v[o + 1:o + d]))
        mn = min(mins) if mins else 1.0
        if mn <= 0:
            v = v + (1.0 - mn) * self.e()
        return v


def solve(c, G, h, l, q, A=None, b=None, P=None, max_iter=100, feastol=1e-8, abstol=1e-8, reltol=1e-8,
          verbose=False, normalise_objective=False):
    """Returns dict(status, x, y, z, s, pcost, dcost, gap, pres, dres, iters).

    normalise_objective: solve with the objective scaled so that its largest coefficient is 1e4 (only when it exceeds that) --
    the arithmetic of the product's solver on GuSTO subproblems whose penalty weight has escalated (csrc/conic_ipm.hpp `osc`);
    the absolute-gap test stays in the units of the original objective, multipliers / costs / gap are returned unscaled."""
    if normalise_objective:
        Pn = None if P is None else sp.csc_matrix(P)
        mc = max(float(np.abs(c).max()) if c.size else 0.0, float(np.abs(Pn.data).max()) if Pn is not None and Pn.nnz else 0.0)
        if mc > 1e4:
            osc = 1e4 / mc
            r = solve(osc * c, G, h, l, q, A, b, None if Pn is None else osc * Pn, max_iter, feastol, osc * abstol, reltol, verbose)
            for key in ("y", "z", "pcost", "dcost", "gap"):
                r[key] = r[key] / osc
            return r
    n = c.size
    K = Cone(l, q)
    m = K.m
    G = sp.csc_matrix(G) if G is not None else sp.csc_matrix((0, n))
    if G.shape[0] > m:          # trailing rows: exponential cones, 3 rows each (solve_exp below)
        assert (G.shape[0] - m) % 3 == 0
        return solve_exp(c, G, h, l, q, (G.shape[0] - m) // 3, A, b, P, max_iter, feastol, abstol, reltol, verbose)
    assert G.shape == (m, n)
    if A is None:
        A = sp.csc_matrix((0, n)); b = np.zeros(0)
    A = sp.csc_matrix(A)
    pe = A.shape[0]
    P = sp.csc_matrix((n, n)) if P is None else sp.csc_matrix(P)
    reg = 1e-10

    def kkt_factor(Wi):
        """Factor the SCALED KKT system (CVXOPT form): with Gt = W^-1 G and dzt = W dz,
             [P A' Gt'; A 0 0; Gt 0 -I] [dx; dy; dzt] = [bx; by; W^-1 bz]
        -- the cone block is -I instead of -W^2, whose condition number is the square of W's and destroys the cone
        rows of the direction once a second-order cone pair approaches the boundary.  Static regularisation
        +-reg with iterative refinement against the unregularised matrix (what ECOS does)."""
        Gt = (Wi @ G).tocsc()
        Kmat = sp.bmat([[P + reg * sp.eye(n), A.T, Gt.T],
                        [A, -reg * sp.eye(pe), None],
                        [Gt, None, -(1.0 + reg) * sp.eye(m)]], format="csc")
        Ktrue = sp.bmat([[P, A.T, Gt.T], [A, sp.csc_matrix((pe, pe)), None], [Gt, None, -sp.eye(m)]], format="csc")
        lu = spla.splu(Kmat)

        def solve_(rhs):
            rhs = rhs.copy()
            rhs[n + pe:] = Wi @ rhs[n + pe:]
            sol = lu.solve(rhs)
            for _ in range(5):  # iterative refinement against the unregularised system
                res = rhs - Ktrue @ sol
                if np.linalg.norm(res) <= 1e-15 * (1 + np.linalg.norm(rhs)):
                    break
                sol = sol + lu.solve(res)
            sol[n + pe:] = Wi.T @ sol[n + pe:]   # dz = W^-1 dzt (W symmetric)
            return sol
        return solve_

    # ---- initial point (cvxopt coneqp) ----
    ks = kkt_factor(sp.eye(m, format="csc"))
    sol = ks(np.concatenate([-c, b, h]))
    x, y, z = sol[:n], sol[n:n + pe], sol[n + pe:]
    s = -z.copy()
    s = K.shift_interior(s)
    z = K.shift_interior(z)

    nrm_b, nrm_h, nrm_c = max(1.0, np.linalg.norm(b)), max(1.0, np.linalg.norm(h)), max(1.0, np.linalg.norm(c))
    status = ITERATION_LIMIT
    info = {}
    for it in range(max_iter + 1):
        Px = P @ x
        rx = Px + A.T @ y + G.T @ z + c
        ry = A @ x - b
        rz = G @ x + s - h
        gap = float(s @ z)
        pcost = 0.5 * float(x @ Px) + float(c @ x)
        dcost = pcost + float(y @ ry) + float(z @ rz) - gap
        pres = max(np.linalg.norm(ry) / nrm_b, np.linalg.norm(rz) / nrm_h)
        dres = np.linalg.norm(rx) / nrm_c
        if pcost < 0:
            relgap = gap / -pcost
        elif dcost > 0:
            relgap = gap / dcost
        else:
            relgap = np.inf
        info = dict(x=x, y=y, z=z, s=s, pcost=pcost, dcost=dcost, gap=gap, pres=pres, dres=dres, relgap=relgap,
                    iters=it)
        if verbose:
            print("%3d pcost % .8e dcost % .8e gap %.2e pres %.2e dres %.2e" % (it, pcost, dcost, gap, pres, dres))
        if pres <= feastol and dres <= feastol and (gap <= abstol or relgap <= reltol):
            status = OPTIMAL
            break
        if it == max_iter:
            break
        try:
            w_l, socs, lam = K.nt_scaling(s, z)
            if not np.all(np.isfinite(lam)):
                raise FloatingPointError
            ks = kkt_factor(K.Winv_matrix(w_l, socs))
        except Exception:
            status = NUMERICAL_ERROR
            break
        mu = float(lam @ lam) / K.deg

        def newton(d_s):
            # rhs third block: -rz - W'(lam \ d_s)
            t = K.apply_W(w_l, socs, K.inv_prod(lam, d_s))  # W symmetric
            sol_ = ks(np.concatenate([-rx, -ry, -rz - t]))
            dx, dy, dz = sol_[:n], sol_[n:n + pe], sol_[n + pe:]
            ds = -rz - G @ dx
            return dx, dy, dz, ds

        lam2 = K.prod(lam, lam)
        dxa, dya, dza, dsa = newton(-lam2)
        a_aff = min(1.0, K.max_step(s, dsa), K.max_step(z, dza))
        sigma = (1 - a_aff) ** 3
        Wdz = K.apply_W(w_l, socs, dza)
        Wids = K.apply_W(w_l, socs, dsa, inverse=True)
        d_s = sigma * mu * K.e() - lam2 - K.prod(Wids, Wdz)
        dx, dy, dz, ds = newton(d_s)
        a = min(1.0, 0.99 * min(K.max_step(s, ds), K.max_step(z, dz)))
        for _ in range(60):  # safeguard: stay strictly inside the cone despite round-off in max_step
            if K.interior(s + a * ds) and K.interior(z + a * dz):
                break
            a *= 0.8
        x = x + a * dx; y = y + a * dy; z = z + a * dz; s = s + a * ds
    if status != OPTIMAL and info.get("pres", 1) <= 1e-6 and info.get("dres", 1) <= 1e-6 and \
            (info.get("gap", 1) <= 1e-6 or info.get("relgap", 1) <= 1e-6):
        status = ALMOST_OPTIMAL
    info["status"] = status
    return info


# ------------------------------------------------------------------------------------------------------------------------------
# Exponential cones (src/parser/cone.jl:36-47: EXP, z = (x, y, w) with y exp(x / y) <= w, y > 0 -- MOI.ExponentialCone), needed
# by GuSTO's softplus penalty (src/solvers/gusto.jl:996-1031).  ECOS handles them with the method of S. Akle Serrano,
# "Algorithms for unsymmetric cone optimization and an implementation for problems with the exponential cone" (2015), restated
# here: the symmetric cones keep their Nesterov-Todd scaling and Mehrotra correction; an exponential cone enters the Newton
# system through the Hessian of the DUAL cone's barrier at its multiplier, scaled by mu, and its complementarity condition is
# s + mu grad F*(z) = 0 (first order only); the step length is found by backtracking so that every exponential pair stays in
# its cones and none of them falls below a tenth of the average complementarity.
#   dual cone  K* = {(u, v, w): u < 0, -u exp(v / u) <= e w};   psi = v - u + u log(-u / w) >= 0
#   barrier    F*(u, v, w) = -log(psi) - log(-u) - log(w)
# ------------------------------------------------------------------------------------------------------------------------------
EXP_CENTRAL = np.array([-1.051383945322714, 0.556409619469370, 1.258967884768947])   # s = z = -grad F*(z) (ECOS's start, MOI order)


def _exp_psi0():
    """constant that makes F(s) + F*(z) + 3 log(mu) vanish on the central path (evaluated at s = z = EXP_CENTRAL, mu = 1)"""
    x, y, w = EXP_CENTRAL
    F = -np.log(y * np.log(w / y) - x) - np.log(y) - np.log(w)
    Fs = -np.log(y - x + x * np.log(-x / w)) - np.log(-x) - np.log(w)
    return -(F + Fs)


EXP_PSI0 = _exp_psi0()
EXP_MARGIN = 1.25     # swept on twelve GuSTO softplus programs (hom 5 / 50 / 500): 1.1 jams one, 1.25 solves all in 29-41 iterations, 1.5: 37-48


def exp_primal_interior(v):
    x, y, w = v
    return y > 0 and w > 0 and y * np.log(w / y) - x > 0


def exp_dual_interior(v):
    u, vv, w = v
    return u < 0 and w > 0 and vv - u + u * np.log(-u / w) > 0


def exp_dual_grad_hess(z):
    u, v, w = z
    L = np.log(-u / w)
    psi = v - u + u * L
    g = np.array([-L / psi - 1.0 / u, -1.0 / psi, (u / w) / psi - 1.0 / w])
    dpsi = np.array([L, 1.0, -u / w])
    H = np.outer(dpsi, dpsi) / psi ** 2
    H[0, 0] += -(1.0 / u) / psi + 1.0 / u ** 2
    H[0, 2] += (1.0 / w) / psi; H[2, 0] += (1.0 / w) / psi
    H[2, 2] += -(u / w ** 2) / psi + 1.0 / w ** 2
    return g, H


def solve_exp(c, G, h, l, q, ne, A=None, b=None, P=None, max_iter=100, feastol=1e-8, abstol=1e-8, reltol=1e-8, verbose=False):
    """the solver above with `ne` exponential cones after the second-order cones (rows in the order x, y, w)."""
    n = c.size
    K = Cone(l, q)
    ms, m = K.m, K.m + 3 * ne
    G = sp.csc_matrix(G)
    assert G.shape == (m, n)
    if A is None:
        A = sp.csc_matrix((0, n)); b = np.zeros(0)
    A = sp.csc_matrix(A)
    pe = A.shape[0]
    P = sp.csc_matrix((n, n)) if P is None else sp.csc_matrix(P)
    reg = 1e-10
    deg = K.deg + 3 * ne
    ex = [slice(ms + 3 * i, ms + 3 * i + 3) for i in range(ne)]

    def kkt_factor(WiT, Wi):
        """scaled system with Gt = W^-T G: WiT = W^-T (scales rows and right-hand sides), Wi = W^-1 (recovers dz)"""
        Gt = (WiT @ G).tocsc()
        Kmat = sp.bmat([[P + reg * sp.eye(n), A.T, Gt.T], [A, -reg * sp.eye(pe), None], [Gt, None, -(1.0 + reg) * sp.eye(m)]], format="csc")
        Ktrue = sp.bmat([[P, A.T, Gt.T], [A, sp.csc_matrix((pe, pe)), None], [Gt, None, -sp.eye(m)]], format="csc")
        lu = spla.splu(Kmat)

        def solve_(rhs):
            rhs = rhs.copy()
            rhs[n + pe:] = WiT @ rhs[n + pe:]
            sol = lu.solve(rhs)
            for _ in range(5):
                res = rhs - Ktrue @ sol
                if np.linalg.norm(res) <= 1e-15 * (1 + np.linalg.norm(rhs)):
                    break
                sol = sol + lu.solve(res)
            sol[n + pe:] = Wi @ sol[n + pe:]
            return sol
        return solve_
    I = sp.eye(m, format="csc")
    ks = kkt_factor(I, I)
    sol = ks(np.concatenate([-c, b, h]))
    x, y, z = sol[:n], sol[n:n + pe], sol[n + pe:].copy()
    s = -z.copy()
    s[:ms] = K.shift_interior(s[:ms]); z[:ms] = K.shift_interior(z[:ms])
    # exponential pairs start on the central ray, (s, z) = (t c, t c) with mu = t^2 equal to the average complementarity of the
    # symmetric part (the cone and its dual are cones, grad F* is homogeneous of degree -1: s = -mu grad F*(z) holds for every t);
    # a start at t = 1 next to symmetric products of 1e3 would leave the exponential pairs outside the neighbourhood
    # s_e'z_e / 3 >= 0.1 mu the line search maintains, and the first steps would be cut to nothing
    t0 = np.sqrt(max(1.0, float(s[:ms] @ z[:ms]) / max(K.deg, 1))) if ms else 1.0
    for e_ in ex:
        s[e_] = t0 * EXP_CENTRAL; z[e_] = t0 * EXP_CENTRAL
    nrm_b, nrm_h, nrm_c = max(1.0, np.linalg.norm(b)), max(1.0, np.linalg.norm(h)), max(1.0, np.linalg.norm(c))
    status = ITERATION_LIMIT
    info = {}

    def exp_ok(sv, zv, mu_t=None):
        """both members of every exponential pair inside their cones; with mu_t: no pair below a tenth of the average complementarity"""
        for e_ in ex:
            if not (exp_primal_interior(sv[e_]) and exp_dual_interior(zv[e_])):
                return False
            if mu_t is not None and sv[e_] @ zv[e_] / 3.0 < 0.1 * mu_t:
                return False
        return True
    for it in range(max_iter + 1):
        Px = P @ x
        rx = Px + A.T @ y + G.T @ z + c
        ry = A @ x - b
        rz = G @ x + s - h
        gap = float(s @ z)
        pcost = 0.5 * float(x @ Px) + float(c @ x)
        dcost = pcost + float(y @ ry) + float(z @ rz) - gap
        pres = max(np.linalg.norm(ry) / nrm_b, np.linalg.norm(rz) / nrm_h)
        dres = np.linalg.norm(rx) / nrm_c
        relgap = gap / -pcost if pcost < 0 else (gap / dcost if dcost > 0 else np.inf)
        info = dict(x=x, y=y, z=z, s=s, pcost=pcost, dcost=dcost, gap=gap, pres=pres, dres=dres, relgap=relgap, iters=it)
        if verbose:
            print("%3d pcost % .8e dcost % .8e gap %.2e pres %.2e dres %.2e" % (it, pcost, dcost, gap, pres, dres))
        if pres <= feastol and dres <= feastol and (gap <= abstol or relgap <= reltol):
            status = OPTIMAL
            break
        if it == max_iter:
            break
        mu = gap / deg
        try:
            w_l, socs, lam = K.nt_scaling(s[:ms], z[:ms])
            if not np.all(np.isfinite(lam)):
                raise FloatingPointError
            blocksT = [K.Winv_matrix(w_l, socs)] if ms else []
            blocks = [K.Winv_matrix(w_l, socs)] if ms else []
            gs = []
            for e_ in ex:
                g, H = exp_dual_grad_hess(z[e_])
                try:
                    Lc = np.linalg.cholesky(mu * H)        # mu H = L L',  W = L'
                except np.linalg.LinAlgError:              # psi -> 0: the rank-one part (1 / psi^2) swamps the rest in double precision
                    Lc = np.linalg.cholesky(mu * (H + 1e-14 * np.trace(H) * np.eye(3)))
                Li = np.linalg.inv(Lc)
                blocksT.append(sp.csc_matrix(Li)); blocks.append(sp.csc_matrix(Li.T))       # W^-T = L^-1,  W^-1 = L^-T
                gs.append(g)
            WiT = sp.block_diag(blocksT, format="csc"); Wi = sp.block_diag(blocks, format="csc")
            ks = kkt_factor(WiT, Wi)
        except Exception:
            status = NUMERICAL_ERROR
            break

        def newton(r3):
            sol_ = ks(np.concatenate([-rx, -ry, r3]))
            dx, dy, dz = sol_[:n], sol_[n:n + pe], sol_[n + pe:]
            return dx, dy, dz, -rz - G @ dx
        # affine direction: third-row right-hand side -rz + s on every cone
        dxa, dya, dza, dsa = newton(-rz + s)
        a_aff = 1.0
        if ms:
            a_aff = min(a_aff, K.max_step(s[:ms], dsa[:ms]), K.max_step(z[:ms], dza[:ms]))
        for _ in range(60):
            if exp_ok(s + a_aff * dsa, z + a_aff * dza):
                break
            a_aff *= 0.8
        sigma = min(1.0, max(1e-4, (1 - a_aff) ** 3))
        r3 = np.empty(m)
        if ms:
            lam2 = K.prod(lam, lam)
            Wdz = K.apply_W(w_l, socs, dza[:ms]); Wids = K.apply_W(w_l, socs, dsa[:ms], inverse=True)
            d_s = sigma * mu * K.e() - lam2 - K.prod(Wids, Wdz)
            r3[:ms] = -rz[:ms] - K.apply_W(w_l, socs, K.inv_prod(lam, d_s))
        for e_, g in zip(ex, gs):
            r3[e_] = -rz[e_] + s[e_] + sigma * mu * g
        dx, dy, dz, ds = newton(r3)
        a = 1.0
        if ms:
            a = min(1.0, 0.99 * min(K.max_step(s[:ms], ds[:ms]), K.max_step(z[:ms], dz[:ms])))
        a = min(a, 0.99) if ne else a
        for _ in range(80):
            sn, zn = s + a * ds, z + a * dz
            # fraction to the boundary of the exponential cones: the pair must still be inside 10 % further along the step (the
            # backtracking alone can stop a hair inside a cone; the next centring steps then collapse -- seen on GuSTO programs)
            if (not ms or (K.interior(sn[:ms]) and K.interior(zn[:ms]))) and exp_ok(sn, zn, float(sn @ zn) / deg) and \
                    exp_ok(s + EXP_MARGIN * a * ds, z + EXP_MARGIN * a * dz):
                break
            a *= 0.8
        x = x + a * dx; y = y + a * dy; z = z + a * dz; s = s + a * ds
    if status != OPTIMAL and info.get("pres", 1) <= 1e-6 and info.get("dres", 1) <= 1e-6 and \
            (info.get("gap", 1) <= 1e-6 or info.get("relgap", 1) <= 1e-6):
        status = ALMOST_OPTIMAL
    info["status"] = status
    return info
