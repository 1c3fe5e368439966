"""CPU ORACLE (test infrastructure, NOT product code): literal restatement of the
PTR algorithm of the reference -- subproblem construction in ECOS standard form
(App. B of SURVEY.md) + the outer loop.

Follows, line by line:
  variables        src/solvers/ptr.jl:247-266 (+ :586,637,691 dX_lq; :813-814 P, Pf)
  dynamics         src/solvers/scp.jl:657-674 -> discretization.jl:424-497
  X / U sets       src/solvers/scp.jl:685-734
  non-convex s     src/solvers/scp.jl:744-794
  BCs              src/solvers/scp.jl:808-895
  trust region     src/solvers/ptr.jl:565-743   (q_tr = Inf: LINF cones)
  cost             src/solvers/scp.jl:552-601, ptr.jl:773-895
  solve/extract    src/solvers/scp.jl:942-950, ptr.jl:399-432
  stopping         src/solvers/ptr.jl:908-932, scp.jl:909-931
  loop             src/solvers/ptr.jl:448-532
The conic solve itself is oracle/ipm.py (restating the ECOS algorithm class).
PARITY STATUS: "parity unpinned" -- the reference ships no golden vectors for this path and cannot be run here; this
restatement is pinned on mathematics (tests/test_oracle_*.py) and on the fixtures generated from it (tests/golden/).
NormInf / NormOne cones are lowered to R+ rows the way MathOptInterface's
bridges do (LINF(1+d) -> 2d rows; L1(1+d) -> d auxiliaries + 2d+1 rows).
"""
import math
import time

import numpy as np
import scipy.sparse as sp

from . import ipm
from . import oracle as orc
from .models import MODELS, linrange


class _Prog:
    """Tiny conic-program assembler over a flat variable vector."""

    def __init__(self):
        self.n = 0
        self.eq = []      # (rows list of (cols, vals), const)
        self.nonpos = []  # z <= 0
        self.soc = []     # list of blocks
        self.exp = []     # exponential cones (x, y, w): y exp(x / y) <= w  (src/parser/cone.jl:45)
        self.c = {}
        self.Pdiag = {}

    def var(self, n):
        o = self.n
        self.n += n
        return np.arange(o, o + n)

    @staticmethod
    def _expr(terms, const):
        """terms: list of (var_index_array, matrix[m, len(idx)]) -> (triplets, const)"""
        m = len(const)
        R, Cc, V = [], [], []
        for idx, M in terms:
            M = np.atleast_2d(np.asarray(M, float))
            assert M.shape == (m, len(idx)), (M.shape, m, len(idx))
            r, c = np.nonzero(M)
            R.append(r); Cc.append(np.asarray(idx)[c]); V.append(M[r, c])
        if R:
            return (np.concatenate(R), np.concatenate(Cc), np.concatenate(V)), np.asarray(const, float)
        return (np.zeros(0, int), np.zeros(0, int), np.zeros(0)), np.asarray(const, float)

    def add_zero(self, terms, const):
        self.eq.append(self._expr(terms, const))

    def add_nonpos(self, terms, const):
        self.nonpos.append(self._expr(terms, const))

    def add_soc(self, terms, const):
        self.soc.append(self._expr(terms, const))

    def add_exp(self, terms, const):
        assert len(const) == 3
        self.exp.append(self._expr(terms, const))

    def add_linf(self, t_idx, terms, const):
        """t >= ||expr||_inf  (MOI NormInfinity bridge: t - v_i >= 0, t + v_i >= 0)."""
        m = len(const)
        ones = np.ones((m, 1))
        self.add_nonpos(list(terms) + [(t_idx, -ones)], const)                       # v - t <= 0
        self.add_nonpos([(i, -np.asarray(M, float)) for i, M in terms] + [(t_idx, -ones)], -np.asarray(const, float))

    def add_l1(self, t_idx, terms, const):
        """t >= ||expr||_1  (MOI NormOne bridge: y_i >= |v_i|, t >= sum y)."""
        m = len(const)
        y = self.var(m)
        I = np.eye(m)
        self.add_nonpos(list(terms) + [(y, -I)], const)
        self.add_nonpos([(i, -np.asarray(M, float)) for i, M in terms] + [(y, -I)], -np.asarray(const, float))
        self.add_nonpos([(y, np.ones((1, m))), (t_idx, -np.ones((1, 1)))], np.zeros(1))

    def add_cost_lin(self, idx, w):
        for i, wi in zip(np.atleast_1d(idx), np.atleast_1d(w)):
            self.c[int(i)] = self.c.get(int(i), 0.0) + float(wi)

    def add_cost_quad_diag(self, idx, w):
        """+ sum_i w_i x_i^2  -> P_ii += 2 w_i."""
        for i, wi in zip(np.atleast_1d(idx), np.atleast_1d(w)):
            self.Pdiag[int(i)] = self.Pdiag.get(int(i), 0.0) + 2.0 * float(wi)

    def _stack(self, blocks, sign=1.0):
        if not blocks:
            return sp.csc_matrix((0, self.n)), np.zeros(0)
        R, Cc, V, consts = [], [], [], []
        off = 0
        for (r, c, v), const in blocks:
            R.append(r + off); Cc.append(c); V.append(v * sign); consts.append(const)
            off += len(const)
        M = sp.csc_matrix((np.concatenate(V), (np.concatenate(R), np.concatenate(Cc))), shape=(off, self.n))
        return M, np.concatenate(consts)

    def solve(self, **kw):
        A, a0 = self._stack(self.eq)                 # A x + a0 = 0
        Gn, g0 = self._stack(self.nonpos)            # Gn x + g0 <= 0  -> G = Gn, h = -g0
        Gs, s0 = self._stack(self.soc, sign=-1.0)    # z = M x + m in Q  -> G = -M, h = m
        Ge, e0 = self._stack(self.exp, sign=-1.0)    # exponential cones last (oracle/ipm.py::solve_exp)
        G = sp.vstack([Gn, Gs, Ge], format="csc")
        h = np.concatenate([-g0, s0, e0])
        q = [len(b[1]) for b in self.soc]
        c = np.zeros(self.n)
        for i, v in self.c.items():
            c[i] = v
        Pd = np.zeros(self.n)
        for i, v in self.Pdiag.items():
            Pd[i] = v
        self.sizes = dict(n=self.n, p=A.shape[0], l=Gn.shape[0], q=q, nexp=len(self.exp))
        return ipm.solve(c, G, h, Gn.shape[0], q, A=A, b=-a0, P=sp.diags(Pd), **kw)


class Scaling:
    """src/solvers/scp.jl:479-511 from bounding boxes."""

    def __init__(self, xb, ub, pb):
        tol = math.sqrt(np.finfo(float).eps)

        def one(bb):
            bb = np.asarray(bb, float).reshape(-1, 2)
            S = bb[:, 1] - bb[:, 0]
            S = np.where(S < tol, 1.0, S)
            return S, bb[:, 0].copy()
        self.Sx, self.cx = one(xb)
        self.Su, self.cu = one(ub)
        self.Sp, self.cp = one(pb)


class PTRParameters:
    """src/solvers/ptr.jl:57-71."""

    def __init__(self, N, Nsub, iter_max, wvc, wtr, eps_abs, eps_rel, feas_tol, q_tr=np.inf, q_exit=np.inf):
        self.N, self.Nsub, self.iter_max = N, Nsub, iter_max
        self.wvc, self.wtr, self.eps_abs, self.eps_rel, self.feas_tol = wvc, wtr, eps_abs, eps_rel, feas_tol
        self.q_tr, self.q_exit = q_tr, q_exit
        assert q_tr in (1, 2, 4, np.inf), "q_tr in {1, 2, 4, Inf} (ptr.jl:582-599; all reference tests use Inf)"


class Sol:
    pass


def _trapz(f, grid):  # helper.jl:560-568
    F = 0.0
    for k in range(len(grid) - 1):
        F += 0.5 * (grid[k + 1] - grid[k]) * (f[k + 1] + f[k])
    return F


def _trapz_weights(grid):
    w = np.zeros(len(grid))
    for k in range(len(grid) - 1):
        d = grid[k + 1] - grid[k]
        w[k] += 0.5 * d
        w[k + 1] += 0.5 * d
    return w


def lower_linf(kind, M, Mp, m0):
    """z = M x + Mp p + m0 = [t; y] in K_inf  <=>  +-y_j - t <= 0: MOI's NormInfinity bridge (ECOS has no LINF cone)."""
    if kind != "LINF":
        return kind, M, Mp, m0
    d = M.shape[0] - 1
    Dm = np.vstack([np.hstack([-np.ones((d, 1)), np.eye(d)]), np.hstack([-np.ones((d, 1)), -np.eye(d)])])
    return "NONPOS", Dm @ M, Dm @ Mp, Dm @ m0


def discretize(mdl, pars, scale, x, u, p):
    """SubproblemSolution(x,u,p,iter,pbm) -> discretize!  (ptr.jl:313-383)."""
    # np_dyn: leading parameters that enter the dynamics (free-flyer: the time dilation; its room-SDF slacks delta only
    # appear in constraints, so the columns of F beyond np_dyn are structurally zero, freeflyer/definition.jl:273-281)
    npd = getattr(mdl, "np_dyn", mdl.np)
    out = orc.discretize(mdl.name, mdl.par(), pars.N, pars.Nsub, x[None], u[None], p[None, :npd], 1.0 / scale.Sx,
                         pars.feas_tol)
    s = Sol()
    s.xd, s.ud, s.p = x, u, p
    s.A = np.swapaxes(out["A"][0], 1, 2); s.Bm = np.swapaxes(out["Bm"][0], 1, 2)
    s.Bp = np.swapaxes(out["Bp"][0], 1, 2); s.F = np.swapaxes(out["F"][0], 1, 2)
    if npd != mdl.np:
        s.F = np.concatenate([s.F, np.zeros(s.F.shape[:2] + (mdl.np - npd,))], axis=2)
    s.r = out["r"][0]; s.E = np.swapaxes(out["E"][0], 1, 2)
    s.defect = out["defect"][0]; s.feas = bool(out["feas"][0])
    s.J_aug = np.nan  # ptr.jl:350
    return s


def solve_subproblem(mdl, pars, scale, ref, pp, ipm_opts=None, algo="ptr", eta=None):
    """One PTR subproblem: formulate (ptr.jl:213-293, 467-480) + solve + extract.
    algo="scvx": the SCvx subproblem instead (scvx.jl:225-303, 578-698, 804-901; used by oracle/scvx_ref.py): hard
    trust region dx_lq[k] + du_lq[k] + dp_lq <= eta, cost L + lambda (trapz(P) + sum(Pf)), no eta variables."""
    scvx = algo == "scvx"
    N, nx, nu, np_ = pars.N, mdl.nx, mdl.nu, mdl.np
    t = linrange(0.0, 1.0, N)
    w = _trapz_weights(t)
    Sx, cx, Su, cu, Sp, cp = scale.Sx, scale.cx, scale.Su, scale.cu, scale.Sp, scale.cp
    P = _Prog()
    xh = [P.var(nx) for _ in range(N)]
    uh = [P.var(nu) for _ in range(N)]
    ph = P.var(np_)
    vd = [P.var(nx) for _ in range(N - 1)]
    if not scvx:
        etax, etau, etap = P.var(N), P.var(N), P.var(1)

    def phys(Mx=None, kx=None, Mu=None, ku=None, Mp=None, const=None):
        """affine expression in physical variables -> terms on scaled variables."""
        terms = []
        const = np.array(const, float).copy()
        if Mx is not None:
            Mx = np.atleast_2d(Mx); terms.append((xh[kx], Mx * Sx[None, :])); const += Mx @ cx
        if Mu is not None:
            Mu = np.atleast_2d(Mu); terms.append((uh[ku], Mu * Su[None, :])); const += Mu @ cu
        if Mp is not None and np_ > 0:
            Mp = np.atleast_2d(Mp); terms.append((ph, Mp * Sp[None, :])); const += Mp @ cp
        return terms, const

    # ---- dynamics (discretization.jl:458-467) ----
    for k in range(N - 1):
        t1, c1 = phys(Mx=np.eye(nx), kx=k + 1, const=np.zeros(nx))
        t2, c2 = phys(Mx=-ref.A[k], kx=k, Mu=-ref.Bm[k], ku=k, Mp=-ref.F[k] if np_ else None, const=-ref.r[k])
        t3, c3 = phys(Mu=-ref.Bp[k], ku=k + 1, const=np.zeros(nx))
        P.add_zero(t1 + t2 + t3 + [(vd[k], -ref.E[k])], c1 + c2 + c3)

    # ---- convex sets (scp.jl:685-734) ----
    def add_set(rows, k, is_x):
        for kind, M, Mp, m0 in rows:
            kind, M, Mp, m0 = lower_linf(kind, M, Mp, m0)
            terms, const = phys(Mx=M if is_x else None, kx=k, Mu=None if is_x else M, ku=k, Mp=Mp, const=m0)
            (P.add_nonpos if kind == "NONPOS" else P.add_soc)(terms, const)
    for k in range(N):
        add_set(mdl.X(t[k], k + 1), k, True)
    for k in range(N):
        add_set(mdl.U(t[k], k + 1), k, False)

    # ---- non-convex path constraints (scp.jl:744-794) ----
    ns = mdl.ns
    vs = [P.var(ns) for _ in range(N)] if ns > 0 else None
    for k in range(N):
        if ns == 0:
            break
        a = (t[k], k + 1, ref.xd[k], ref.ud[k], ref.p)
        s, C, D, G = mdl.s(*a), mdl.C(*a), mdl.D(*a), mdl.G(*a)
        r = s - C @ ref.xd[k] - D @ ref.ud[k] - (G @ ref.p if np_ else 0.0)
        terms, const = phys(Mx=C, kx=k, Mu=D, ku=k, Mp=G if np_ else None, const=r)
        P.add_nonpos(terms + [(vs[k], -np.eye(ns))], const)

    # ---- boundary conditions (scp.jl:808-895) ----
    gic = mdl.gic(ref.xd[0], ref.p, pp); H0 = mdl.H0(ref.xd[0], ref.p, pp); K0 = mdl.K0(ref.xd[0], ref.p, pp)
    l0 = gic - H0 @ ref.xd[0] - (K0 @ ref.p if np_ else 0.0)
    vic = P.var(len(gic))
    terms, const = phys(Mx=H0, kx=0, Mp=K0 if np_ else None, const=l0)
    P.add_zero(terms + [(vic, np.eye(len(gic)))], const)
    gtc = mdl.gtc(ref.xd[-1], ref.p, pp); Hf = mdl.Hf(ref.xd[-1], ref.p, pp); Kf = mdl.Kf(ref.xd[-1], ref.p, pp)
    lf = gtc - Hf @ ref.xd[-1] - (Kf @ ref.p if np_ else 0.0)
    vtc = P.var(len(gtc))
    terms, const = phys(Mx=Hf, kx=N - 1, Mp=Kf if np_ else None, const=lf)
    P.add_zero(terms + [(vtc, np.eye(len(gtc)))], const)

    # ---- trust region (ptr.jl:565-743), q = Inf ----
    xh_ref = (ref.xd - cx) / Sx
    uh_ref = (ref.ud - cu) / Su
    ph_ref = (ref.p - cp) / Sp if np_ else np.zeros(0)
    q_tr = getattr(pars, "q_tr", np.inf)

    def geom2(w_idx, terms_eta, const_eta):
        """(w, eta, 1) in GEOM (src/parser/cone.jl:36-47: geometric mean of (eta, 1) >= w) -- what JuMP's geometric-mean bridge
        hands a second-order-cone solver such as ECOS: w^2 <= eta * 1  <=>  (eta + 1, 2 w, eta - 1) in Q^3.
        `eta` = terms_eta (variables) + const_eta (a number)."""
        e3 = np.array([[1.0], [0.0], [1.0]])
        P.add_soc([(idx, e3 @ np.atleast_2d(M)) for idx, M in terms_eta] + [(w_idx, np.array([[0.0], [2.0], [0.0]]))],
                  np.array([const_eta + 1.0, 0.0, const_eta - 1.0]))

    def add_norm(t_idx, terms, const):
        """(t, expr) in the cone of the q_tr-norm: q2cone = {1: L1, 2: SOC, 4: SOC, Inf: LINF} (ptr.jl:582-583)."""
        if q_tr == np.inf:
            P.add_linf(t_idx, terms, const)
        elif q_tr == 1:
            P.add_l1(t_idx, terms, const)
        else:
            (idx, M), = terms
            n_ = len(const)
            P.add_soc([(t_idx, np.vstack([np.ones((1, 1)), np.zeros((n_, 1))])), (idx, np.vstack([np.zeros((1, n_)), M]))],
                      np.concatenate([[0.0], const]))
    dp_lq = P.var(1)
    if np_ > 0:
        add_norm(dp_lq, [(ph, np.eye(np_))], -ph_ref)   # ph = iSp*(p - cp) is the scaled variable itself
    else:
        P.add_nonpos([(dp_lq, -np.ones((1, 1)))], np.zeros(1))  # ||[]||_inf = 0 <= dp_lq
    one = np.ones((1, 1))

    def eta_link(lq, eta_var):
        """PTR: ||.||_q <= eta (ptr.jl:586-599); q = 4: (w, lq) in SOC and (w, eta, 1) in GEOM, i.e. lq^2 <= eta (:601-622)."""
        if q_tr == 4:
            wv = P.var(1)
            P.add_soc([(wv, np.array([[1.0], [0.0]])), (lq, np.array([[0.0], [1.0]]))], np.zeros(2))
            geom2(wv, [(eta_var, one)], 0.0)
        else:
            P.add_nonpos([(lq, one), (eta_var, -one)], np.zeros(1))
    if not scvx:
        eta_link(dp_lq, etap)
    dx_lq = P.var(N)
    for k in range(N):
        add_norm(dx_lq[k:k + 1], [(xh[k], np.eye(nx))], -xh_ref[k])
        if not scvx:
            eta_link(dx_lq[k:k + 1], etax[k:k + 1])
    du_lq = P.var(N)
    for k in range(N):
        add_norm(du_lq[k:k + 1], [(uh[k], np.eye(nu))], -uh_ref[k])
        if not scvx:
            eta_link(du_lq[k:k + 1], etau[k:k + 1])
    if scvx:   # trust region bound, scvx.jl:646-675: dx_lq[k] + du_lq[k] + dp_lq - eta <= 0;
        for k in range(N):      # q = 4 (:648-662): (w, dx_lq, du_lq, dp_lq) in SOC, (w, eta, 1) in GEOM
            if q_tr == 4:
                wv = P.var(1)
                e = lambda i: np.eye(4)[:, i:i + 1]
                P.add_soc([(wv, e(0)), (dx_lq[k:k + 1], e(1)), (du_lq[k:k + 1], e(2)), (dp_lq, e(3))], np.zeros(4))
                geom2(wv, [], float(eta))
            else:
                P.add_nonpos([(dx_lq[k:k + 1], one), (du_lq[k:k + 1], one), (dp_lq, one)], np.array([-float(eta)]))

    # ---- cost (scp.jl:552-601; ptr.jl:773-789, 799-895) ----
    ct = mdl.cost_terms()
    cost_const = 0.0
    for k in range(N):
        # Gamma = sum Qu_i u_i^2 + lu'u + lx'x with u = Su uh + cu
        P.add_cost_quad_diag(uh[k], w[k] * ct["Qu"] * Su * Su)
        P.add_cost_lin(uh[k], w[k] * (2 * ct["Qu"] * cu * Su + ct["lu"] * Su))
        P.add_cost_lin(xh[k], w[k] * ct["lx"] * Sx)
        cost_const += w[k] * (ct["Qu"] @ (cu * cu) + ct["lu"] @ cu + ct["lx"] @ cx)
    P.add_cost_lin(xh[N - 1], ct["tx"] * Sx)
    cost_const += ct["tx"] @ cx
    if np_ > 0:
        P.add_cost_lin(ph, ct["tp"] * Sp + 2 * ct["Qp"] * cp * Sp)
        P.add_cost_quad_diag(ph, ct["Qp"] * Sp * Sp)
        cost_const += ct["tp"] @ cp + ct["Qp"] @ (cp * cp)
    if not scvx:
        P.add_cost_lin(etax, pars.wtr * w); P.add_cost_lin(etau, pars.wtr * w); P.add_cost_lin(etap, pars.wtr)
    Pk = P.var(N); Pf = P.var(2)
    for k in range(N):
        if ns > 0:
            if k < N - 1:
                P.add_l1(Pk[k:k + 1], [(vd[k], np.vstack([ref.E[k], np.zeros((ns, nx))])),
                                       (vs[k], np.vstack([np.zeros((nx, ns)), np.eye(ns)]))], np.zeros(nx + ns))
            else:
                P.add_l1(Pk[k:k + 1], [(vs[k], np.eye(ns))], np.zeros(ns))
        else:
            if k < N - 1:
                P.add_l1(Pk[k:k + 1], [(vd[k], ref.E[k])], np.zeros(nx))
            else:
                P.add_zero([(Pk[k:k + 1], np.ones((1, 1)))], np.zeros(1))
    P.add_l1(Pf[0:1], [(vic, np.eye(len(gic)))], np.zeros(len(gic)))
    P.add_l1(Pf[1:2], [(vtc, np.eye(len(gtc)))], np.zeros(len(gtc)))
    wpen = pars.lam if scvx else pars.wvc     # scvx.jl:895-898: trapz(lambda P) + sum(lambda Pf)
    P.add_cost_lin(Pk, wpen * w); P.add_cost_lin(Pf, wpen * np.ones(2))

    t0 = time.perf_counter()
    res = P.solve(**(ipm_opts or {}))
    t_solve = time.perf_counter() - t0
    z = res["x"]
    x = np.stack([Sx * z[i] + cx for i in xh])      # value(blk) un-scales (block.jl:368-394)
    u = np.stack([Su * z[i] + cu for i in uh])
    p = Sp * z[ph] + cp if np_ else np.zeros(0)
    out = dict(x=x, u=u, p=p, vd=np.stack([z[i] for i in vd]), vs=np.stack([z[i] for i in vs]) if ns else None,
               vic=z[vic], vtc=z[vtc], status=res["status"], ipm=res, sizes=P.sizes, t_solve=t_solve,
               P=z[Pk], Pf=z[Pf], dx_lq=z[dx_lq], du_lq=z[du_lq], dp_lq=float(z[dp_lq][0]))
    if not scvx:
        out.update(etax=z[etax], etau=z[etau], etap=float(z[etap][0]))
    J = cost_const
    for k in range(N):
        J += w[k] * (ct["Qu"] @ (u[k] * u[k]) + ct["lu"] @ u[k] + ct["lx"] @ x[k]) - \
            w[k] * (ct["Qu"] @ (cu * cu) + ct["lu"] @ cu + ct["lx"] @ cx)
    J += ct["tx"] @ x[-1] - ct["tx"] @ cx
    if np_:
        J += ct["tp"] @ p + ct["Qp"] @ (p * p) - (ct["tp"] @ cp + ct["Qp"] @ (cp * cp))
    out["J"] = float(J)
    if scvx:   # L, L_pen, L_aug (scvx.jl:440-442)
        out["L"] = float(J)
        out["L_pen"] = float(pars.lam * (_trapz(z[Pk], t) + z[Pf].sum()))
        out["L_aug"] = out["L"] + out["L_pen"]
        return out
    out["J_tr"] = float(pars.wtr * (_trapz(z[etax], t) + _trapz(z[etau], t) + z[etap][0]))
    out["J_vc"] = float(pars.wvc * (_trapz(z[Pk], t) + z[Pf].sum()))
    out["J_aug"] = out["J"] + out["J_tr"] + out["J_vc"]
    return out


def solution_deviation(scale, pars, ref, sol):
    """src/solvers/scp.jl:909-931."""
    q = pars.q_exit
    xh, xr = (sol.xd - scale.cx) / scale.Sx, (ref.xd - scale.cx) / scale.Sx
    dp = np.linalg.norm((sol.p - ref.p) / scale.Sp, q) if sol.p.size else 0.0
    dx = max(np.linalg.norm(xh[k] - xr[k], q) for k in range(xh.shape[0]))
    return dp + dx


def ptr_solve(model, pars, pp=None, guess=None, ipm_opts=None, verbose=False):
    """`PTR.solve(pbm)` (src/solvers/ptr.jl:448-532) for one problem.
    Returns (status, history list of dicts)."""
    mdl = MODELS[model]() if isinstance(model, str) else model
    pp = mdl.nominal_pp() if pp is None else np.asarray(pp, float)
    scale = Scaling(*mdl.bbox())
    x, u, p = mdl.guess(pars.N, pp) if guess is None else guess
    ref = discretize(mdl, pars, scale, x, u, p)             # generate_initial_guess, ptr.jl:548-555
    hist = []
    k = 1
    status = "SCP_SOLVED"
    while True:
        sub = solve_subproblem(mdl, pars, scale, ref, pp, ipm_opts)
        t0 = time.perf_counter()
        sol = discretize(mdl, pars, scale, sub["x"], sub["u"], sub["p"])
        sub["t_discretize"] = time.perf_counter() - t0
        sol.J_aug = sub["J_aug"]
        if sub["status"] not in (ipm.OPTIMAL, ipm.ALMOST_OPTIMAL):  # unsafe_solution, scp.jl:965-980
            status = "SCP_FAILED (%s)" % sub["status"]
            hist.append(dict(sub=sub, sol=sol, ref=ref, stop=False)); break
        dev = solution_deviation(scale, pars, ref, sol)
        improv_rel = (ref.J_aug - sol.J_aug) / abs(ref.J_aug) if not np.isnan(ref.J_aug) else np.nan
        stop = k > 1 and (sol.feas and (abs(improv_rel) <= pars.eps_rel or dev <= pars.eps_abs))  # ptr.jl:924-927
        hist.append(dict(sub=sub, sol=sol, ref=ref, stop=stop, deviation=dev, improv_rel=improv_rel))
        if verbose:
            print("k=%2d %s J=% .6e Jtr=%.3e Jvc=%.3e Jaug=% .6e dev=%.3e feas=%s ipm_it=%d" % (
                k, sub["status"][:8], sub["J"], sub["J_tr"], sub["J_vc"], sub["J_aug"], dev, sol.feas,
                sub["ipm"]["iters"]))
        if stop:
            break
        ref = sol
        k += 1
        if k > pars.iter_max:
            break
    return status, hist
