"""CPU MIRROR of the product's stage-structured interior-point subproblem solver
(test infrastructure; NOT product code, NOT reference-derived).  Parity status of the oracle family:
"parity unpinned" (no golden vectors in the reference, see oracle/ptr_ref.py).

The HIP solver (scptoolbox.jl_amd/csrc/ipm2_kernel.hpp, ipm2_newton.hpp, ipm2_run.hpp) solves the *reduced* PTR
subproblem (see oracle/admm_ref.py for the reduction and its proof-by-test of
equivalence with the reference's literal conic program) with the same
Mehrotra/NT primal-dual method as oracle/ipm.py, but exploiting the time-staged
structure: all constraints are inequalities, the epigraph variables are
eliminated analytically from the Newton system, and what remains is a
symmetric positive definite block-tridiagonal + arrow system in
(z_1..z_N, p), z_k = (xh_k, uh_k), solved by a block Cholesky sweep.

This file holds (a) `StageProblem`: the canonical stage-form data layout the
GPU assembly kernel produces, built here from the oracle's model definitions,
and (b) `solve`: a numpy transliteration of the GPU algorithm.  Tests compare
the GPU arrays and iterates against these.

Row groups (all rows written as  a(zeta) <= 0  unless SOC):
  dyn_k   (type A, k<N-1): a = D_k z_k + E_k z_{k+1} + Fp_k p + c_k, |a_i| <= y_i, cost om_i y_i
  tr_k    (type B): a = z_k - zref_k split in the x-block and the u-block, |a_i| <= eta, cost t_k eta
  trp     (type B): a = p - pref, cost wtr eta_p
  loc_k   : a = Kl_k z_k + Kp_k p + cl_k with rows [0,ns): hinge (type C, cost hw_i v_i),
            [ns, ns+nl): plain a <= 0 (type D), then nsoc cones of 4 rows: a in Q^4 (type E)
  glin    : rows on p only, Lp p + lp <= 0 (type D)
  ic, tc  (type A): a = H z_0|z_{N-1} (x part) + Kp p + l, cost bw_i y_i
"""
import numpy as np

from .models import linrange
from .ptr_ref import _trapz_weights


class StageProblem:
    pass


def build_stage_problem(mdl, pars, scale, ref, pp):
    N, nx, nu, np_ = pars.N, mdl.nx, mdl.nu, mdl.np
    nz = nx + nu
    t = linrange(0.0, 1.0, N)
    w = _trapz_weights(t)
    Sx, cx, Su, cu, Sp, cp = scale.Sx, scale.cx, scale.Su, scale.cu, scale.Sp, scale.cp
    Sz = np.concatenate([Sx, Su]); cz = np.concatenate([cx, cu])
    P = StageProblem()
    P.N, P.nx, P.nu, P.np, P.nz = N, nx, nu, np_, nz
    # ---- cost ----
    ct = mdl.cost_terms()
    P.Qd = np.zeros((N, nz)); P.q = np.zeros((N, nz)); P.Qp = np.zeros(np_); P.qp = np.zeros(np_)
    const = 0.0
    for k in range(N):
        P.Qd[k, nx:] = 2 * w[k] * ct["Qu"] * Su * Su
        P.q[k, nx:] = w[k] * (2 * ct["Qu"] * cu * Su + ct["lu"] * Su)
        P.q[k, :nx] = w[k] * ct["lx"] * Sx
        const += w[k] * (ct["Qu"] @ (cu * cu) + ct["lu"] @ cu + ct["lx"] @ cx)
    P.q[N - 1, :nx] += ct["tx"] * Sx
    const += ct["tx"] @ cx
    if np_:
        P.qp = ct["tp"] * Sp + 2 * ct["Qp"] * cp * Sp
        P.Qp = 2 * ct["Qp"] * Sp * Sp
        const += ct["tp"] @ cp + ct["Qp"] @ (cp * cp)
    P.cost_const = const
    # ---- dynamics rows, scaled by iSx ----
    iSx = 1.0 / Sx
    P.D = np.zeros((N - 1, nx, nz)); P.E = np.zeros((N - 1, nx, nz)); P.Fp = np.zeros((N - 1, nx, np_))
    P.cd = np.zeros((N - 1, nx)); P.om = np.zeros((N - 1, nx))
    for k in range(N - 1):
        P.D[k, :, :nx] = -(iSx[:, None] * ref.A[k] * Sx[None, :])
        P.D[k, :, nx:] = -(iSx[:, None] * ref.Bm[k] * Su[None, :])
        P.E[k, :, :nx] = np.eye(nx)
        P.E[k, :, nx:] = -(iSx[:, None] * ref.Bp[k] * Su[None, :])
        if np_:
            P.Fp[k] = -(iSx[:, None] * ref.F[k] * Sp[None, :])
        cphys = cx - ref.A[k] @ cx - ref.Bm[k] @ cu - ref.Bp[k] @ cu - (ref.F[k] @ cp if np_ else 0.0) - ref.r[k]
        P.cd[k] = iSx * cphys
        P.om[k] = pars.wvc * w[k] * Sx
    # ---- trust regions ----
    P.zref = np.concatenate([(ref.xd - cx) / Sx, (ref.ud - cu) / Su], axis=1)
    P.ttr = pars.wtr * w
    P.pref = (ref.p - cp) / Sp if np_ else np.zeros(0)
    P.ttrp = pars.wtr
    # ---- stage-local rows: hinge (s), lin (X/U NONPOS with state/input part), soc ----
    ns = mdl.ns
    lin_rows, soc_rows, glin = [], [], []
    for k in range(N):
        lr, sr = [], []
        for is_x, rows in ((True, mdl.X(t[k], k + 1)), (False, mdl.U(t[k], k + 1))):
            for kind, M, Mp, m0 in rows:
                Mz = np.zeros((M.shape[0], nz))
                if is_x:
                    Mz[:, :nx] = M
                else:
                    Mz[:, nx:] = M
                if kind == "NONPOS":
                    if not np.any(Mz) and np.any(Mp):
                        if k == 0:
                            glin.append((Mp, m0))   # p-only rows: kept once (the reference repeats them per node)
                    else:
                        lr.append((Mz, Mp, m0))
                else:
                    assert not np.any(Mp)
                    sr.append((Mz, m0))
        lin_rows.append(lr); soc_rows.append(sr)
    nl = sum(r[0].shape[0] for r in lin_rows[0]); nsoc = len(soc_rows[0])
    P.ns, P.nl, P.nsoc = ns, nl, nsoc
    ml = ns + nl + 4 * nsoc
    P.ml = ml
    P.Kl = np.zeros((N, ml, nz)); P.Kp = np.zeros((N, ml, np_)); P.cl = np.zeros((N, ml)); P.hw = np.zeros((N, ns))
    for k in range(N):
        r0 = 0
        if ns:
            a = (t[k], k + 1, ref.xd[k], ref.ud[k], ref.p)
            s, C, Dm, G = mdl.s(*a), mdl.C(*a), mdl.D(*a), mdl.G(*a)
            rr = s - C @ ref.xd[k] - Dm @ ref.ud[k] - (G @ ref.p if np_ else 0.0)
            Mz = np.concatenate([C, Dm], axis=1)
            Kz = Mz * Sz[None, :]; Kpp = G * Sp[None, :] if np_ else np.zeros((ns, 0))
            cc = rr + Mz @ cz + (G @ cp if np_ else 0.0)
            nrm = np.sqrt((Kz * Kz).sum(1) + (Kpp * Kpp).sum(1)); e = 1.0 / np.maximum(nrm, 1e-12)
            P.Kl[k, :ns] = Kz * e[:, None]; P.Kp[k, :ns] = Kpp * e[:, None]; P.cl[k, :ns] = cc * e
            P.hw[k] = pars.wvc * w[k] / e
            r0 = ns
        for Mz, Mp, m0 in lin_rows[k]:
            mm = Mz.shape[0]
            Kz = Mz * Sz[None, :]; Kpp = Mp * Sp[None, :] if np_ else np.zeros((mm, 0))
            cc = m0 + Mz @ cz + (Mp @ cp if np_ else 0.0)
            nrm = np.sqrt((Kz * Kz).sum(1) + (Kpp * Kpp).sum(1)); e = np.where(nrm > 0.0, 1.0 / np.maximum(nrm, 1e-12), 1.0)
            P.Kl[k, r0:r0 + mm] = Kz * e[:, None]; P.Kp[k, r0:r0 + mm] = Kpp * e[:, None]; P.cl[k, r0:r0 + mm] = cc * e
            r0 += mm
        for Mz, m0 in soc_rows[k]:
            Kz = Mz * Sz[None, :]
            cc = m0 + Mz @ cz
            e = 1.0 / max(np.abs(Kz).max(), 1e-12)
            P.Kl[k, r0:r0 + 4] = Kz * e; P.cl[k, r0:r0 + 4] = cc * e
            r0 += 4
    # ---- global lin rows on p ----
    ng = sum(g[0].shape[0] for g in glin)
    P.ng = ng
    P.Lp = np.zeros((ng, np_)); P.lp = np.zeros(ng)
    r0 = 0
    for Mp, m0 in glin:
        mm = Mp.shape[0]
        Kpp = Mp * Sp[None, :]; cc = m0 + Mp @ cp
        e = 1.0 / np.maximum(np.sqrt((Kpp * Kpp).sum(1)), 1e-12)
        P.Lp[r0:r0 + mm] = Kpp * e[:, None]; P.lp[r0:r0 + mm] = cc * e
        r0 += mm
    # ---- boundary conditions ----
    def bc(g, H, Kk, xb):
        l = g - H @ xb - (Kk @ ref.p if np_ else 0.0)
        Hs = H * Sx[None, :]; Ks = Kk * Sp[None, :] if np_ else np.zeros((len(g), 0))
        cc = l + H @ cx + (Kk @ cp if np_ else 0.0)
        e = 1.0 / np.maximum(np.sqrt((Hs * Hs).sum(1) + (Ks * Ks).sum(1)), 1e-12)
        return Hs * e[:, None], Ks * e[:, None], cc * e, pars.wvc / e
    P.H0, P.K0, P.l0, P.bw0 = bc(mdl.gic(ref.xd[0], ref.p, pp), mdl.H0(ref.xd[0], ref.p, pp),
                                 mdl.K0(ref.xd[0], ref.p, pp), ref.xd[0])
    P.Hf, P.Kf, P.lf, P.bwf = bc(mdl.gtc(ref.xd[-1], ref.p, pp), mdl.Hf(ref.xd[-1], ref.p, pp),
                                 mdl.Kf(ref.xd[-1], ref.p, pp), ref.xd[-1])
    P.nic, P.ntc = len(P.l0), len(P.lf)
    P.scale = scale
    return P


# ---------------------------------------------------------------------------------------------
# structured IPM
# ---------------------------------------------------------------------------------------------

class State:
    """primal (z, p, aux), slacks s and duals lam per row group.  Pair groups carry
    index 0 for the '+a - aux <= 0' row and 1 for the '-a - aux <= 0' row."""


def _rows_eval(P, z, p):
    """a(zeta) for every row group."""
    N, nx = P.N, P.nx
    a = {}
    a["dyn"] = np.einsum("kij,kj->ki", P.D, z[:-1]) + np.einsum("kij,kj->ki", P.E, z[1:]) + \
        (P.Fp @ p if P.np else 0.0) + P.cd
    a["tr"] = z - P.zref
    a["trp"] = p - P.pref
    a["loc"] = np.einsum("kij,kj->ki", P.Kl, z) + (P.Kp @ p if P.np else 0.0) + P.cl
    a["glin"] = (P.Lp @ p if P.np else np.zeros(P.ng)) + P.lp
    a["ic"] = P.H0 @ z[0, :nx] + (P.K0 @ p if P.np else 0.0) + P.l0
    a["tc"] = P.Hf @ z[-1, :nx] + (P.Kf @ p if P.np else 0.0) + P.lf
    return a


def _rows_T(P, mu):
    """K' mu -> (gz[N,nz], gp[np]) for multipliers mu per group (same shapes as a)."""
    N, nx, nz = P.N, P.nx, P.nz
    gz = np.zeros((N, nz)); gp = np.zeros(P.np)
    gz[:-1] += np.einsum("kij,ki->kj", P.D, mu["dyn"])
    gz[1:] += np.einsum("kij,ki->kj", P.E, mu["dyn"])
    gz += mu["tr"]
    gz += np.einsum("kij,ki->kj", P.Kl, mu["loc"])
    gz[0, :nx] += P.H0.T @ mu["ic"]
    gz[-1, :nx] += P.Hf.T @ mu["tc"]
    if P.np:
        gp += np.einsum("kij,ki->j", P.Fp, mu["dyn"]) + mu["trp"] + np.einsum("kij,ki->j", P.Kp, mu["loc"])
        gp += P.Lp.T @ mu["glin"] + P.K0.T @ mu["ic"] + P.Kf.T @ mu["tc"]
    return gz, gp


def _soc_split(P, v):
    """view of the soc rows of a loc-shaped array: [N, nsoc, 4]."""
    o = P.ns + P.nl
    return v[:, o:].reshape(P.N, P.nsoc, 4)


def _nt(s, z):
    """NT scaling of one SOC pair -> (W [4,4], lam = W z, W^-1, W^-2), all in closed form:
    W = eta [w0 w1'; w1 I + w1 w1'/(1+w0)],  W^-1 = 1/eta [w0 -w1'; -w1 I + w1 w1'/(1+w0)],
    W^-2 = (2 v v' - J)/eta^2 with v = (w0, -w1), J = diag(1,-1,-1,-1)."""
    sres = np.sqrt(s[0] ** 2 - s[1:] @ s[1:]); zres = np.sqrt(z[0] ** 2 - z[1:] @ z[1:])
    sb, zb = s / sres, z / zres
    gamma = np.sqrt((1 + sb @ zb) / 2)
    wb = np.empty(4); wb[0] = (sb[0] + zb[0]) / (2 * gamma); wb[1:] = (sb[1:] - zb[1:]) / (2 * gamma)
    eta = np.sqrt(sres / zres)
    W = np.empty((4, 4)); W[0, 0] = wb[0]; W[0, 1:] = wb[1:]; W[1:, 0] = wb[1:]
    W[1:, 1:] = np.eye(3) + np.outer(wb[1:], wb[1:]) / (1 + wb[0])
    Wi = W.copy(); Wi[0, 1:] = -wb[1:]; Wi[1:, 0] = -wb[1:]
    W = W * eta
    Wi = Wi / eta
    v = np.concatenate([[wb[0]], -wb[1:]])
    Wi2 = (2 * np.outer(v, v) - np.diag([1.0, -1.0, -1.0, -1.0])) / (eta * eta)
    return W, W @ z, Wi, Wi2


def _typeB_matrix(w1, w2):
    """Schur complement of the shared epigraph variable of an L_inf block,
    diag(d) - h h'/W  (d = w1 + w2, h = w1 - w2, W = sum d), evaluated without
    cancellation:  diag(4 w1 w2 / d)  +  M,  M_ii = h_i^2 (sum_{j != i} d_j) / (d_i W),
    M_ij = -h_i h_j / W."""
    d = w1 + w2; h = w1 - w2; W = d.sum()
    M = -np.outer(h, h) / W
    n = d.size
    for i in range(n):
        rest = d[:i].sum() + d[i + 1:].sum()
        M[i, i] = 4 * w1[i] * w2[i] / d[i] + h[i] * h[i] * rest / (d[i] * W)
    return M


def _jprod(u, v):
    return np.concatenate([[u @ v], u[0] * v[1:] + v[0] * u[1:]])


def _jinv(lam, d):
    u0 = (lam[0] * d[0] - lam[1:] @ d[1:]) / (lam[0] ** 2 - lam[1:] @ lam[1:])
    return np.concatenate([[u0], (d[1:] - u0 * lam[1:]) / lam[0]])


def _soc_max_step(s, ds):
    s0, s1, d0, d1 = s[0], s[1:], ds[0], ds[1:]
    qa = d0 * d0 - d1 @ d1; qb = 2 * (s0 * d0 - s1 @ d1); qc = s0 * s0 - s1 @ s1
    roots = []
    if abs(qa) <= 1e-14 * (d0 * d0 + d1 @ d1 + 1e-300):
        if qb < 0:
            roots.append(-qc / qb)
    else:
        disc = qb * qb - 4 * qa * qc
        if disc >= 0:
            sq = np.sqrt(disc)
            qq = -0.5 * (qb + (sq if qb >= 0 else -sq))
            roots.append(qq / qa)
            if qq != 0:
                roots.append(qc / qq)
    a = np.inf
    for r in roots:
        if r > 0 and s0 + r * d0 >= -1e-12 * (abs(s0) + abs(r * d0)):
            a = min(a, r)
    return a


def chol_solve_arrow(T, U, C, Dp, rz, rp):
    """Solve [[T, C],[C', Dp]] [dz; dp] = [rz; rp]; T block tridiagonal with diagonal blocks
    T[k] (nz x nz) and sub-diagonal blocks U[k] = T_{k+1,k}.  Block Cholesky sweep (what the GPU does)."""
    N, nz = T.shape[0], T.shape[1]
    npp = Dp.shape[0]
    L = np.zeros_like(T); Lo = np.zeros_like(U)
    V = np.zeros((N, nz, npp)); yh = np.zeros((N, nz))
    for k in range(N):
        Tk = T[k].copy()
        rhs = rz[k].copy(); Ck = C[k].copy()
        if k > 0:
            Tk -= Lo[k - 1] @ Lo[k - 1].T
            rhs -= Lo[k - 1] @ yh[k - 1]
            Ck -= Lo[k - 1] @ V[k - 1]
        L[k] = np.linalg.cholesky(Tk)
        yh[k] = np.linalg.solve(L[k], rhs)
        V[k] = np.linalg.solve(L[k], Ck)
        if k < N - 1:
            Lo[k] = np.linalg.solve(L[k], U[k].T).T   # L_{k+1,k} = U_k L_kk^-T
    if npp:
        S = Dp - np.einsum("kip,kiq->pq", V, V)
        dp = np.linalg.solve(S, rp - np.einsum("kip,ki->p", V, yh))
    else:
        dp = np.zeros(0)
    dz = np.zeros((N, nz))
    for k in range(N - 1, -1, -1):
        rhs = yh[k] - (V[k] @ dp if npp else 0.0)
        if k < N - 1:
            rhs = rhs - Lo[k].T @ dz[k + 1]
        dz[k] = np.linalg.solve(L[k].T, rhs)
    return dz, dp


def qd_factor(H0, Dt, Et, kinv):
    """Block LDL' of the quasi-definite interleaved system (z_0, nu_0, z_1, nu_1, ..., z_{N-1}, nu_{N-1}):
         H0_k z_k + Dt_k' nu_k + Et_{k-1}' nu_{k-1}            = b_k
         Dt_k z_k + Et_k z_{k+1} - diag(kinv_k) nu_k           = t_k        (Et_{N-1} = 0)
    nu_0 = [ic rows; dyn_0], nu_k = dyn_k, nu_{N-1} = tc rows.  Forward recursion (every update ADDS
    positive semidefinite terms -- no cancellation; kappa only enters through kinv = 1/kappa):
         Sz_k  = H0_k + Et_{k-1}' Snu_{k-1}^-1 Et_{k-1}
         Snu_k = diag(kinv_k) + Dt_k Sz_k^-1 Dt_k'
    Dt, Et, kinv are lists (row counts vary).  Returns Cholesky factors (Lz[k], Lnu[k])."""
    N = H0.shape[0]
    Lz = np.zeros_like(H0); Lnu = []
    for k in range(N):
        Sz = H0[k].copy()
        if k > 0:
            X = np.linalg.solve(Lnu[k - 1], Et[k - 1])        # Lnu^-1 E
            Sz += X.T @ X
        Lz[k] = np.linalg.cholesky(Sz)
        Y = np.linalg.solve(Lz[k], Dt[k].T)               # Lz^-1 D'
        Lnu.append(np.linalg.cholesky(np.diag(kinv[k]) + Y.T @ Y))
    return Lz, Lnu


def qd_solve(Lz, Lnu, Dt, Et, b, t):
    """Solve the interleaved system; b[N, nz(,m)], t = list of [rows_k(,m)]."""
    N = Lz.shape[0]
    bp = np.zeros_like(b); tp = [None] * N
    cs = lambda L, r: np.linalg.solve(L.T, np.linalg.solve(L, r))
    for k in range(N):
        bp[k] = b[k]
        if k > 0:
            bp[k] = bp[k] + Et[k - 1].T @ cs(Lnu[k - 1], tp[k - 1])
        tp[k] = t[k] - Dt[k] @ cs(Lz[k], bp[k])
    z = np.zeros_like(b); nu = [None] * N
    nu[N - 1] = cs(Lnu[N - 1], -tp[N - 1])
    z[N - 1] = cs(Lz[N - 1], bp[N - 1] - Dt[N - 1].T @ nu[N - 1])
    for k in range(N - 2, -1, -1):
        nu[k] = cs(Lnu[k], Et[k] @ z[k + 1] - tp[k])
        z[k] = cs(Lz[k], bp[k] - Dt[k].T @ nu[k])
    return z, nu


def solve(P, max_iter=100, feastol=1e-8, abstol=1e-8, reltol=1e-8, verbose=False, trace=None, debug=False, nref=1, stall=3, hook=None, reg=5e-11, ref_gap=1e-2, init="two", resid_scale=False, sigma_min=0.0, ref_affine=True, ref_tol=0.0, ref_log=None, split_step=None, warm=None, warm_delta=1e-2,
          sigma_rule="ecos", nbhd=0.0, ncorr=0, step_frac=0.99, mu0=None, corr_delta=0.3, corr_accept=0.1, log=None, track_rz=False, probe=None):
    """Structured primal-dual IPM.  Returns dict(status, z, p, iters, pcost, ...)."""
    if split_step is None:   # separate primal / dual step lengths: optional (default off, like the device solver)
        split_step = False
    N, nx, nu, nz, npp = P.N, P.nx, P.nu, P.nz, P.np
    ns, nl, nsoc = P.ns, P.nl, P.nsoc
    o_soc = ns + nl

    # ---- groups: pair groups have arrays [..., 2]; aux costs ----
    groups_pair = {"dyn": (N - 1, nx), "ic": (P.nic,), "tc": (P.ntc,), "tr": (N, nz), "trp": (npp,), "hinge": (N, ns)}
    aux_cost = {"dyn": P.om, "ic": P.bw0, "tc": P.bwf, "hinge": P.hw}

    def zeros_like_rows():
        d = {g: np.zeros(sh + (2,)) for g, sh in groups_pair.items()}
        d["lin"] = np.zeros((N, nl)); d["glin"] = np.zeros(P.ng); d["soc"] = np.zeros((N, nsoc, 4))
        return d

    def G_apply(z, p, aux):
        """G xi per row (xi = main + aux)."""
        a = _rows_eval_lin(P, z, p)
        out = zeros_like_rows()
        for g in ("dyn", "ic", "tc"):
            out[g][..., 0] = a[g] - aux[g]; out[g][..., 1] = -a[g] - aux[g]
        out["tr"][:, :nx, 0] = a["tr"][:, :nx] - aux["etax"][:, None]; out["tr"][:, :nx, 1] = -a["tr"][:, :nx] - aux["etax"][:, None]
        out["tr"][:, nx:, 0] = a["tr"][:, nx:] - aux["etau"][:, None]; out["tr"][:, nx:, 1] = -a["tr"][:, nx:] - aux["etau"][:, None]
        out["trp"][..., 0] = a["trp"] - aux["etap"]; out["trp"][..., 1] = -a["trp"] - aux["etap"]
        out["hinge"][..., 0] = a["loc"][:, :ns] - aux["v"]; out["hinge"][..., 1] = -aux["v"]
        out["lin"] = a["loc"][:, ns:o_soc]
        out["glin"] = a["glin"]
        out["soc"] = -_soc_split(P, a["loc"])    # G = -M for cone rows
        return out

    def h_vec():
        """h per row (G xi + s = h)."""
        c = _rows_eval(P, np.zeros((N, nz)), np.zeros(npp))  # a(0) = constants
        out = zeros_like_rows()
        for g in ("dyn", "ic", "tc"):
            out[g][..., 0] = -c[g]; out[g][..., 1] = c[g]
        out["tr"][..., 0] = -c["tr"]; out["tr"][..., 1] = c["tr"]
        out["trp"][..., 0] = -c["trp"]; out["trp"][..., 1] = c["trp"]
        out["hinge"][..., 0] = -c["loc"][:, :ns]
        out["lin"] = -c["loc"][:, ns:o_soc]; out["glin"] = -c["glin"]
        out["soc"] = _soc_split(P, c["loc"])
        return out

    def GT_apply(lam):
        """G' lam -> (gz, gp, gaux)."""
        mu = {"dyn": lam["dyn"][..., 0] - lam["dyn"][..., 1], "ic": lam["ic"][..., 0] - lam["ic"][..., 1],
              "tc": lam["tc"][..., 0] - lam["tc"][..., 1], "tr": lam["tr"][..., 0] - lam["tr"][..., 1],
              "trp": lam["trp"][..., 0] - lam["trp"][..., 1], "glin": lam["glin"]}
        loc = np.zeros((N, P.ml)); loc[:, :ns] = lam["hinge"][..., 0]; loc[:, ns:o_soc] = lam["lin"]
        loc[:, o_soc:] = -lam["soc"].reshape(N, -1)
        mu["loc"] = loc
        gz, gp = _rows_T(P, mu)
        gaux = {"dyn": -(lam["dyn"][..., 0] + lam["dyn"][..., 1]), "ic": -(lam["ic"][..., 0] + lam["ic"][..., 1]),
                "tc": -(lam["tc"][..., 0] + lam["tc"][..., 1]),
                "etax": -(lam["tr"][:, :nx, :].sum(axis=(1, 2))), "etau": -(lam["tr"][:, nx:, :].sum(axis=(1, 2))),
                "etap": -(lam["trp"].sum()) if npp else 0.0, "v": -(lam["hinge"][..., 0] + lam["hinge"][..., 1])}
        return gz, gp, gaux

    caux = {"dyn": P.om, "ic": P.bw0, "tc": P.bwf, "etax": P.ttr, "etau": P.ttr, "etap": P.ttrp if npp else 0.0,
            "v": P.hw}
    LPG = ("dyn", "ic", "tc", "tr", "trp", "hinge", "lin", "glin")
    h = h_vec()
    nrm_h = max(1.0, np.sqrt(sum((h[g] ** 2).sum() for g in h)))
    nrm_c = max(1.0, np.sqrt((P.q ** 2).sum() + (P.qp ** 2).sum() + sum(np.sum(np.asarray(v, float) ** 2) for v in
                                                                         (P.om, P.bw0, P.bwf, P.ttr, P.ttr, P.hw))
                             + (P.ttrp ** 2 if npp else 0.0)))
    deg = sum(h[g].size for g in LPG) + N * nsoc

    def newton(w, Wsoc_i, rtil, rx):
        """Solve (P + G'W^-2 G) dxi = -rx - G'W^-2 rtil with aux elimination; returns dz, dp, daux.
        w: LP row weights (lam/s) per group; Wsoc[N,nsoc,4,4]: NT W; rtil: per-row r~z; rx = (rxz, rxp, rxaux)."""
        rxz, rxp, rxaux = rx
        rho = {g: w[g] * rtil[g] for g in LPG}
        T = np.zeros((N, nz, nz)); U = np.zeros((max(N - 1, 0), nz, nz)); C = np.zeros((N, nz, npp)); Dp = np.diag(P.Qp.copy()) if npp else np.zeros((0, 0))
        bz = -rxz.copy(); bp = -rxp.copy()
        T[:, np.arange(nz), np.arange(nz)] += P.Qd
        tau = {}
        # type A groups
        for g in ("dyn", "ic", "tc"):
            w1, w2 = w[g][..., 0], w[g][..., 1]; r1, r2 = rho[g][..., 0], rho[g][..., 1]
            Wt = w1 + w2
            rth = -rxaux[g] + r1 + r2
            kap = 4 * w1 * w2 / Wt
            tau[g] = -(r1 - r2) + (w1 - w2) * rth / Wt
            tau[g + "_kap"] = kap; tau[g + "_rth"] = rth; tau[g + "_Wt"] = Wt
        # dyn rows are kept in AUGMENTED form (nu_k = kappa a_k - tau): forming E'kappa E etc. in the
        # normal equations cancels catastrophically (penalty-method ill-conditioning) -- see qd_factor
        # type B: tr (x-block, u-block), trp
        trinfo = {}
        for name, sl, rxa in (("etax", slice(0, nx), rxaux["etax"]), ("etau", slice(nx, nz), rxaux["etau"])):
            w1, w2 = w["tr"][:, sl, 0], w["tr"][:, sl, 1]; r1, r2 = rho["tr"][:, sl, 0], rho["tr"][:, sl, 1]
            Wt = (w1 + w2).sum(1); hv = w1 - w2
            rth = -rxa + (r1 + r2).sum(1)
            idx = np.arange(sl.start, sl.stop)
            for k in range(N):
                T[k][np.ix_(idx, idx)] += _typeB_matrix(w1[k], w2[k])
                bz[k, sl] += -(r1[k] - r2[k]) + hv[k] * rth[k] / Wt[k]
            trinfo[name] = (hv, Wt, rth)
        if npp:
            w1, w2 = w["trp"][..., 0], w["trp"][..., 1]; r1, r2 = rho["trp"][..., 0], rho["trp"][..., 1]
            Wt = (w1 + w2).sum(); hv = w1 - w2; rth = -rxaux["etap"] + (r1 + r2).sum()
            Dp += _typeB_matrix(w1, w2)
            bp += -(r1 - r2) + hv * rth / Wt
            trinfo["etap"] = (hv, Wt, rth)
        # type C hinge, D lin, E soc -> via loc rows
        kap_loc = np.zeros((N, P.ml)); tau_loc = np.zeros((N, P.ml))
        if ns:
            # hinge rows go to the augmented nu-blocks as well (kap_loc stays 0 for them)
            w1, w2 = w["hinge"][..., 0], w["hinge"][..., 1]; r1, r2 = rho["hinge"][..., 0], rho["hinge"][..., 1]
            Wt = w1 + w2; rth = -rxaux["v"] + r1 + r2
            tau["hinge_kap"] = w1 * w2 / Wt
            tau["hinge"] = -r1 + w1 * rth / Wt
            tau["hinge_rth"] = rth; tau["hinge_Wt"] = Wt
        kap_loc[:, ns:o_soc] = w["lin"]; tau_loc[:, ns:o_soc] = -rho["lin"]
        for k in range(N):
            Kz, Kpk = P.Kl[k], P.Kp[k]
            # factored weights: rows scaled by sqrt(kappa) (LP rows) / W^-1 (cones) so that the
            # contribution is Y'Y -- forming W^-2 = (2vv' - J)/eta^2 explicitly cancels catastrophically
            Lm = np.diag(np.sqrt(kap_loc[k]))
            tl = tau_loc[k].copy()
            for j in range(nsoc):
                Wi = Wsoc_i[k, j]
                sl = slice(o_soc + 4 * j, o_soc + 4 * j + 4)
                Lm[sl, sl] = Wi
                # G = -M:  contribution  M' W^-2 M  and rhs  -G' W^-2 rtil = + M' W^-1 (W^-1 rtil)
                tl[sl] = Wi @ (Wi @ rtil["soc"][k, j])
            Yz = Lm @ Kz
            T[k] += Yz.T @ Yz
            bz[k] += Kz.T @ tl
            if npp:
                Yp = Lm @ Kpk
                C[k] += Yz.T @ Yp; Dp += Yp.T @ Yp; bp += Kpk.T @ tl
        if npp and P.ng:
            Dp += P.Lp.T @ (w["glin"][:, None] * P.Lp); bp += P.Lp.T @ (-rho["glin"])
        # nu blocks (augmented rows attached to stage k): [ic (k=0) | dyn_k (k<N-1) | hinge_k | tc (k=N-1)]
        Dt, Et, Ft, kinv, tt, segs = [], [], [], [], [], []
        zx = lambda M: np.concatenate([M, np.zeros((M.shape[0], nu))], axis=1)
        for k in range(N):
            Dl, El, Fl, kl, tl_, sg = [], [], [], [], [], {}
            r0 = 0
            def add(name, Dm, Em, Fm, kap_, tau_):
                nonlocal r0
                Dl.append(Dm); El.append(Em); Fl.append(Fm); kl.append(1.0 / kap_); tl_.append(tau_ / kap_)
                sg[name] = slice(r0, r0 + Dm.shape[0]); r0 += Dm.shape[0]
            if k == 0:
                add("ic", zx(P.H0), np.zeros((P.nic, nz)), P.K0, tau["ic_kap"], tau["ic"])
            if k < N - 1:
                add("dyn", P.D[k], P.E[k], P.Fp[k], tau["dyn_kap"][k], tau["dyn"][k])
            if ns:
                add("hinge", P.Kl[k, :ns], np.zeros((ns, nz)), P.Kp[k, :ns], tau["hinge_kap"][k], tau["hinge"][k])
            if k == N - 1:
                add("tc", zx(P.Hf), np.zeros((P.ntc, nz)), P.Kf, tau["tc_kap"], tau["tc"])
            Dt.append(np.vstack(Dl)); Et.append(np.vstack(El)); Ft.append(np.vstack(Fl))
            kinv.append(np.concatenate(kl)); tt.append(np.concatenate(tl_)); segs.append(sg)
        Lz, Lnu = qd_factor(T, Dt, Et, [ki + reg for ki in kinv])   # static dual regularisation (as ECOS), refined away
        if npp:
            # arrow: (z, nu) = y_b - Y_C p ;  (Dp - [C; Ft]'Y_C) p = bp - [C; Ft]'y_b
            yb_z, yb_nu = qd_solve(Lz, Lnu, Dt, Et, bz, tt)
            Yc_z, Yc_nu = qd_solve(Lz, Lnu, Dt, Et, C, Ft)
            Sp = Dp - np.einsum("kip,kiq->pq", C, Yc_z) - sum(Ft[k].T @ Yc_nu[k] for k in range(N))
            rp = bp - np.einsum("kip,ki->p", C, yb_z) - sum(Ft[k].T @ yb_nu[k] for k in range(N))
            dp = np.linalg.solve(Sp, rp)
            dz = yb_z - Yc_z @ dp
            nu_l = [yb_nu[k] - Yc_nu[k] @ dp for k in range(N)]
        else:
            dz, nu_l = qd_solve(Lz, Lnu, Dt, Et, bz, tt)
            dp = np.zeros(0)
        nus = {"dyn": np.zeros((N - 1, nx)), "hinge": np.zeros((N, ns)), "ic": nu_l[0][segs[0]["ic"]],
               "tc": nu_l[N - 1][segs[N - 1]["tc"]]}
        for k in range(N):
            if k < N - 1:
                nus["dyn"][k] = nu_l[k][segs[k]["dyn"]]
            if ns:
                nus["hinge"][k] = nu_l[k][segs[k]["hinge"]]
        # recover aux
        a = _rows_eval_lin(P, dz, dp)
        daux = {}
        for g in ("dyn", "ic", "tc"):
            w1, w2 = w[g][..., 0], w[g][..., 1]
            daux[g] = (tau[g + "_rth"] + (w1 - w2) * a[g]) / tau[g + "_Wt"]
        for name, sl in (("etax", slice(0, nx)), ("etau", slice(nx, nz))):
            hv, Wt, rth = trinfo[name]
            daux[name] = (rth + (hv * a["tr"][:, sl]).sum(1)) / Wt
        if npp:
            hv, Wt, rth = trinfo["etap"]
            daux["etap"] = (rth + hv @ a["trp"]) / Wt
        else:
            daux["etap"] = 0.0
        if ns:
            daux["v"] = (tau["hinge_rth"] + w["hinge"][..., 0] * a["loc"][:, :ns]) / tau["hinge_Wt"]
        else:
            daux["v"] = np.zeros((N, 0))
        return dz, dp, daux, nus

    # ---------------- initial point: (P + G'G) xi = -c + G'h, lam = G xi - h, s = -lam, shift ----------------
    w1 = {g: np.ones_like(h[g]) for g in LPG}
    Wsoc = np.tile(np.eye(4), (N, nsoc, 1, 1)); Wsoc_i = Wsoc.copy(); Wsoc_i2 = Wsoc.copy()
    rtil0 = {g: -h[g] for g in h}
    rx0 = (P.q.copy(), P.qp.copy(), {k_: np.array(v, float) * 1.0 for k_, v in caux.items()})
    z, p, aux, _ = newton(w1, Wsoc_i, rtil0, rx0)

    def check_newton(w_, Wsoc_, rtil_, rx_, dz_, dp_, daux_):
        """residual of the un-eliminated Newton system  P dxi + G'W^-2(G dxi + r~z) + rx = 0."""
        Gd_ = G_apply(dz_, dp_, daux_)
        lam_ = {g: w_[g] * (Gd_[g] + rtil_[g]) for g in LPG}
        lam_["soc"] = np.zeros((N, nsoc, 4))
        for k in range(N):
            for j in range(nsoc):
                lam_["soc"][k, j] = Wsoc_[k, j] @ (Wsoc_[k, j] @ (Gd_["soc"][k, j] + rtil_["soc"][k, j]))
        gz_, gp_, gaux_ = GT_apply(lam_)
        e = [np.abs(P.Qd * dz_ + gz_ + rx_[0]).max()]
        if npp:
            e.append(np.abs(P.Qp * dp_ + gp_ + rx_[1]).max())
        for k_ in gaux_:
            e.append(np.max(np.abs(np.asarray(gaux_[k_]) + np.asarray(rx_[2][k_]))) if np.size(gaux_[k_]) else 0.0)
        return max(e)
    if debug:
        print("init newton residual", check_newton(w1, Wsoc_i, rtil0, rx0, z, p, aux))

    def dlam_from(w_, Wi_, Gd_, rtil_, nus_, rxaux_):
        """dlam = W^-2 (G dxi + r~z); for the penalised pair rows (dyn, ic, tc, hinge) the multipliers are
        recovered from the augmented unknown nu and the aux dual-feasibility row instead, which avoids
        the amplification by w = lam/s ~ omega^2/mu:   type A: dl1 - dl2 = nu, dl1 + dl2 = rx_y;
        hinge: dl1 = nu, dl1 + dl2 = rx_v."""
        out = {g: w_[g] * (Gd_[g] + rtil_[g]) for g in LPG}
        for g in ("dyn", "ic", "tc"):
            out[g] = np.stack([0.5 * (rxaux_[g] + nus_[g]), 0.5 * (rxaux_[g] - nus_[g])], axis=-1)
        if ns:
            out["hinge"] = np.stack([nus_["hinge"], rxaux_["v"] - nus_["hinge"]], axis=-1)
        out["soc"] = np.zeros((N, nsoc, 4))
        for k in range(N):
            for j in range(nsoc):
                out["soc"][k, j] = Wi_[k, j] @ (Wi_[k, j] @ (Gd_["soc"][k, j] + rtil_["soc"][k, j]))
        return out

    def newton_refined(w_, W_, Wi_, rtil_, rx_, nref):
        """Newton step with iterative refinement in AUGMENTED form (dxi and dlam are both iterates):
             r1 = -rx - P dxi - G'dlam,   r2 = -r~z - G dxi + W^2 dlam   (O(1)-scaled residuals)
           correction: H e = r1 + G'W^-2 r2,  elam = W^-2 (G e - r2)."""
        dz_, dp_, daux_, nus_ = newton(w_, Wi_, rtil_, rx_)
        Gd_ = G_apply(dz_, dp_, daux_)
        dl_ = dlam_from(w_, Wi_, Gd_, rtil_, nus_, rx_[2])
        for _ in range(nref):
            gz_, gp_, gaux_ = GT_apply(dl_)
            r1 = (-(rx_[0] + P.Qd * dz_ + gz_), -(rx_[1] + P.Qp * dp_ + gp_),
                  {k_: -(np.asarray(rx_[2][k_]) + np.asarray(gaux_[k_])) for k_ in gaux_})
            r2 = {g: -rtil_[g] - Gd_[g] + dl_[g] / w_[g] for g in LPG}
            r2["soc"] = np.zeros((N, nsoc, 4))
            for k in range(N):
                for j in range(nsoc):
                    r2["soc"][k, j] = -rtil_["soc"][k, j] - Gd_["soc"][k, j] + W_[k, j] @ (W_[k, j] @ dl_["soc"][k, j])
            if debug:
                print("      refine: |r1z| %.2e |r1p| %.2e |r1aux| %.2e |r2| %.2e" % (
                    np.abs(r1[0]).max(), np.abs(r1[1]).max() if npp else 0,
                    max(np.max(np.abs(v)) if np.size(v) else 0 for v in r1[2].values()),
                    max(np.max(np.abs(v)) if np.size(v) else 0 for v in r2.values())))
            # adaptive: skip the correction solve when the residual of the computed direction is already tiny
            # relative to the right-hand side it was computed for
            r1n = max([np.abs(r1[0]).max(), np.abs(r1[1]).max() if npp else 0.0] + [np.max(np.abs(v)) if np.size(v) else 0.0 for v in r1[2].values()])
            rxn = max([np.abs(rx_[0]).max(), np.abs(rx_[1]).max() if npp else 0.0] + [np.max(np.abs(v)) if np.size(v) else 0.0 for v in rx_[2].values()])
            r2n = max(np.max(np.abs(v)) if np.size(v) else 0.0 for v in r2.values())
            rtn = max(np.max(np.abs(v)) if np.size(v) else 0.0 for v in rtil_.values())
            if ref_log is not None:
                ref_log.append((r1n, rxn, r2n, rtn, np.sqrt(sum((np.asarray(v) ** 2).sum() for v in [r1[0], r1[1]] + list(r1[2].values()))) / nrm_c))
            r1rel = np.sqrt(sum((np.asarray(v) ** 2).sum() for v in [r1[0], r1[1]] + list(r1[2].values()))) / nrm_c
            r2rel = np.sqrt(sum((np.asarray(v) ** 2).sum() for v in r2.values())) / nrm_h
            if r1rel <= ref_tol * feastol and r2rel <= ref_tol * feastol:
                break
            mr2 = {g: -r2[g] for g in r2}
            mr1aux = {k_: -r1[2][k_] for k_ in r1[2]}
            ez, ep, eaux, enus = newton(w_, Wi_, mr2, (-r1[0], -r1[1], mr1aux))
            Ge = G_apply(ez, ep, eaux)
            el = dlam_from(w_, Wi_, Ge, mr2, enus, mr1aux)
            dz_ = dz_ + ez; dp_ = dp_ + ep
            daux_ = {k_: daux_[k_] + eaux[k_] for k_ in daux_}
            dl_ = {g: dl_[g] + el[g] for g in dl_}
            Gd_ = {g: Gd_[g] + Ge[g] for g in Gd_}
        return dz_, dp_, daux_, dl_, Gd_

    Gx = G_apply(z, p, aux)
    lam = {g: Gx[g] - h[g] for g in h}
    s = {g: -lam[g] for g in h}
    if init == "two":
        # ECOS-style: primal point from min |G xi - h|^2 (+ xi'P xi), dual point from min |lam|^2 s.t. dual feasibility
        rx_zero = (np.zeros_like(P.q), np.zeros_like(P.qp), {k_: np.zeros_like(np.array(v, float)) for k_, v in caux.items()})
        z, p, aux, _ = newton(w1, Wsoc_i, rtil0, rx_zero)
        Gx = G_apply(z, p, aux)
        s = {g: h[g] - Gx[g] for g in h}
        rt_zero = {g: np.zeros_like(h[g]) for g in h}
        zd, pd, auxd, _ = newton(w1, Wsoc_i, rt_zero, rx0)
        lam = G_apply(zd, pd, auxd)

    def min_margin(v):
        m = min([v[g].min() for g in LPG if v[g].size] + [np.inf])
        if nsoc:
            m = min(m, (v["soc"][..., 0] - np.linalg.norm(v["soc"][..., 1:], axis=-1)).min())
        return m

    def shift(v):
        m = min_margin(v)
        if m <= 0:
            for g in LPG:
                v[g] = v[g] + (1.0 - m)
            if nsoc:
                v["soc"][..., 0] += (1.0 - m)
        return v
    s = shift(s); lam = shift(lam)
    if init == "mehrotra":
        # Mehrotra's LP starting-point balancing on top of the two-solve point
        def tot(v):
            return sum(v[g].sum() for g in LPG) + (v["soc"][..., 0].sum() if nsoc else 0.0)
        sl = sum((s[g] * lam[g]).sum() for g in h)
        ds_ = 0.5 * sl / tot(lam); dl_ = 0.5 * sl / tot(s)
        for g in LPG:
            s[g] = s[g] + ds_; lam[g] = lam[g] + dl_
        if nsoc:
            s["soc"][..., 0] += ds_; lam["soc"][..., 0] += dl_
    if init == "struct":
        m0 = mu0 if mu0 is not None else 1e-3
        z = P.zref.copy(); p = P.pref.copy()
        a0 = _rows_eval(P, z, p)
        aux = {}
        def typeA(a_, om_):
            om_ = np.maximum(om_, 1e-300)
            return (m0 + np.sqrt(m0 * m0 + om_ * om_ * a_ * a_)) / om_
        aux["dyn"] = typeA(a0["dyn"], P.om); aux["ic"] = typeA(a0["ic"], P.bw0); aux["tc"] = typeA(a0["tc"], P.bwf)
        aux["etax"] = 2 * nx * m0 / P.ttr; aux["etau"] = 2 * nu * m0 / P.ttr
        aux["etap"] = (2 * npp * m0 / P.ttrp) if npp else 0.0
        if ns:
            ah = a0["loc"][:, :ns]
            # hinge: s1 = v - a, s2 = v, lam1 + lam2 = hw, s1 lam1 = s2 lam2 = m0 -> m0/(v-a) + m0/v = hw
            # hw v^2 - (hw a + 2 m0) v + m0 a = 0
            bq = P.hw * ah + 2 * m0
            aux["v"] = (bq + np.sqrt(bq * bq - 4 * P.hw * m0 * ah)) / (2 * P.hw)
        else:
            aux["v"] = np.zeros((N, 0))
        Gx = G_apply(z, p, aux)
        s = {g: h[g] - Gx[g] for g in h}
        # non-epigraph rows: floor the slack
        fl = np.sqrt(m0)
        for g in ("lin", "glin"):
            s[g] = np.maximum(s[g], fl)
        if nsoc:
            marg = s["soc"][..., 0] - np.linalg.norm(s["soc"][..., 1:], axis=-1)
            s["soc"][..., 0] += np.maximum(fl - marg, 0.0)
        lam = {g: m0 / s[g] for g in LPG}
        if nsoc:
            # lam = m0 * s^-1 (Jordan inverse): s o lam = m0 e
            lam["soc"] = np.zeros_like(s["soc"])
            for k in range(N):
                for j in range(nsoc):
                    sv = s["soc"][k, j]; det = sv[0] ** 2 - sv[1:] @ sv[1:]
                    lam["soc"][k, j] = m0 * np.concatenate([[sv[0]], -sv[1:]]) / det
    if warm is not None:
        # warm start from the previous subproblem's iterate: keep (xi, s, lam) but push the complementarity pairs
        # back into the interior (s, lam) <- (s, lam) + delta * (cold-start scale)
        z = warm["z"].copy(); p = warm["p"].copy(); aux = {k_: np.array(v, float).copy() for k_, v in warm["aux"].items()}
        for g in h:
            if g == "soc":
                continue
            ds_ = warm_delta * np.maximum(1.0, np.abs(warm["s"][g]).max() if warm["s"][g].size else 1.0)
            s[g] = warm["s"][g] + warm_delta * max(1.0, float(np.mean(s[g]))) if s[g].size else s[g]
            lam[g] = warm["lam"][g] + warm_delta * max(1.0, float(np.mean(lam[g]))) if lam[g].size else lam[g]
        if nsoc:
            s["soc"] = warm["s"]["soc"].copy(); lam["soc"] = warm["lam"]["soc"].copy()
            s["soc"][..., 0] += warm_delta * max(1.0, float(np.mean(np.abs(s["soc"][..., 0]))))
            lam["soc"][..., 0] += warm_delta * max(1.0, float(np.mean(np.abs(lam["soc"][..., 0]))))

    status = "ITERATION_LIMIT"
    info = {}
    best = None
    w_last = None
    for it in range(max_iter + 1):
        gz, gp, gaux = GT_apply(lam)
        rxz = P.Qd * z + P.q + gz
        rxp = P.Qp * p + P.qp + gp
        rxaux = {k_: np.asarray(caux[k_], float) + gaux[k_] for k_ in caux}
        Gx = G_apply(z, p, aux)
        rz = {g: Gx[g] + s[g] - h[g] for g in h}
        if track_rz and it > 0:
            # the Newton step reduces the (linear) primal residual exactly by (1 - alpha): track it instead of
            # re-evaluating G xi + s - h, whose O(1) terms cancel and leave O(eps) noise that the next step would
            # try to remove from slacks that are themselves O(eps / weight) on the heavily penalised rows
            rz = {g: (1.0 - a_last) * rz_last[g] for g in h}
            if np.sqrt(sum((rz[g] ** 2).sum() for g in h)) / nrm_h < 1e-14:
                rz = {g: np.zeros_like(rz[g]) for g in h}
        rz_last = rz
        gap = sum((s[g] * lam[g]).sum() for g in h)
        cx_lin = (P.q * z).sum() + P.qp @ p + (P.om * aux["dyn"]).sum() + P.bw0 @ aux["ic"] + P.bwf @ aux["tc"] + \
            P.ttr @ aux["etax"] + P.ttr @ aux["etau"] + (P.ttrp * aux["etap"] if npp else 0.0) + (P.hw * aux["v"]).sum()
        pcost = 0.5 * ((P.Qd * z * z).sum() + (P.Qp * p * p).sum()) + cx_lin
        dcost = pcost + sum((lam[g] * rz[g]).sum() for g in h) - gap
        pres = np.sqrt(sum((rz[g] ** 2).sum() for g in h)) / nrm_h
        dres = np.sqrt((rxz ** 2).sum() + (rxp ** 2).sum() + sum(np.sum(np.asarray(v) ** 2) for v in rxaux.values())) / nrm_c
        relgap = gap / -pcost if pcost < 0 else (gap / dcost if dcost > 0 else np.inf)
        info = dict(z=z, p=p, aux=aux, s=s, lam=lam, pcost=pcost, dcost=dcost, gap=gap, pres=pres, dres=dres,
                    relgap=relgap, iters=it)
        if trace is not None:
            trace.append(dict(it=it, pcost=pcost, dcost=dcost, gap=gap, pres=pres, dres=dres))
        if debug:
            ij = np.unravel_index(np.argmax(np.abs(rxz)), rxz.shape)
            print("   argmax rxz at stage %d comp %d; w tr there: %s" % (ij[0], ij[1], w_last["tr"][ij[0], ij[1]] if w_last else None))
            print("   rx comps: z %.2e p %.2e " % (np.abs(rxz).max(), np.abs(rxp).max() if npp else 0) +
                  " ".join("%s %.2e" % (k_, np.max(np.abs(v)) if np.size(v) else 0) for k_, v in rxaux.items()))
        if verbose:
            print("%3d pcost % .8e dcost % .8e gap %.2e pres %.2e dres %.2e" % (it, pcost, dcost, gap, pres, dres))
        merit = max(pres / feastol, dres / feastol, min(gap / abstol, relgap / reltol))
        if best is None or merit < best[0]:
            best = (merit, dict(info))
        if merit <= 1.0:
            status = "OPTIMAL"
            break
        if it == max_iter:
            break
        # stall / divergence guard: normal equations lose accuracy once the gap is tiny; stop when the
        # merit has not improved for `stall` iterations (the best iterate is returned)
        if best[0] <= 1e3 and it - best[1]["iters"] >= stall:
            break
        # scalings
        w = {g: lam[g] / s[g] for g in LPG}
        w_last = w
        Wsoc = np.zeros((N, nsoc, 4, 4)); lsoc = np.zeros((N, nsoc, 4))
        Wsoc_i = np.zeros((N, nsoc, 4, 4)); Wsoc_i2 = np.zeros((N, nsoc, 4, 4))
        for k in range(N):
            for j in range(nsoc):
                Wsoc[k, j], lsoc[k, j], Wsoc_i[k, j], Wsoc_i2[k, j] = _nt(s["soc"][k, j], lam["soc"][k, j])
        if not (np.all(np.isfinite(Wsoc)) and all(np.all(np.isfinite(w[g])) for g in LPG)):
            status = "NUMERICAL_ERROR"
            break
        mu = gap / deg
        # iterative refinement only once the gap is small: the Newton system is well conditioned early on
        nref_it = nref if relgap < ref_gap else 0
        # affine direction: r~z = rz - s (LP), rz + W'(lam\(-lam o lam)) = rz - W lam_s = rz - s (SOC too)
        rtil = {g: rz[g] - s[g] for g in h}
        if debug and hook is not None:
            hook(dict(it=it, w=w, Wsoc=Wsoc, Wsoc_i=Wsoc_i, rtil=rtil, rx=(rxz, rxp, rxaux), newton=newton,
                      G_apply=G_apply, GT_apply=GT_apply, h=h, LPG=LPG, caux=caux))
        try:
            dz, dp, daux, dla, Gd = newton_refined(w, Wsoc, Wsoc_i, rtil, (rxz, rxp, rxaux), nref_it if ref_affine else 0)
        except np.linalg.LinAlgError:
            if debug:
                raise
            status = "NUMERICAL_ERROR"
            break
        dsa = {g: -rz[g] - Gd[g] for g in h}

        def max_step(v, dv):
            a = np.inf
            for g in LPG:
                neg = dv[g] < 0
                if neg.any():
                    a = min(a, np.min(-v[g][neg] / dv[g][neg]))
            for k in range(N):
                for j in range(nsoc):
                    a = min(a, _soc_max_step(v["soc"][k, j], dv["soc"][k, j]))
            return a
        a_aff = min(1.0, max_step(s, dsa), max_step(lam, dla))
        sigma = max(sigma_min, (1 - a_aff) ** 3)
        if sigma_rule == "mehrotra":
            # Mehrotra's original rule: (mu_aff / mu)^3 with mu_aff the complementarity after the affine step
            g_aff = sum(((s[g] + a_aff * dsa[g]) * (lam[g] + a_aff * dla[g])).sum() for g in h)
            sigma = max(sigma_min, min(1.0, (max(g_aff, 0.0) / gap) ** 3))
        rs_ = (1.0 - sigma) if resid_scale else 1.0   # CVXOPT/ECOS: residuals scaled by (1 - sigma) in the combined step
        # combined direction: d_s = sigma mu e - lam o lam - (W^-T ds_a) o (W dz_a)
        rtil2 = {g: rs_ * rz[g] - s[g] + (sigma * mu - dsa[g] * dla[g]) / lam[g] for g in LPG}
        rtil2["soc"] = np.zeros((N, nsoc, 4))
        for k in range(N):
            for j in range(nsoc):
                W = Wsoc[k, j]; l_ = lsoc[k, j]
                e = np.array([1.0, 0, 0, 0])
                d_s = sigma * mu * e - _jprod(l_, l_) - _jprod(Wsoc_i[k, j] @ dsa["soc"][k, j], W @ dla["soc"][k, j])
                rtil2["soc"][k, j] = rs_ * rz["soc"][k, j] + W @ _jinv(l_, d_s)
        try:
            dz, dp, daux, dl, Gd = newton_refined(w, Wsoc, Wsoc_i, rtil2, (rs_ * rxz, rs_ * rxp, {k_: rs_ * v for k_, v in rxaux.items()}), nref_it)
        except np.linalg.LinAlgError:
            if debug:
                raise
            status = "NUMERICAL_ERROR"
            break
        ds = {g: -rs_ * rz[g] - Gd[g] for g in h}
        # ---- Gondzio multiple centrality correctors (LP rows only) ----
        for ic_ in range(ncorr):
            a0_ = min(1.0, step_frac * min(max_step(s, ds), max_step(lam, dl)))
            at_ = min(1.0, a0_ + corr_delta)
            mu_t = sigma * mu
            bmin, bmax = 0.1, 10.0
            rt3 = {g: rtil2[g].copy() for g in rtil2}
            for g in LPG:
                v_ = (s[g] + at_ * ds[g]) * (lam[g] + at_ * dl[g])
                t_ = np.clip(v_, bmin * mu_t, bmax * mu_t)
                c_ = np.maximum(t_ - v_, -bmax * mu_t)
                rt3[g] = rtil2[g] + c_ / lam[g]
            dz3, dp3, daux3, dl3, Gd3 = newton_refined(w, Wsoc, Wsoc_i, rt3, (rs_ * rxz, rs_ * rxp, {k_: rs_ * v for k_, v in rxaux.items()}), nref_it)
            ds3 = {g: -rs_ * rz[g] - Gd3[g] for g in h}
            a3_ = min(1.0, step_frac * min(max_step(s, ds3), max_step(lam, dl3)))
            if log is not None:
                log.append(("corr", it, a0_, a3_))
            if a3_ >= a0_ + corr_accept * corr_delta:
                dz, dp, daux, dl, Gd, ds, rtil2 = dz3, dp3, daux3, dl3, Gd3, ds3, rt3
            else:
                break
        a = min(1.0, step_frac * min(max_step(s, ds), max_step(lam, dl)))
        if nbhd > 0.0:
            for _ in range(40):
                prods = np.concatenate([((s[g] + a * ds[g]) * (lam[g] + a * dl[g])).ravel() for g in LPG])
                tot_ = prods.sum() + (sum(((s["soc"] + a * ds["soc"]) * (lam["soc"] + a * dl["soc"])).sum() for _q in [0]) if nsoc else 0.0)
                if prods.min() >= nbhd * tot_ / deg:
                    break
                a *= 0.9
        for _ in range(60):
            sn = {g: s[g] + a * ds[g] for g in h}; ln = {g: lam[g] + a * dl[g] for g in h}
            if min_margin(sn) > 0 and min_margin(ln) > 0:
                break
            a *= 0.8
        if debug:
            ms_s = max_step(s, ds); ms_l = max_step(lam, dl)
            print("   a_aff %.3e sigma %.3e a %.3e maxstep s %.3e lam %.3e newton res %.2e" % (
                a_aff, sigma, a, ms_s, ms_l, 0.0))
            # which group limits
            for g in h:
                vv = {gg: (s[gg] if gg == g else np.ones_like(s[gg]) * 1e30) for gg in h}
            lim = []
            for g in LPG:
                neg = ds[g] < 0
                if neg.any():
                    lim.append((np.min(-s[g][neg] / ds[g][neg]), g, "s"))
                neg = dl[g] < 0
                if neg.any():
                    lim.append((np.min(-lam[g][neg] / dl[g][neg]), g, "lam"))
            lim.sort()
            print("   limiting:", lim[:3])
        if split_step:
            a_p = min(1.0, 0.99 * max_step(s, ds)); a_d = min(1.0, 0.99 * max_step(lam, dl))
            for _ in range(60):
                sn = {g: s[g] + a_p * ds[g] for g in h}
                if min_margin(sn) > 0:
                    break
                a_p *= 0.8
            for _ in range(60):
                ln = {g: lam[g] + a_d * dl[g] for g in h}
                if min_margin(ln) > 0:
                    break
                a_d *= 0.8
            z = z + a_p * dz; p = p + a_p * dp
            aux = {k_: aux[k_] + a_p * daux[k_] for k_ in aux}
            s = sn; lam = ln
            continue
        a_last = a
        if probe is not None:
            probe(locals())
        z = z + a * dz; p = p + a * dp
        aux = {k_: aux[k_] + a * daux[k_] for k_ in aux}
        s = sn; lam = ln
    if status != "OPTIMAL":
        info = best[1]
        # ECOS "reduced tolerances" (feastol_inacc 1e-4, abstol_inacc / reltol_inacc 5e-5) -> ALMOST_OPTIMAL
        if info["pres"] <= 1e-4 and info["dres"] <= 1e-4 and (info["gap"] <= 5e-5 or info["relgap"] <= 5e-5):
            status = "ALMOST_OPTIMAL"
        info["iters_total"] = it
    info["status"] = status
    return info


def _rows_eval_lin(P, z, p):
    """linear part of the row functions (no constants)."""
    a = _rows_eval(P, z, p)
    c = _rows_eval(P, np.zeros_like(z), np.zeros_like(p))
    return {g: a[g] - c[g] for g in a}


def G_apply_dir(P, dz, dp, daux, G_apply, N, nz, npp):
    return G_apply(dz, dp, daux)


def unpack(P, z, p):
    s = P.scale
    x = z[:, :P.nx] * s.Sx + s.cx
    u = z[:, P.nx:] * s.Su + s.cu
    pp_ = p * s.Sp + s.cp if P.np else np.zeros(0)
    return x, u, pp_
