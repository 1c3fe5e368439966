"""CPU MIRROR of an ADMM subproblem solver prototype (test
infrastructure; NOT reference-derived and NOT product code; parity status "parity unpinned", see oracle/ptr_ref.py).

An earlier design (measured and rejected) solved the PTR
subproblem in a *reduced* form that is mathematically equivalent to the conic
program the reference builds (src/solvers/ptr.jl:213-293,565-895; restated
literally in oracle/ptr_ref.py): the epigraph variables eta, dX_lq, P, Pf and
the virtual controls vd, vs, vic, vtc are eliminated analytically,

    eta_x[k] = ||xh_k - xh_ref_k||_inf      (cost weight wtr*w_k > 0)
    P[k]     = ||E_k vd_k||_1 + ||vs_k||_1,  E_k vd_k = dynamics defect,
    vs_k     = max(linearised s_k, 0),       vic = -(H0 x_1 + K0 p + l0), ...

leaving   min_z 1/2 z'Qz + q'z + sum_j g_j(K_j z + c_j)   over z = (xh, uh, ph)
with prox-friendly g_j (weighted L1, L_inf norm, hinge, cone indicators).  This
file builds that reduced problem from the *oracle's* model definitions with
generic sparse algebra and runs the same OSQP-style ADMM iteration the GPU
kernels implement with block-tridiagonal structure.  tests/ use it (a) to show
the reduction is exact (same optimum as the literal conic form solved by
oracle/ipm.py) and (b) to check the HIP kernels step by step.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .models import linrange
from .ptr_ref import _trapz_weights


def proj_l1_ball(v, radius):
    """Euclidean projection onto {x : ||x||_1 <= radius} (sort-based)."""
    if radius <= 0:
        return np.zeros_like(v)
    a = np.abs(v)
    if a.sum() <= radius:
        return v.copy()
    srt = np.sort(a)[::-1]
    css = np.cumsum(srt)
    k = np.arange(1, a.size + 1)
    cond = srt - (css - radius) / k > 0
    rho = k[cond][-1]
    theta = (css[rho - 1] - radius) / rho
    return np.sign(v) * np.maximum(a - theta, 0.0)


def proj_soc(v):
    t, x = v[0], v[1:]
    nx = np.linalg.norm(x)
    if nx <= t:
        return v.copy()
    if nx <= -t:
        return np.zeros_like(v)
    a = 0.5 * (1 + t / nx)
    return np.concatenate([[a * nx], a * x])


class Reduced:
    pass


def build_reduced(mdl, pars, scale, ref, pp):
    """Reduced PTR subproblem about `ref` (see module docstring)."""
    N, nx, nu, np_ = pars.N, mdl.nx, mdl.nu, mdl.np
    nz = nx + nu
    n = N * nz + np_
    t = linrange(0.0, 1.0, N)
    w = _trapz_weights(t)
    Sx, cx, Su, cu, Sp, cp = scale.Sx, scale.cx, scale.Su, scale.cu, scale.Sp, scale.cp
    ix = lambda k: np.arange(k * nz, k * nz + nx)
    iu = lambda k: np.arange(k * nz + nx, (k + 1) * nz)
    ip = np.arange(N * nz, n)
    rows_r, rows_c, rows_v, consts, blocks = [], [], [], [], []
    m = [0]

    def add_block(kind, terms, const, **par):
        mm = len(const)
        for idx, M in terms:
            M = np.atleast_2d(np.asarray(M, float))
            r, c = np.nonzero(M)
            rows_r.append(r + m[0]); rows_c.append(np.asarray(idx)[c]); rows_v.append(M[r, c])
        consts.append(np.asarray(const, float))
        blocks.append(dict(kind=kind, start=m[0], len=mm, **par))
        m[0] += mm

    def phys(kx=None, Mx=None, ku=None, Mu=None, Mp=None, const=None):
        terms, const = [], np.array(const, float).copy()
        if Mx is not None:
            Mx = np.atleast_2d(Mx); terms.append((ix(kx), Mx * Sx[None, :])); const += Mx @ cx
        if Mu is not None:
            Mu = np.atleast_2d(Mu); terms.append((iu(ku), Mu * Su[None, :])); const += Mu @ cu
        if Mp is not None and np_ > 0:
            Mp = np.atleast_2d(Mp); terms.append((ip, Mp * Sp[None, :])); const += Mp @ cp
        return terms, const

    def row_scale(terms, const, e):
        return [(i, M * e[:, None]) for i, M in terms], const * e

    iSx = 1.0 / Sx
    # (1) dynamics defect rows, scaled by iSx: d_hat = iSx * (x_{k+1} - A x_k - ... - r);  g = wvc w_k ||Sx d_hat||_1
    for k in range(N - 1):
        t1, c1 = phys(kx=k + 1, Mx=np.eye(nx), const=np.zeros(nx))
        t2, c2 = phys(kx=k, Mx=-ref.A[k], ku=k, Mu=-ref.Bm[k], Mp=-ref.F[k] if np_ else None, const=-ref.r[k])
        t3, c3 = phys(ku=k + 1, Mu=-ref.Bp[k], const=np.zeros(nx))
        terms, const = row_scale(t1 + t2 + t3, c1 + c2 + c3, iSx)
        add_block("l1w", terms, const, weights=pars.wvc * w[k] * Sx, tag=("dyn", k))
    # (2,3) trust regions (identity rows on the scaled variables)
    xh_ref = (ref.xd - cx) / Sx
    uh_ref = (ref.ud - cu) / Su
    for k in range(N):
        add_block("linf", [(ix(k), np.eye(nx))], -xh_ref[k], weight=pars.wtr * w[k], tag=("trx", k))
        add_block("linf", [(iu(k), np.eye(nu))], -uh_ref[k], weight=pars.wtr * w[k], tag=("tru", k))
    if np_ > 0:
        add_block("linf", [(ip, np.eye(np_))], -(ref.p - cp) / Sp, weight=pars.wtr, tag=("trp", 0))
    # (5) linearised non-convex rows, hinge-penalised (vs_k = max(.,0), weight wvc w_k)
    for k in range(N):
        if mdl.ns == 0:
            break
        a = (t[k], k + 1, ref.xd[k], ref.ud[k], ref.p)
        s, C, D, G = mdl.s(*a), mdl.C(*a), mdl.D(*a), mdl.G(*a)
        r = s - C @ ref.xd[k] - D @ ref.ud[k] - (G @ ref.p if np_ else 0.0)
        terms, const = phys(kx=k, Mx=C, ku=k, Mu=D, Mp=G if np_ else None, const=r)
        # row normalisation e_i (any e > 0 is exact: g(y) = (w/e) max(y, 0) on the scaled row)
        nrm = np.sqrt(sum((M * M).sum(axis=1) for _, M in terms))
        e = 1.0 / np.maximum(nrm, 1e-12)
        terms, const = row_scale(terms, const, e)
        add_block("hinge", terms, const, weights=pars.wvc * w[k] / e, tag=("s", k))
    # (6) convex sets
    def add_set(rows, k, is_x):
        for kind, M, Mp, m0 in rows:
            terms, const = phys(kx=k, Mx=M if is_x else None, ku=k, Mu=None if is_x else M, Mp=Mp, const=m0)
            if kind == "NONPOS":
                nrm = np.sqrt(sum((Mm * Mm).sum(axis=1) for _, Mm in terms))
                e = 1.0 / np.maximum(nrm, 1e-12)
                terms, const = row_scale(terms, const, e)
                add_block("nonpos", terms, const, tag=("set", k))
            else:
                mx = max(np.abs(Mm).max() for _, Mm in terms)
                e = np.full(len(const), 1.0 / max(mx, 1e-12))
                terms, const = row_scale(terms, const, e)
                add_block("soc", terms, const, tag=("set", k))
    for k in range(N):
        add_set(mdl.X(t[k], k + 1), k, True)
        add_set(mdl.U(t[k], k + 1), k, False)
    # (7) boundary conditions, L1-penalised
    gic = mdl.gic(ref.xd[0], ref.p, pp); H0 = mdl.H0(ref.xd[0], ref.p, pp); K0 = mdl.K0(ref.xd[0], ref.p, pp)
    l0 = gic - H0 @ ref.xd[0] - (K0 @ ref.p if np_ else 0.0)
    terms, const = phys(kx=0, Mx=H0, Mp=K0 if np_ else None, const=l0)
    nrm = np.sqrt(sum((Mm * Mm).sum(axis=1) for _, Mm in terms)); e = 1.0 / np.maximum(nrm, 1e-12)
    terms, const = row_scale(terms, const, e)
    add_block("l1w", terms, const, weights=pars.wvc / e, tag=("ic", 0))
    gtc = mdl.gtc(ref.xd[-1], ref.p, pp); Hf = mdl.Hf(ref.xd[-1], ref.p, pp); Kf = mdl.Kf(ref.xd[-1], ref.p, pp)
    lf = gtc - Hf @ ref.xd[-1] - (Kf @ ref.p if np_ else 0.0)
    terms, const = phys(kx=N - 1, Mx=Hf, Mp=Kf if np_ else None, const=lf)
    nrm = np.sqrt(sum((Mm * Mm).sum(axis=1) for _, Mm in terms)); e = 1.0 / np.maximum(nrm, 1e-12)
    terms, const = row_scale(terms, const, e)
    add_block("l1w", terms, const, weights=pars.wvc / e, tag=("tc", 0))

    K = sp.csc_matrix((np.concatenate(rows_v), (np.concatenate(rows_r), np.concatenate(rows_c))), shape=(m[0], n))
    c = np.concatenate(consts)
    # cost: Gamma = sum Qu u^2 + lu'u + lx'x (trapz), phi = tx'x_N + tp'p + Qp p^2
    ct = mdl.cost_terms()
    Qd = np.zeros(n); q = np.zeros(n); const = 0.0
    for k in range(N):
        Qd[iu(k)] += 2 * w[k] * ct["Qu"] * Su * Su
        q[iu(k)] += w[k] * (2 * ct["Qu"] * cu * Su + ct["lu"] * Su)
        q[ix(k)] += w[k] * ct["lx"] * Sx
        const += w[k] * (ct["Qu"] @ (cu * cu) + ct["lu"] @ cu + ct["lx"] @ cx)
    q[ix(N - 1)] += ct["tx"] * Sx
    const += ct["tx"] @ cx
    if np_ > 0:
        q[ip] += ct["tp"] * Sp + 2 * ct["Qp"] * cp * Sp
        Qd[ip] += 2 * ct["Qp"] * Sp * Sp
        const += ct["tp"] @ cp + ct["Qp"] @ (cp * cp)
    R = Reduced()
    R.K, R.c, R.blocks, R.Qd, R.q, R.cost_const = K, c, blocks, Qd, q, const
    R.n, R.m, R.N, R.nx, R.nu, R.np, R.nz = n, m[0], N, nx, nu, np_, nz
    R.z_ref = np.concatenate([np.concatenate([xh_ref[k], uh_ref[k]]) for k in range(N)] +
                             [(ref.p - cp) / Sp if np_ else np.zeros(0)])
    R.scale, R.pars, R.w = scale, pars, w
    return R


def prox_blocks(R, v, rho):
    """y = prox_{g/rho}(v) block by block (rho: per-row vector, uniform inside linf/soc blocks)."""
    y = np.empty_like(v)
    for b in R.blocks:
        s, l = b["start"], b["len"]
        vb, rb = v[s:s + l], rho[s:s + l]
        kind = b["kind"]
        if kind == "l1w":
            th = b["weights"] / rb
            y[s:s + l] = np.sign(vb) * np.maximum(np.abs(vb) - th, 0.0)
        elif kind == "hinge":
            th = b["weights"] / rb
            y[s:s + l] = np.where(vb <= 0, vb, np.where(vb >= th, vb - th, 0.0))
        elif kind == "linf":
            y[s:s + l] = vb - proj_l1_ball(vb, b["weight"] / rb[0])
        elif kind == "nonpos":
            y[s:s + l] = np.minimum(vb, 0.0)
        elif kind == "soc":
            y[s:s + l] = proj_soc(vb)
        else:
            raise KeyError(kind)
    return y


def g_value(R, v):
    tot = 0.0
    for b in R.blocks:
        s, l = b["start"], b["len"]
        vb = v[s:s + l]
        kind = b["kind"]
        if kind == "l1w":
            tot += b["weights"] @ np.abs(vb)
        elif kind == "hinge":
            tot += b["weights"] @ np.maximum(vb, 0.0)
        elif kind == "linf":
            tot += b["weight"] * np.max(np.abs(vb))
    return tot


def default_rho(R, rho0):
    rho = np.full(R.m, rho0)
    return rho


def admm(R, z0=None, lam0=None, rho0=1.0, sigma=1e-6, alpha=1.6, iters=500, eps=1e-6, adapt=True, verbose=False,
         check_every=25):
    """OSQP-style ADMM on min 1/2 z'Qz + q'z + g(Kz + c).  Returns dict."""
    K, c = R.K, R.c
    KT = K.T.tocsc()
    z = R.z_ref.copy() if z0 is None else z0.copy()
    rho = default_rho(R, rho0)
    y = prox_blocks(R, K @ z + c, rho)
    lam = np.zeros(R.m) if lam0 is None else lam0.copy()

    def factor(rho):
        M = sp.diags(R.Qd + sigma) + KT @ sp.diags(rho) @ K
        return spla.splu(M.tocsc())
    lu = factor(rho)
    nfac = 1
    hist = []
    for it in range(1, iters + 1):
        rhs = sigma * z - R.q + KT @ (rho * (y - c) - lam)
        zt = lu.solve(rhs)
        Kz = K @ zt + c
        v = alpha * Kz + (1 - alpha) * y + lam / rho
        y_new = prox_blocks(R, v, rho)
        lam = lam + rho * (alpha * Kz + (1 - alpha) * y - y_new)
        z = alpha * zt + (1 - alpha) * z
        y = y_new
        if it % check_every == 0 or it == iters:
            Kzc = K @ z + c
            rp = np.max(np.abs(Kzc - y))
            rd = np.max(np.abs(R.Qd * z + R.q + KT @ lam))
            sp_ = max(np.max(np.abs(Kzc)), np.max(np.abs(y)), 1e-12)
            sd_ = max(np.max(np.abs(R.Qd * z)), np.max(np.abs(KT @ lam)), np.max(np.abs(R.q)), 1e-12)
            hist.append((it, rp, rd))
            if verbose:
                print("it %5d rp %.3e rd %.3e obj % .8e rho %.3g" % (it, rp, rd, objective(R, z), rho[0]))
            if rp <= eps * (1 + sp_) and rd <= eps * (1 + sd_):
                break
            if adapt:
                ratio = np.sqrt((rp / sp_) / max(rd / sd_, 1e-30))
                if ratio > 5 or ratio < 0.2:
                    rho = np.clip(rho * ratio, 1e-6, 1e6)
                    lu = factor(rho)
                    nfac += 1
    return dict(z=z, y=y, lam=lam, iters=it, hist=hist, nfac=nfac, rho=rho)


def objective(R, z):
    return 0.5 * z @ (R.Qd * z) + R.q @ z + R.cost_const + g_value(R, R.K @ z + R.c)


def unpack(R, z):
    N, nx, nu, nz = R.N, R.nx, R.nu, R.nz
    s = R.scale
    Z = z[: N * nz].reshape(N, nz)
    x = Z[:, :nx] * s.Sx + s.cx
    u = Z[:, nx:] * s.Su + s.cu
    p = z[N * nz:] * s.Sp + s.cp if R.np else np.zeros(0)
    return x, u, p
