"""TEST INFRASTRUCTURE: ctypes binding of oracle/_build/libconic_host.so, the HOST build of the product's generic conic
solver body (scptoolbox.jl_amd/csrc/conic_ipm.hpp + conic_symbolic.hpp).  Lets tests/test_conic_cpu.py check the
solver's numerics without a GPU; never imported by the product."""
import ctypes
import os

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Opts(ctypes.Structure):
    _fields_ = [("max_iter", ctypes.c_int), ("feastol", ctypes.c_double), ("abstol", ctypes.c_double),
                ("reltol", ctypes.c_double), ("reg", ctypes.c_double), ("dyn_eps", ctypes.c_double),
                ("dyn_delta", ctypes.c_double), ("nref", ctypes.c_int), ("ref_tol", ctypes.c_double),
                ("step", ctypes.c_double)]


def default_opts(**kw):
    o = Opts(100, 1e-8, 1e-8, 1e-8, -1.0, 1e-13, 2e-7, 10, 1e-11, 0.99)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(os.environ.get("CONIC_HOST_LIB", os.path.join(_HERE, "_build", "libconic_host.so")))
    return _LIB


def csc_parts(M, shape, upper=False):
    """(indptr, indices, order) of the canonical (sorted, deduplicated) CSC pattern of M."""
    if M is None:
        return np.zeros(shape[1] + 1, np.int32), np.zeros(0, np.int32), None
    M = sp.csc_matrix(M)
    if upper:
        M = sp.triu(M, format="csc")
    M.sum_duplicates(); M.sort_indices()
    return M.indptr.astype(np.int32), M.indices.astype(np.int32), M


def ip(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def solve(c, G, h, l, q, A=None, b=None, P=None, B=None, values=None, shared_mask=0, perm=None, **optkw):
    """CONIC_HOST_ORDER = auto emulates Engine::launch (conic_api.hip): nested-dissection schedule first, every problem it
    does not bring to OPTIMAL / a certificate re-solved with the sequential schedule.  See _solve."""
    mode = os.environ.get("CONIC_HOST_ORDER", "seq")
    if mode != "auto" or perm is not None:
        return _solve(c, G, h, l, q, A, b, P, B, values, shared_mask, perm, **optkw)
    os.environ["CONIC_HOST_ORDER"] = "nd"
    try:
        r = _solve(c, G, h, l, q, A, b, P, B, values, shared_mask, None, **optkw)
        one = B is None
        st = np.atleast_1d(r["status"])
        bad = np.nonzero((st >= 1) & (st <= 3))[0]
        r["fallback"] = bad.size
        if r["stats"][4] == 0 or bad.size == 0:
            return r
        os.environ["CONIC_HOST_ORDER"] = "seq"
        if one:
            r2 = _solve(c, G, h, l, q, A, b, P, B, values, shared_mask, None, **optkw)
            r2["fallback"] = 1
            return r2
        keys = {"c": 1, "b": 2, "h": 4, "Gx": 8, "Ax": 16, "Px": 32}
        sub = {k: (v if (shared_mask & keys[k]) else np.asarray(v)[bad]) for k, v in (values or {}).items()}
        r2 = _solve(c, G, h, l, q, A, b, P, bad.size, sub, shared_mask, None, **optkw)
        for k in ("x", "y", "z", "s", "status", "iters", "info", "pcost", "dcost", "gap", "pres", "dres"):
            r[k][bad] = r2[k]
        return r
    finally:
        os.environ["CONIC_HOST_ORDER"] = mode


def analyse(T):
    """Symbolic analysis only (B = 0) of a subproblem template (subproblem.py): [nnzL, multiply-adds, KKT dimension, nnz(Gt),
    dissection depth, elimination levels, backward levels, 0] under the ordering CONIC_HOST_ORDER selects."""
    ones = lambda M: sp.csc_matrix((np.ones(len(M.indices)), M.indices, M.indptr), shape=M.shape)
    Gp, Gi, _ = csc_parts(ones(T.G), T.G.shape)
    Ap, Ai, _ = csc_parts(ones(T.A), T.A.shape)
    Pp, Pi, _ = csc_parts(ones(T.P), T.P.shape, upper=True)
    qa = np.asarray(T.q, np.int32)
    stats = np.zeros(8, np.int64)
    rc = lib().conic_host_solve(
        ctypes.c_int(T.n), ctypes.c_int(T.p), ctypes.c_int(T.m), ctypes.c_int(int(T.l)), ctypes.c_int(len(qa)), ip(qa),
        ip(Pp), ip(Pi), ip(Ap), ip(Ai), ip(Gp), ip(Gi), None, ctypes.c_int(0), None, None, None, None, None, None,
        ctypes.c_uint(0), None, None, None, None, None, None, None, None, ip(stats))
    if rc != 0:
        raise ValueError("conic_host_solve: bad pattern")
    return stats


def schedule_profile(T):
    """per elimination level [columns, row entries, longest row, L entries, operand pairs, longest pair list, critical path of
    the pivot / forward phase with the long rows chunked, the same for the entry phase] of a template's schedule under
    CONIC_HOST_ORDER (conic_host_schedule_profile)"""
    ones = lambda M: sp.csc_matrix((np.ones(len(M.indices)), M.indices, M.indptr), shape=M.shape)
    Gp, Gi, _ = csc_parts(ones(T.G), T.G.shape)
    Ap, Ai, _ = csc_parts(ones(T.A), T.A.shape)
    Pp, Pi, _ = csc_parts(ones(T.P), T.P.shape, upper=True)
    qa = np.asarray(T.q, np.int32)
    args = (ctypes.c_int(T.n), ctypes.c_int(T.p), ctypes.c_int(T.m), ctypes.c_int(int(T.l)), ctypes.c_int(len(qa)), ip(qa),
            ip(Pp), ip(Pi), ip(Ap), ip(Ai), ip(Gp), ip(Gi))
    nlev = lib().conic_host_schedule_profile(*args, None, ctypes.c_int(0))
    if nlev < 0:
        raise ValueError("conic_host_schedule_profile: bad pattern")
    prof = np.zeros((nlev, 8), np.int64)
    lib().conic_host_schedule_profile(*args, ip(prof), ctypes.c_int(nlev))
    return prof


def _solve(c, G, h, l, q, A=None, b=None, P=None, B=None, values=None, shared_mask=0, perm=None, **optkw):
    """Solve one program (B None) or a batch: `values` = dict of per-problem value arrays [B, len] overriding the pattern
    matrices' own values (keys c, b, h, Gx, Ax, Px).  Returns dict of arrays."""
    c = np.asarray(c, float)
    n = c.shape[-1]
    m = int(l + sum(abs(int(v)) for v in q))      # q[c] = -3: exponential cone (3 rows)
    Gp, Gi, Gm = csc_parts(G, (m, n))
    pe = 0 if A is None else sp.csc_matrix(A).shape[0]
    Ap, Ai, Am = csc_parts(A, (pe, n))
    Pp, Pi, Pm = csc_parts(P, (n, n), upper=True)
    one = B is None
    Bn = 1 if one else B
    vals = dict(values or {})

    def arr(key, default, ln):
        v = vals.get(key)
        if v is None:
            v = np.asarray(default, float).reshape(-1)
            if not (shared_mask & {"c": 1, "b": 2, "h": 4, "Gx": 8, "Ax": 16, "Px": 32}[key]):
                v = np.tile(v, (Bn, 1))
        return np.ascontiguousarray(v, dtype=float)
    ca = arr("c", c, n)
    ba = arr("b", np.zeros(0) if b is None else b, pe)
    ha = arr("h", h, m)
    Gxa = arr("Gx", Gm.data if Gm is not None else np.zeros(0), len(Gi))
    Axa = arr("Ax", Am.data if Am is not None else np.zeros(0), len(Ai))
    Pxa = arr("Px", Pm.data if Pm is not None else np.zeros(0), len(Pi))
    qa = np.asarray(q, np.int32)
    x = np.zeros((Bn, n)); y = np.zeros((Bn, pe)); z = np.zeros((Bn, m)); s = np.zeros((Bn, m))
    status = np.zeros(Bn, np.int32); iters = np.zeros(Bn, np.int32); info = np.zeros((Bn, 8)); stats = np.zeros(8, np.int64)
    o = default_opts(**optkw)
    pa = None if perm is None else np.ascontiguousarray(perm, np.int32)
    rc = lib().conic_host_solve(
        ctypes.c_int(n), ctypes.c_int(pe), ctypes.c_int(m), ctypes.c_int(int(l)), ctypes.c_int(len(qa)), ip(qa),
        ip(Pp), ip(Pi), ip(Ap), ip(Ai), ip(Gp), ip(Gi), ip(pa) if pa is not None else None, ctypes.c_int(Bn),
        ip(ca), ip(ba), ip(ha), ip(Gxa), ip(Axa), ip(Pxa), ctypes.c_uint(shared_mask), ctypes.byref(o),
        ip(x), ip(y), ip(z), ip(s), ip(status), ip(iters), ip(info), ip(stats))
    if rc != 0:
        raise ValueError("conic_host_solve: bad pattern")
    out = dict(x=x, y=y, z=z, s=s, status=status, iters=iters, info=info, stats=stats,
               pcost=info[:, 0], dcost=info[:, 1], gap=info[:, 2], pres=info[:, 3], dres=info[:, 4])
    if one:
        out = {k: (v[0] if k != "stats" else v) for k, v in out.items()}
    return out
