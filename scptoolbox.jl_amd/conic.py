"""Host-side mirror of the reference's `ConicProgram` solver seam for BATCHES of programs with one sparsity pattern.

Reference: `ConicProgram(...; solver, solver_options)` + `solve!(prg)` (src/parser/program.jl:63-76,419-424) hand the
program to ECOS in the standard form

    min 1/2 x'Px + c'x   s.t.  A x = b,   G x + s = h,   s in R+^l x Q^{q_1} x ...

`ConicProgramBatch` is the object behind `pars.solver` on the MI355X side: the pattern is analysed once at
construction (`scp_conic_create`, the `ECOS_setup` of a whole SCP run) and `solve` ships only values
(`scp_conic_solve_batch_host`).  Status codes are the MOI termination codes the reference inspects
(src/solvers/scp.jl:965-980).  Everything runs in libscp_mi355x.so; there is no CPU fallback.
"""
import ctypes

import numpy as np
import scipy.sparse as sp

from . import _lib

OPTIMAL, ALMOST_OPTIMAL, ITERATION_LIMIT, NUMERICAL_ERROR, INFEASIBLE, DUAL_INFEASIBLE = range(6)
STATUS_NAMES = ("OPTIMAL", "ALMOST_OPTIMAL", "ITERATION_LIMIT", "NUMERICAL_ERROR", "INFEASIBLE", "DUAL_INFEASIBLE")
SHARED_BITS = {"c": 1, "b": 2, "h": 4, "Gx": 8, "Ax": 16, "Px": 32}


def _canon(M, shape, upper=False):
    """canonical CSC (sorted, no duplicates; explicit zeros kept: they are pattern entries)."""
    if M is None:
        return sp.csc_matrix(shape)
    M = sp.csc_matrix(M)
    if M.shape != tuple(shape):
        raise ValueError("matrix shape %s, expected %s" % (M.shape, tuple(shape)))
    if upper:
        M = sp.csc_matrix(sp.triu(M))
    M.sum_duplicates()
    M.sort_indices()
    return M


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def default_options(**kw):
    o = _lib.ScpConicOpts()
    _lib.lib().scp_conic_default_opts(ctypes.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError("unknown solver option %r" % k)
        setattr(o, k, v)
    return o


class ConicProgramBatch:
    """A batch of conic programs sharing the pattern of (P, A, G) and the cone (l, q)."""

    def __init__(self, n, G, l, q, A=None, P=None, batch_capacity=1, perm=None, device=0):
        self.n = int(n)
        self.l = int(l)
        self.q = np.asarray(q, np.int32).reshape(-1)
        self.m = self.l + int(np.abs(self.q).sum())      # q[c] = -3: exponential cone (include/scp_conic.h)
        self.G = _canon(G, (self.m, self.n))
        self.p = 0 if A is None else sp.csc_matrix(A).shape[0]
        self.A = _canon(A, (self.p, self.n))
        self.P = _canon(P, (self.n, self.n), upper=True)
        self.cap = int(batch_capacity)
        self._h = ctypes.c_void_p()
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        self._keep = [i32(self.P.indptr), i32(self.P.indices), i32(self.A.indptr), i32(self.A.indices),
                      i32(self.G.indptr), i32(self.G.indices), self.q, None if perm is None else i32(perm)]
        Pp, Pi, Ap, Ai, Gp, Gi, qq, pm = self._keep
        rc = _lib.lib().scp_conic_create(self.n, self.p, self.m, self.l, len(self.q), _ptr(qq), _ptr(Pp), _ptr(Pi),
                                         _ptr(Ap), _ptr(Ai), _ptr(Gp), _ptr(Gi), _ptr(pm), self.cap, int(device),
                                         ctypes.byref(self._h))
        if rc != 0:
            self._h = ctypes.c_void_p()
            raise _lib.ScpError(rc, "scp_conic_create")

    # -- bookkeeping -------------------------------------------------------------------------------------------
    def close(self):
        if self._h:
            _lib.lib().scp_conic_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise _lib.ScpError(rc, _lib.lib().scp_conic_last_error(self._h).decode(errors="replace"))

    def stats(self):
        st = np.zeros(16, np.int64)
        self._check(_lib.lib().scp_conic_stats(self._h, _ptr(st)))
        return dict(nnzL=int(st[0]), factor_madds=int(st[1]), kkt_dim=int(st[2]), nnzGt=int(st[3]),
                    bytes_per_problem=int(st[4]), levels=int(st[5]), back_levels=int(st[6]), waves=int(st[7]),
                    nd_depth=int(st[8]), fallback_solves=int(st[9]), solves=int(st[10]), fallback_levels=int(st[11]), fallback_rescued=int(st[12]))

    # -- solve -------------------------------------------------------------------------------------------------
    def solve(self, c, h, b=None, Gx=None, Ax=None, Px=None, shared=(), B=None, **opts):
        """Solve B programs.  Each value array is [B, len] (one row per problem) or, when its name is in `shared`,
        [len]; Gx/Ax/Px default to the values of the pattern matrices given at construction (shared).  Returns a dict
        with x[B,n], y[B,p], z[B,m], s[B,m], status[B], iters[B], pcost/dcost/gap/pres/dres/relgap[B], seconds."""
        shared = set(shared)
        vals = {"c": c, "h": h, "b": b, "Gx": Gx, "Ax": Ax, "Px": Px}
        lens = {"c": self.n, "h": self.m, "b": self.p, "Gx": self.G.nnz, "Ax": self.A.nnz, "Px": self.P.nnz}
        defaults = {"Gx": self.G.data, "Ax": self.A.data, "Px": self.P.data, "b": np.zeros(self.p)}
        for k in ("Gx", "Ax", "Px", "b"):
            if vals[k] is None:
                vals[k] = defaults[k]
                shared.add(k)
        if B is None:
            B = next((np.asarray(vals[k]).shape[0] for k in vals if k not in shared), 1)
        mask = 0
        arrs = {}
        for k, v in vals.items():
            a = np.ascontiguousarray(v, dtype=np.float64)
            if k in shared:
                if a.size != lens[k]:
                    raise ValueError("%s: expected %d values, got %d" % (k, lens[k], a.size))
                mask |= SHARED_BITS[k]
            elif a.shape != (B, lens[k]):
                raise ValueError("%s: expected shape (%d, %d), got %s" % (k, B, lens[k], a.shape))
            arrs[k] = a
        o = default_options(**opts)
        x = np.zeros((B, self.n)); y = np.zeros((B, self.p)); z = np.zeros((B, self.m)); s = np.zeros((B, self.m))
        status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); info = np.zeros((B, 8))
        sec = ctypes.c_double(0.0)
        self._check(_lib.lib().scp_conic_solve_batch_host(
            self._h, int(B), _ptr(arrs["c"]), _ptr(arrs["b"]), _ptr(arrs["h"]), _ptr(arrs["Gx"]), _ptr(arrs["Ax"]),
            _ptr(arrs["Px"]), mask, ctypes.byref(o), _ptr(x), _ptr(y), _ptr(z), _ptr(s), _ptr(status), _ptr(iters),
            _ptr(info), ctypes.byref(sec)))
        return dict(x=x, y=y, z=z, s=s, status=status, iters=iters, pcost=info[:, 0], dcost=info[:, 1], gap=info[:, 2],
                    pres=info[:, 3], dres=info[:, 4], relgap=info[:, 5], dyn_regs=info[:, 6], refinements=info[:, 7],
                    seconds=sec.value)


def socp_solve_batch(c, G, h, l, q, A=None, b=None):
    """One-shot `socp_solve_batch`: c[B,n], h[B,m], b[B,p]; G, A scipy matrices whose values are
    broadcast to the batch.  Returns (x, y, s, z, status)."""
    c = np.ascontiguousarray(c, np.float64)
    B, n = c.shape
    q = np.asarray(q, np.int32).reshape(-1)
    m = int(l) + int(np.abs(q).sum())
    Gm = _canon(G, (m, n))
    p = 0 if A is None else sp.csc_matrix(A).shape[0]
    Am = _canon(A, (p, n))
    h = np.ascontiguousarray(h, np.float64)
    b = np.zeros((B, p)) if b is None else np.ascontiguousarray(b, np.float64)
    Gx = np.ascontiguousarray(np.tile(Gm.data, (B, 1)))
    Ax = np.ascontiguousarray(np.tile(Am.data, (B, 1)))
    i32 = lambda a: np.ascontiguousarray(a, np.int32)
    Gp, Gi, Ap, Ai = i32(Gm.indptr), i32(Gm.indices), i32(Am.indptr), i32(Am.indices)
    x = np.zeros((B, n)); y = np.zeros((B, p)); s = np.zeros((B, m)); z = np.zeros((B, m)); st = np.zeros(B, np.int32)
    rc = _lib.lib().socp_solve_batch(n, m, p, int(l), len(q), _ptr(q), _ptr(Gp), _ptr(Gi), _ptr(Gx), _ptr(Ap), _ptr(Ai),
                                     _ptr(Ax), _ptr(c), _ptr(h), _ptr(b), B, _ptr(x), _ptr(y), _ptr(s), _ptr(z), _ptr(st))
    if rc != 0:
        raise _lib.ScpError(rc, "socp_solve_batch")
    return x, y, s, z, st
