"""Generic `solve_subproblem!` on the device for any conic template (scptoolbox.jl_amd/subproblem.py).

`GenericSubproblem` binds a template to an `SCPProblem` handle (scp_sub_create): the pattern is analysed once, the
affine value maps and read-out indices are uploaded, and every `solve` is discretize! -> linearise -> gather ->
conic_ipm_kernel -> read-out -> discretize! on the MI355X (scp_sub_solve_batch_host).  The reference does the same work
per iteration and per problem in Julia/JuMP + ECOS (src/solvers/scp.jl:942-950 and the add_*! functions)."""
import ctypes

import numpy as np

from . import _lib
from .conic import default_options
from .subproblem import trapz_weights


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _amap(m, keep):
    am = _lib.ScpAffineMap()
    arrs = [np.ascontiguousarray(m.const, np.float64), np.ascontiguousarray(m.ptr, np.int32),
            np.ascontiguousarray(m.src, np.int32), np.ascontiguousarray(m.coef, np.float64)]
    keep.extend(arrs)
    am.len = int(m.size)
    am.val0, am.ptr, am.sidx, am.coef = (a.ctypes.data for a in arrs)
    return am


class LinearFunctionals:
    """fun[j] = const[j] + sum coef * x[idx]: rows over the conic solution."""

    def __init__(self, n):
        self.n = n
        self.rows = []

    def add(self, idx, coef, const=0.0):
        self.rows.append((np.atleast_1d(idx).astype(np.int64), np.atleast_1d(np.asarray(coef, float)), float(const)))
        return len(self.rows) - 1

    def as_map(self):
        from .affine import AffineMap
        const = np.array([r[2] for r in self.rows])
        slot = np.concatenate([np.full(r[0].size, j) for j, r in enumerate(self.rows)]) if self.rows else np.zeros(0, np.int64)
        src = np.concatenate([r[0] for r in self.rows]) if self.rows else np.zeros(0, np.int64)
        coef = np.concatenate([r[1] for r in self.rows]) if self.rows else np.zeros(0)
        return AffineMap.from_terms(const, slot, src, coef)


def standard_functionals(T):
    """Cost pieces the outer loops read from the solved subproblem (ptr.jl:773-789,889-892; scvx.jl:895-898):
    fun[0] = trapz(P) + sum(Pf);  PTR additionally fun[1] = trapz(eta_x) + trapz(eta_u) + eta_p."""
    F = LinearFunctionals(T.n)
    w = trapz_weights(T.N)
    v = T.variables
    if "P" in v:
        F.add(np.concatenate([v["P"], v["Pf"]]), np.concatenate([w, np.ones(2)]))
    if getattr(T, "algo", "") == "gusto":      # the penalty variables themselves: v_tr[k], then v_st[k][i] (gusto.jl:534-550)
        for k in range(T.N):
            F.add(v["v_tr"][k], 1.0)
        for k in range(T.N):
            for i in range(T.nst):
                F.add(T.v_st_nodes[k, i], 1.0)
    if "etax" in v:
        F.add(np.concatenate([v["etax"], v["etau"], v["etap"]]), np.concatenate([w, w, np.ones(1)]))
    return F


class GenericSubproblem:
    def __init__(self, pbm, T, functionals=None):
        self.pbm, self.T = pbm, T
        L = _lib.lib()
        nscal = T.sources.segs["scal"][1][0]
        offs = np.zeros(21, np.int32)
        nsrc = ctypes.c_int(0)
        _lib.check(L.scp_sub_source_layout(pbm.handle, nscal, _ptr(offs), ctypes.byref(nsrc)), pbm.handle)
        mine = [T.sources.segs[k][0] for k in ("xref", "uref", "pref", "A", "Bm", "Bp", "F", "r", "E", "C", "D", "Gs", "rs",
                                               "H0", "K0", "l0", "Hf", "Kf", "lf", "scal")] + [T.sources.n]
        if list(offs) != mine or nsrc.value != T.nsrc:
            raise RuntimeError("source layout of the template does not match the library's (scp_sub_source_layout)")
        self.F = functionals if functionals is not None else standard_functionals(T)
        keep = []
        st = _lib.ScpSubTemplate()
        st.n, st.p, st.m, st.l, st.ncones = T.n, T.p, T.m, T.l, len(T.q)
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        arrs = dict(q=i32(T.q), Pp=i32(T.P.indptr), Pi=i32(T.P.indices), Ap=i32(T.A.indptr), Ai=i32(T.A.indices),
                    Gp=i32(T.G.indptr), Gi=i32(T.G.indices))
        N = T.N
        arrs["ix"] = i32(T.variables["xh"])     # xh blocks were created node by node: [N, nx] row-major == (nx, N) col-major
        arrs["iu"] = i32(T.variables["uh"])
        arrs["ip"] = i32(T.variables.get("ph", np.zeros(0, np.int64)))
        for k, a in arrs.items():
            setattr(st, k, a.ctypes.data)
        keep.extend(arrs.values())
        for k in ("c", "b", "h", "Gx", "Ax", "Px"):
            setattr(st, k, _amap(T.maps[k], keep))
        st.nsrc, st.nscal = T.nsrc, nscal
        fm = self.F.as_map()
        st.nfun = fm.size
        st.fun = _amap(fm, keep)
        self.nfun, self.nscal = fm.size, nscal
        self._keep = keep
        self._h = ctypes.c_void_p()
        rc = L.scp_sub_create(pbm.handle, ctypes.byref(st), ctypes.byref(self._h))
        if rc != 0:
            self._h = ctypes.c_void_p()
            _lib.check(rc, pbm.handle)
        if not hasattr(pbm, "_children"):
            pbm._children = []
        pbm._children.append(self)       # the problem handle must outlive this one (SCPProblem.close closes us first)

    def close(self):
        if self._h:
            _lib.lib().scp_sub_destroy(self._h)
            self._h = ctypes.c_void_p()
        if self in getattr(self.pbm, "_children", []):
            self.pbm._children.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise _lib.ScpError(rc, _lib.lib().scp_sub_last_error(self._h).decode(errors="replace"))

    def stats(self):
        """symbolic / run statistics of this subproblem's conic engine (scp_sub_stats: the fields of ConicProgramBatch.stats)"""
        st = np.zeros(16, np.int64)
        self._check(_lib.lib().scp_sub_stats(self._h, _ptr(st)))
        return dict(nnzL=int(st[0]), factor_madds=int(st[1]), kkt_dim=int(st[2]), nnzGt=int(st[3]), bytes_per_problem=int(st[4]),
                    levels=int(st[5]), back_levels=int(st[6]), waves=int(st[7]), nd_depth=int(st[8]), fallback_solves=int(st[9]),
                    solves=int(st[10]), fallback_levels=int(st[11]), fallback_rescued=int(st[12]))

    def solve(self, xd, ud, p, pp=None, scal=None, want_conic=False, **opts):
        """`solve_subproblem!` about the reference trajectories xd[B,N,nx], ud[B,N,nu], p[B,np]."""
        pbm = self.pbm
        xd = np.ascontiguousarray(xd, np.float64); ud = np.ascontiguousarray(ud, np.float64)
        p = np.ascontiguousarray(p, np.float64)
        B, N = xd.shape[0], pbm.pars.N
        npp = pbm.info.npp
        pp = np.ascontiguousarray(np.tile(pbm.traj.mdl.nominal_pp(), (B, 1)) if pp is None else pp, np.float64)
        assert pp.shape == (B, npp)
        scal = np.zeros((B, self.nscal)) if scal is None else np.ascontiguousarray(scal, np.float64).reshape(B, self.nscal)
        x = np.zeros((B, N, pbm.nx)); u = np.zeros((B, N, pbm.nu)); po = np.zeros((B, pbm.np))
        fun = np.zeros((B, self.nfun)); xc = np.zeros((B, self.T.n)) if want_conic else None
        status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); info = np.zeros((B, 8))
        defect = np.zeros((B, N - 1, pbm.nx)); feas = np.zeros(B, np.uint8)
        sec = ctypes.c_double(0.0)
        o = default_options(**opts)
        self._check(_lib.lib().scp_sub_solve_batch_host(
            self._h, B, _ptr(xd), _ptr(ud), _ptr(p) if pbm.np else None, _ptr(pp) if npp else None, _ptr(scal),
            ctypes.byref(o), _ptr(x), _ptr(u), _ptr(po) if pbm.np else None, _ptr(fun), _ptr(xc), _ptr(status), _ptr(iters),
            _ptr(info), _ptr(defect), _ptr(feas), ctypes.byref(sec)))
        out = dict(x=x, u=u, p=po, fun=fun, status=status, iters=iters, pcost=info[:, 0] + self.T.cost_const, gap=info[:, 2],
                   pres=info[:, 3], dres=info[:, 4], defect=defect, feas=feas.astype(bool), seconds=sec.value)
        if want_conic:
            out["xconic"] = xc
            for name in ("vd", "vs", "vic", "vtc", "P", "Pf", "etax", "etau", "etap", "dx_lq", "du_lq", "dp_lq"):
                if name in self.T.variables:
                    out[name] = xc[:, self.T.variables[name]]
        return out
