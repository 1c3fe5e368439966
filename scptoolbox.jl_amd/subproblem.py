"""Host-side formulation of the SCP convex subproblems as conic TEMPLATES (pattern + affine value maps).

Mirrors, function by function, the formulation code of the reference -- but records each coefficient as an affine
function of the per-problem sources (csrc/scp_generic.hip fills them on the device) instead of building a JuMP model
per iteration:

    add_dynamics!                    src/solvers/scp.jl:657-674  -> state_update!, discretization.jl:424-497
    add_convex_state/input_...!      src/solvers/scp.jl:685-734  (the model's X / U sets, via scp_model_rows)
    add_nonconvex_constraints!       src/solvers/scp.jl:744-794
    add_bcs!                         src/solvers/scp.jl:808-895
    PTR  add_trust_region!, cost     src/solvers/ptr.jl:565-743, 753-895
    SCvx add_trust_region!, cost     src/solvers/scvx.jl:578-678, 688-698, 804-901
    correct_convex!                  src/solvers/scp.jl:275-361

Variables are the reference's scaled blocks (x = Sx xh + cx, src/parser/block.jl:368-394, scaling.jl:134-141); rows are
not rescaled.  The L1 / LINF cones are lowered exactly as MOI's NormOne / NormInfinity bridges do (ECOS has neither).

Source segments (per problem, column-major, see `standard_sources`): the reference trajectory, ref.dyn, the
linearisation of s and of the boundary conditions about the reference, and the algorithm's scalars (SCvx: eta).
"""
import ctypes
import copy
import functools
import hashlib
import os
import pickle

import numpy as np

from . import _lib
from .affine import Aff, ConicAssembler, Sources
from .models import MODEL_IDS, linrange


class ModelRows:
    """Convex sets and cost of a compiled model, evaluated on the host through the C ABI (scp_model_rows).

    Parameter vector: p = [global (np_glob); node parameters (np_node, N)] (include/scp_mi355x.h, scp_model_info).  The
    library reports parameter Jacobians COMPACTLY (global columns, then the node's own); this class scatters them into the
    full np = np_glob + np_node N columns the formulation works with and declares which columns can be non-zero
    (`s_param_cols`, `bc_param_cols`) so that only those become sources / KKT entries."""

    def __init__(self, mdl, N=None):
        L = _lib.lib()
        self.name = mdl.name
        self.model_id = MODEL_IDS[mdl.name]
        info = _lib.ScpModelInfo()
        _lib.check(L.scp_model_query(self.model_id, ctypes.byref(info)))
        self.info = info
        if not info.has_subproblem:
            raise NotImplementedError("model '%s' is compiled for discretize! / propagate / the initial guess only" % mdl.name)
        self.nx, self.nu = info.nx, info.nu
        self.np_glob, self.np_node, self.npc = info.np, info.np_node, info.np + info.np_node
        self.npF, self.Fcols = info.npF, [info.Fcols[j] for j in range(info.npF)]
        self.ns, self.nic, self.ntc = info.ns, info.nic, info.ntc
        self.nl, self.nsoc, self.ng = info.nl, info.nsoc, info.ng
        self.global_rows_in_X = bool(info.global_rows_in_X)
        self.gusto_ok = bool(info.s_input_free)
        self.par = np.ascontiguousarray(mdl.par(), np.float64)
        assert self.par.size == info.npar, "model parameter blob: %d values, the library expects %d" % (self.par.size, info.npar)
        self.np = None
        if N is not None:
            self.bind(N)
        elif info.np_node == 0:
            self.np = info.np

    def bind(self, N):
        self.N = int(N)
        self.np = self.np_glob + self.np_node * self.N

    def node_cols(self, k):
        """0-based positions in p of the compact parameter columns at node k (1-based)"""
        return np.concatenate([np.arange(self.np_glob), self.np_glob + self.np_node * (k - 1) + np.arange(self.np_node)]).astype(np.int64)

    def s_param_cols(self, N, k):
        return self.node_cols(k)

    def bc_param_cols(self):
        return np.arange(self.np_glob)

    def linf_groups(self, N, k):
        g, r = self.info.linf_groups, self.info.linf_rows
        return [list(range(i * r, (i + 1) * r)) for i in range(g)]

    def state_indicators(self, N):
        nq = ctypes.c_int(0)
        _lib.check(_lib.lib().scp_model_state_indicators(self.model_id, self.par.ctypes.data_as(ctypes.c_void_p), N, ctypes.byref(nq)))
        return nq.value

    def rows(self, N, k):
        """(L, Lp, l, Mm, m) at node k (1-based); Lp over the full parameter vector."""
        nz = self.nx + self.nu
        L = np.zeros((self.nl, nz)); Lpc = np.zeros((self.nl, self.npc)); l = np.zeros(self.nl)
        Mm = np.zeros((4 * self.nsoc, nz)); m = np.zeros(4 * self.nsoc)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(_lib.lib().scp_model_rows(self.model_id, p(self.par), N, k, p(L), p(Lpc), p(l), p(Mm), p(m), None, None,
                                             None))
        Lp = np.zeros((self.nl, self.np_glob + self.np_node * N))
        Lp[:, self.node_cols(k)] = Lpc
        return L, Lp, l, Mm, m

    def global_rows(self, N):
        Lgc = np.zeros((self.ng, self.np_glob)); lg = np.zeros(self.ng)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(_lib.lib().scp_model_rows(self.model_id, p(self.par), N, 1, None, None, None, None, None, p(Lgc), p(lg),
                                             None))
        Lg = np.zeros((self.ng, self.np_glob + self.np_node * N))
        Lg[:, :self.np_glob] = Lgc
        return Lg, lg

    def cost_terms(self, N):
        c = np.zeros(2 * self.nu + 2 * self.nx + 2 * self.npc)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(_lib.lib().scp_model_rows(self.model_id, p(self.par), N, 1, None, None, None, None, None, None, None, p(c)))
        nu, nx, npc, g = self.nu, self.nx, self.npc, self.np_glob
        o = np.cumsum([0, nu, nu, nx, nx, npc, npc])
        full = lambda v: np.concatenate([v[:g], np.tile(v[g:], N)])      # node entries apply to the parameters of every node
        return dict(Qu=c[o[0]:o[1]], lu=c[o[1]:o[2]], lx=c[o[2]:o[3]], tx=c[o[3]:o[4]], tp=full(c[o[4]:o[5]]), Qp=full(c[o[5]:o[6]]))


def standard_sources(mr, N, nscal):
    """The source vector layout shared with csrc/scp_generic.hip (scp_gen_source_layout): order and shapes."""
    S = Sources()
    nx, nu, np_, ns = mr.nx, mr.nu, mr.np, mr.ns
    S.add("xref", (nx, N)); S.add("uref", (nu, N)); S.add("pref", (np_,))
    S.add("A", (nx, nx, N - 1)); S.add("Bm", (nx, nu, N - 1)); S.add("Bp", (nx, nu, N - 1))
    S.add("F", (nx, mr.npF, N - 1)); S.add("r", (nx, N - 1)); S.add("E", (nx, nx, N - 1))
    # parameter Jacobians of s and of the boundary conditions: by default all np columns (the layout the device fills);
    # a model may declare the columns that can be non-zero -- per node for s (`s_param_cols(N, k)`), once for the
    # boundary conditions (`bc_param_cols()`) -- and only those become sources (free-flyer: 6 of its 1 + 6 N parameters per
    # node; a dense declaration couples every node to every parameter in the KKT pattern)
    ngc = len(s_param_cols(mr, N, 1)); nkc = len(bc_param_cols(mr))
    S.add("C", (ns, nx, N)); S.add("D", (ns, nu, N)); S.add("Gs", (ns, ngc, N)); S.add("rs", (ns, N))
    S.add("H0", (mr.nic, nx)); S.add("K0", (mr.nic, nkc)); S.add("l0", (mr.nic,))
    S.add("Hf", (mr.ntc, nx)); S.add("Kf", (mr.ntc, nkc)); S.add("lf", (mr.ntc,))
    S.add("scal", (nscal,))
    return S


def s_param_cols(mr, N, k):
    """parameter columns (0-based) the non-convex constraint s can depend on at node k (1-based); same count at every node
    (compiled models: the global parameters and the node's own, ModelRows.node_cols)"""
    f = getattr(mr, "s_param_cols", None)
    return np.arange(mr.np) if f is None else np.asarray(f(N, k), np.int64)


def bc_param_cols(mr):
    f = getattr(mr, "bc_param_cols", None)
    return np.arange(mr.np) if f is None else np.asarray(f(), np.int64)


def scatter_param_columns(M, cols, np_):
    """(rows x len(cols)) affine matrix -> (rows x np) with column j placed at parameter cols[j] (zeros elsewhere)."""
    rows, nc = M.shape
    if nc == np_ and np.array_equal(cols, np.arange(np_)):
        return M
    newpos = (M.pos // nc) * np_ + np.asarray(cols)[M.pos % nc] if nc else M.pos
    c0 = np.zeros((rows, np_))
    if nc:
        c0[:, cols] = M.c0
    return Aff(c0, newpos, M.src, M.coef)


def trapz_weights(N):
    """weights of trapz on LinRange(0,1,N) (src/utils/helper.jl:560-568)."""
    t = linrange(0.0, 1.0, N)
    w = np.zeros(N)
    for k in range(N - 1):
        d = t[k + 1] - t[k]
        w[k] += 0.5 * d
        w[k + 1] += 0.5 * d
    return w


class _Formulation:
    """Shared pieces of the three algorithms' subproblems (src/solvers/scp.jl)."""

    def __init__(self, mr, N, scale, nscal):
        if hasattr(mr, "bind"):
            mr.bind(N)          # models with per-node parameters: np = np_glob + np_node N
        self.mr, self.N, self.scale = mr, N, scale
        self.S = standard_sources(mr, N, nscal)
        self.P = ConicAssembler(self.S)
        nx, nu, np_ = mr.nx, mr.nu, mr.np
        P = self.P
        self.xh = [P.var(nx, "xh") for _ in range(N)]
        self.uh = [P.var(nu, "uh") for _ in range(N)]
        self.ph = P.var(np_, "ph")

    # affine expression in PHYSICAL variables -> terms on the scaled variables (scaling.jl:134-141)
    def phys(self, Mx=None, kx=None, Mu=None, ku=None, Mp=None, const=None):
        sc = self.scale
        terms = []
        const = Aff.lift(const)
        if Mx is not None:
            Mx = Aff.lift(Mx); terms.append((self.xh[kx], Mx * sc.Sx[None, :])); const = const + Mx @ sc.cx
        if Mu is not None:
            Mu = Aff.lift(Mu); terms.append((self.uh[ku], Mu * sc.Su[None, :])); const = const + Mu @ sc.cu
        if Mp is not None and self.mr.np > 0:
            Mp = Aff.lift(Mp); terms.append((self.ph, Mp * sc.Sp[None, :])); const = const + Mp @ sc.cp
        return terms, const

    def Fp_matrix(self, k):
        """ref.dyn.F[:, :, k] as an nx x np affine matrix (only the structurally non-zero columns are sources)."""
        mr = self.mr
        F = self.S.ref("F")[:, :, k]
        full = Aff(np.zeros((mr.nx, mr.np)))
        for j, col in enumerate(mr.Fcols):
            sel = np.zeros((1, mr.np)); sel[0, col] = 1.0
            colj = F[:, j:j + 1]                                  # nx x 1
            full = full + Aff(np.zeros((mr.nx, mr.np)), colj.pos * mr.np + col, colj.src, colj.coef)
        return full

    def add_dynamics(self, relaxed=True):
        """x_{k+1} = A x_k + B- u_k + B+ u_{k+1} + F p + r + E v_k  (discretization.jl:458-467)."""
        mr, N, P, S = self.mr, self.N, self.P, self.S
        nx = mr.nx
        A, Bm, Bp, r, E = S.ref("A"), S.ref("Bm"), S.ref("Bp"), S.ref("r"), S.ref("E")
        self.vd = [P.var(nx, "vd") for _ in range(N - 1)] if relaxed else None
        for k in range(N - 1):
            t1, c1 = self.phys(Mx=np.eye(nx), kx=k + 1, const=np.zeros(nx))
            t2, c2 = self.phys(Mx=-A[:, :, k], kx=k, Mu=-Bm[:, :, k], ku=k, Mp=-self.Fp_matrix(k) if mr.np else None,
                               const=-r[:, k])
            t3, c3 = self.phys(Mu=-Bp[:, :, k], ku=k + 1, const=np.zeros(nx))
            terms = t1 + t2 + t3
            if relaxed:
                terms = terms + [(self.vd[k], -E[:, :, k])]
            P.add_zero(terms, c1 + c2 + c3)

    def add_convex_sets(self):
        """the model's X and U sets at every node (scp.jl:685-734) + its parameter-only rows (kept once)."""
        mr, N, P = self.mr, self.N, self.P
        nx = mr.nx
        for k in range(N):
            L, Lp, l, Mm, m = mr.rows(N, k + 1)
            for i in range(mr.nl):
                terms, const = self.phys(Mx=L[i:i + 1, :nx], kx=k, Mu=L[i:i + 1, nx:], ku=k,
                                         Mp=Lp[i:i + 1] if mr.np else None, const=l[i:i + 1])
                P.add_nonpos(terms, const)
            for c in range(mr.nsoc):
                rows = slice(4 * c, 4 * c + 4)
                terms, const = self.phys(Mx=Mm[rows, :nx], kx=k, Mu=Mm[rows, nx:], ku=k, const=m[rows])
                P.add_soc(terms, const)
        if mr.ng > 0:
            Lg, lg = mr.global_rows(N)
            terms, const = self.phys(Mp=Lg, const=lg)
            P.add_nonpos(terms, const)

    def add_nonconvex(self):
        """C x + D u + G p + (s - C xr - D ur - G pr) - vs <= 0  (scp.jl:770-787)."""
        mr, N, P, S = self.mr, self.N, self.P, self.S
        ns = mr.ns
        self.vs = [P.var(ns, "vs") for _ in range(N)] if ns > 0 else None
        if ns == 0:
            return
        C, D, G, rs = S.ref("C"), S.ref("D"), S.ref("Gs"), S.ref("rs")
        for k in range(N):
            Gk = scatter_param_columns(G[:, :, k], s_param_cols(mr, N, k + 1), mr.np) if mr.np else None
            terms, const = self.phys(Mx=C[:, :, k], kx=k, Mu=D[:, :, k], ku=k, Mp=Gk, const=rs[:, k])
            P.add_nonpos(terms + [(self.vs[k], -np.eye(ns))], const)

    def add_bcs(self, relaxed=True):
        """H0 x_1 + K0 p + l0 + vic = 0,  Hf x_N + Kf p + lf + vtc = 0  (scp.jl:823-893)."""
        mr, N, P, S = self.mr, self.N, self.P, self.S
        self.vic = P.var(mr.nic, "vic") if relaxed else None
        self.vtc = P.var(mr.ntc, "vtc") if relaxed else None
        kc = bc_param_cols(mr)
        K0 = scatter_param_columns(S.ref("K0"), kc, mr.np) if mr.np else None
        Kf = scatter_param_columns(S.ref("Kf"), kc, mr.np) if mr.np else None
        t, c = self.phys(Mx=S.ref("H0"), kx=0, Mp=K0, const=S.ref("l0"))
        P.add_zero(t + ([(self.vic, np.eye(mr.nic))] if relaxed else []), c)
        t, c = self.phys(Mx=S.ref("Hf"), kx=N - 1, Mp=Kf, const=S.ref("lf"))
        P.add_zero(t + ([(self.vtc, np.eye(mr.ntc))] if relaxed else []), c)

    def add_norm_cone(self, q, t_idx, var_idx, n, ref_scaled):
        """(t, xh - xh_ref) in the cone of the q-norm (ptr.jl:582-599: q2cone = {1: L1, 2: SOC, 4: SOC, Inf: LINF})."""
        P = self.P
        if n == 0:
            P.add_nonpos([(t_idx, -np.ones((1, 1)))], np.zeros(1))     # ||[]|| = 0 <= t
            return
        if q == np.inf:
            P.add_linf(t_idx, [(var_idx, np.eye(n))], -ref_scaled)
        elif q == 1:
            P.add_l1(t_idx, [(var_idx, np.eye(n))], -ref_scaled)
        elif q in (2, 4):
            P.add_soc([(t_idx, np.vstack([np.ones((1, 1)), np.zeros((n, 1))])),
                       (var_idx, np.vstack([np.zeros((1, n)), np.eye(n)]))], Aff.vstack([np.zeros((1, 1)), (-ref_scaled).reshape(n, 1)]).reshape(n + 1))
        else:
            raise ValueError("q_tr must be one of 1, 2, 4, Inf (ptr.jl:582)")

    def scaled_refs(self):
        sc, S = self.scale, self.S
        xr = (S.ref("xref") - sc.cx[:, None]) * (1.0 / sc.Sx)[:, None]
        ur = (S.ref("uref") - sc.cu[:, None]) * (1.0 / sc.Su)[:, None]
        pr = (S.ref("pref") - sc.cp) * (1.0 / sc.Sp) if self.mr.np else Aff(np.zeros(0))
        return xr, ur, pr

    def add_original_cost(self):
        """phi(x_N, p) + trapz Gamma (scp.jl:552-601) for the models' cost form (linear + diagonal quadratic)."""
        mr, N, P, sc = self.mr, self.N, self.P, self.scale
        ct = mr.cost_terms(N)
        w = trapz_weights(N)
        const = 0.0
        for k in range(N):
            P.add_cost_quad_diag(self.uh[k], w[k] * ct["Qu"] * sc.Su * sc.Su)
            P.add_cost_lin(self.uh[k], w[k] * (2 * ct["Qu"] * sc.cu * sc.Su + ct["lu"] * sc.Su))
            P.add_cost_lin(self.xh[k], w[k] * ct["lx"] * sc.Sx)
            const += w[k] * (ct["Qu"] @ (sc.cu * sc.cu) + ct["lu"] @ sc.cu + ct["lx"] @ sc.cx)
        P.add_cost_lin(self.xh[N - 1], ct["tx"] * sc.Sx)
        const += ct["tx"] @ sc.cx
        if mr.np > 0:
            P.add_cost_lin(self.ph, ct["tp"] * sc.Sp + 2 * ct["Qp"] * sc.cp * sc.Sp)
            P.add_cost_quad_diag(self.ph, ct["Qp"] * sc.Sp * sc.Sp)
            const += ct["tp"] @ sc.cp + ct["Qp"] @ (sc.cp * sc.cp)
        self.cost_const = float(const)

    def add_vc_penalty(self, weight):
        """||[E_k vd_k; vs_k]||_1 <= P_k, ||vic||_1 <= Pf_1, ||vtc||_1 <= Pf_2; cost weight (trapz(P) + sum Pf)
        (ptr.jl:799-895; scvx.jl:804-901 with weight = lambda)."""
        mr, N, P, S = self.mr, self.N, self.P, self.S
        nx, ns = mr.nx, mr.ns
        E = S.ref("E")
        w = trapz_weights(N)
        self.Pk = P.var(N, "P"); self.Pf = P.var(2, "Pf")
        for k in range(N):
            tk = self.Pk[k:k + 1]
            if ns > 0:
                if k < N - 1:
                    P.add_l1(tk, [(self.vd[k], Aff.vstack([E[:, :, k], np.zeros((ns, nx))])),
                                  (self.vs[k], np.vstack([np.zeros((nx, ns)), np.eye(ns)]))], np.zeros(nx + ns))
                else:
                    P.add_l1(tk, [(self.vs[k], np.eye(ns))], np.zeros(ns))
            else:
                if k < N - 1:
                    P.add_l1(tk, [(self.vd[k], E[:, :, k])], np.zeros(nx))
                else:
                    P.add_zero([(tk, np.ones((1, 1)))], np.zeros(1))       # P_N = 0 (ptr.jl:861-870)
        P.add_l1(self.Pf[0:1], [(self.vic, np.eye(mr.nic))], np.zeros(mr.nic))
        P.add_l1(self.Pf[1:2], [(self.vtc, np.eye(mr.ntc))], np.zeros(mr.ntc))
        P.add_cost_lin(self.Pk, weight * w); P.add_cost_lin(self.Pf, weight * np.ones(2))

    def finish(self, extra=None):
        T = self.P.finalize()
        T.cost_const = getattr(self, "cost_const", 0.0)
        T.sources = self.S
        T.N = self.N
        T.scale = self.scale
        T.mr = self.mr
        for k, v in (extra or {}).items():
            setattr(T, k, v)
        return T


# ------------------------------------------------------------------------------------------------------------------
# Template cache (round 5).  A template depends on (algorithm, options, compiled model + its constants, N, scaling) only -- not on
# the batch, the iterate or the device --, and formulating the free-flyer N = 200 GuSTO template costs seconds of host time per
# `create` (1.0e4 variables, 2.2e4 cone rows on arrays of affine scalars).  Templates are memoised in the process and, unless
# SCP_TEMPLATE_CACHE is "0" / "off", pickled under SCP_TEMPLATE_CACHE (default ~/.cache/scptoolbox_jl_amd).  The key covers every
# input of the formulation incl. the sources of this module and of affine.py and the identity of the loaded library (the model rows
# come from it), so a stale entry cannot be served.  The reference re-formulates every subproblem of every iteration in JuMP.
# ------------------------------------------------------------------------------------------------------------------
_MEMO = {}


def _cache_dir():
    d = os.environ.get("SCP_TEMPLATE_CACHE", os.path.join(os.path.expanduser("~"), ".cache", "scptoolbox_jl_amd"))
    return None if d.strip().lower() in ("", "0", "off", "no") else d


def _source_stamp():
    h = hashlib.sha256()
    here = os.path.dirname(os.path.abspath(__file__))
    for f in ("subproblem.py", "affine.py", "models.py"):     # models.py: time grid, trapezoid weights, model ids
        with open(os.path.join(here, f), "rb") as fh:
            h.update(fh.read())
    import scipy
    h.update(("numpy %s scipy %s" % (np.__version__, scipy.__version__)).encode())      # the pickles hold their sparse matrices
    try:
        st = os.stat(_lib.LIB_PATH)
        h.update(("%s:%d:%d" % (_lib.LIB_PATH, st.st_size, st.st_mtime_ns)).encode())
    except OSError:
        pass
    return h.hexdigest()


def _template_key(name, mr, N, scale, args, kwargs):
    h = hashlib.sha256()
    h.update(_source_stamp().encode())
    h.update(repr((name, type(mr).__name__, mr.name, int(N), args, sorted(kwargs.items()), bool(ConicAssembler.row_scaling), float(ConicAssembler.row_scaling_power))).encode())
    par = getattr(mr, "par", None)
    if par is not None:
        h.update(np.ascontiguousarray(par, np.float64).tobytes())
    for a in (scale.Sx, scale.cx, scale.Su, scale.cu, scale.Sp, scale.cp):
        h.update(np.ascontiguousarray(a, np.float64).tobytes())
    return h.hexdigest()


def _cached_template(fn):
    @functools.wraps(fn)
    def wrapper(mr, N, scale, *args, **kwargs):
        if not hasattr(mr, "par") or scale is None:           # test doubles of ModelRows (tests/template_util.py::OracleRows): never cached
            return fn(mr, N, scale, *args, **kwargs)
        key = _template_key(fn.__name__, mr, N, scale, args, kwargs)
        T = _MEMO.get(key)
        d = _cache_dir()
        path = None if d is None else os.path.join(d, "%s_%s_N%d_%s.pkl" % (fn.__name__, mr.name, int(N), key[:24]))
        if T is None and path is not None and os.path.exists(path):
            try:
                with open(path, "rb") as fh:
                    T = pickle.load(fh)
            except Exception:      # noqa: BLE001 -- a truncated / foreign file is rebuilt
                T = None
        if T is None:
            T = fn(mr, N, scale, *args, **kwargs)
            if path is not None:
                try:
                    os.makedirs(d, exist_ok=True)
                    mr_, T.mr = T.mr, None          # (ModelRows holds ctypes state; it is re-attached on load)
                    tmp = "%s.%d.tmp" % (path, os.getpid())
                    with open(tmp, "wb") as fh:
                        pickle.dump(T, fh, protocol=pickle.HIGHEST_PROTOCOL)
                    os.replace(tmp, path)
                    T.mr = mr_
                except Exception:      # noqa: BLE001 -- the cache is an optimisation only
                    T.mr = mr
        _MEMO[key] = T
        Tp = copy.copy(T)        # a shallow copy per problem: the memoised object is shared, its model-rows binding is the caller's
        Tp.mr = mr
        return Tp
    return wrapper


@_cached_template
def build_ptr(mr, N, scale, wvc, wtr, q_tr=np.inf):
    """`Subproblem(pbm, iter, ref)` of PTR (src/solvers/ptr.jl:213-293, 467-480): soft trust region."""
    f = _Formulation(mr, N, scale, nscal=1)
    P = f.P
    f.add_dynamics(); f.add_convex_sets(); f.add_nonconvex(); f.add_bcs()
    etax, etau, etap = P.var(N, "etax"), P.var(N, "etau"), P.var(1, "etap")
    xr, ur, pr = f.scaled_refs()
    one = np.ones((1, 1))

    def eta_link(lq, eta):
        if q_tr == 4:      # ptr.jl:601-622: (w, lq) in SOC, (w, eta, 1) in GEOM  <=>  lq^2 <= w^2 <= eta
            wv = P.var(1, "w_q4")
            P.add_soc([(wv, np.array([[1.0], [0.0]])), (lq, np.array([[0.0], [1.0]]))], np.zeros(2))
            # geometric mean of (eta, 1) >= w  <=>  (eta + 1, 2 w, eta - 1) in Q^3
            P.add_soc([(eta, np.array([[1.0], [0.0], [1.0]])), (wv, np.array([[0.0], [2.0], [0.0]]))],
                      np.array([1.0, 0.0, -1.0]))
        else:
            P.add_nonpos([(lq, one), (eta, -one)], np.zeros(1))             # lq - eta <= 0
    dp_lq = P.var(1, "dp_lq")
    f.add_norm_cone(q_tr, dp_lq, f.ph, mr.np, pr)
    eta_link(dp_lq, etap)
    dx_lq = P.var(N, "dx_lq")
    for k in range(N):
        f.add_norm_cone(q_tr, dx_lq[k:k + 1], f.xh[k], mr.nx, xr[:, k])
        eta_link(dx_lq[k:k + 1], etax[k:k + 1])
    du_lq = P.var(N, "du_lq")
    for k in range(N):
        f.add_norm_cone(q_tr, du_lq[k:k + 1], f.uh[k], mr.nu, ur[:, k])
        eta_link(du_lq[k:k + 1], etau[k:k + 1])
    f.add_original_cost()
    w = trapz_weights(N)
    P.add_cost_lin(etax, wtr * w); P.add_cost_lin(etau, wtr * w); P.add_cost_lin(etap, wtr)   # ptr.jl:783-786
    f.add_vc_penalty(wvc)
    return f.finish(dict(algo="ptr", wvc=wvc, wtr=wtr, q_tr=q_tr))


@_cached_template
def build_scvx(mr, N, scale, lam, q_tr=np.inf):
    """`Subproblem(pbm, iter, eta, ref)` of SCvx (src/solvers/scvx.jl:225-303): hard trust region
    dx_lq[k] + du_lq[k] + dp_lq <= eta (:663-675), cost L + lambda (trapz(P) + sum Pf) (:895-898).
    Source scal[0] = eta."""
    f = _Formulation(mr, N, scale, nscal=1)
    P = f.P
    f.add_dynamics(); f.add_convex_sets(); f.add_nonconvex(); f.add_bcs()
    xr, ur, pr = f.scaled_refs()
    dp_lq = P.var(1, "dp_lq")
    f.add_norm_cone(q_tr, dp_lq, f.ph, mr.np, pr)
    dx_lq = P.var(N, "dx_lq")
    for k in range(N):
        f.add_norm_cone(q_tr, dx_lq[k:k + 1], f.xh[k], mr.nx, xr[:, k])
    du_lq = P.var(N, "du_lq")
    for k in range(N):
        f.add_norm_cone(q_tr, du_lq[k:k + 1], f.uh[k], mr.nu, ur[:, k])
    eta = f.S.ref("scal")[0:1]
    one = np.ones((1, 1))
    for k in range(N):
        if q_tr == 4:      # scvx.jl:648-662: (w, dx_lq, du_lq, dp_lq) in SOC, (w, eta, 1) in GEOM  <=>  dx^2 + du^2 + dp^2 <= eta
            wv = P.var(1, "w_q4")
            e = lambda i: np.eye(4)[:, i:i + 1]
            P.add_soc([(wv, e(0)), (dx_lq[k:k + 1], e(1)), (du_lq[k:k + 1], e(2)), (dp_lq, e(3))], np.zeros(4))
            P.add_soc([(wv, np.array([[0.0], [2.0], [0.0]]))], geom2_const(eta))
        else:
            P.add_nonpos([(dx_lq[k:k + 1], one), (du_lq[k:k + 1], one), (dp_lq, one)], -eta)
    f.add_original_cost()
    f.add_vc_penalty(lam)
    return f.finish(dict(algo="scvx", lam=lam, q_tr=q_tr))


def geom2_const(eta):
    """constant part of the Q^3 form of `(w, eta, 1) in GEOM` (src/parser/cone.jl:36-47: geomean((eta, 1)) >= w), the form a
    second-order-cone solver receives from the geometric-mean bridge:  w^2 <= eta  <=>  (eta + 1, 2 w, eta - 1) in Q^3.
    eta: an affine scalar (a source such as the trust-region radius) of shape (1,)."""
    e = Aff.lift(eta).reshape(1, 1)
    return Aff.vstack([e + 1.0, np.zeros((1, 1)), e - 1.0]).reshape(3)


def split_state_rows(mr, N, k):
    """(X rows, U rows) of the model's linear rows at node k: a row with an input column belongs to U (hard in every
    algorithm, problem.jl:534-542), the others to X (soft-penalised by GuSTO, gusto.jl:883-934)."""
    L, Lp, l, Mm, m = mr.rows(N, k)
    nx = mr.nx
    xrows = [i for i in range(mr.nl) if not np.any(L[i, nx:] != 0.0) and (np.any(L[i] != 0.0) or np.any(Lp[i] != 0.0))]
    urows = [i for i in range(mr.nl) if np.any(L[i, nx:] != 0.0)]
    return L, Lp, l, Mm, m, xrows, urows


def state_cones(mr, Mm):
    """indices of the second-order cones of the model that constrain the state only (members of X)"""
    return [c for c in range(mr.nsoc) if not np.any(Mm[4 * c:4 * c + 4, mr.nx:] != 0.0)]


@_cached_template
def build_gusto(mr, N, scale, q_tr=np.inf, literal_slack=False, pen="quad", hom=500.0):
    """`Subproblem(pbm, iter, lambda, eta, ref)` of GuSTO (src/solvers/gusto.jl:218-287, 534-550) with `pen = "quad"` (described
    first) or `pen = "softplus"` (:996-1031, `soft()` below: two exponential cones per penalised quantity) and q_tr in {1, 2, 4, Inf}: un-relaxed dynamics and boundary conditions (:452-454), U hard, the convex state rows and the linearised
    non-convex rows soft (soft_penalty :936-995: u >= 0, f + u - v <= 0, cost lambda v^2, summed with trapz :798-831),
    soft trust region dx_lq[k] + dp_lq <= eta + tr[k] with tr penalised the same way (:1056-1170).
    Sources scal = [eta, lambda]; lambda enters the quadratic cost (the P values are per problem).

    literal_slack: the reference writes a soft penalty as `u >= 0, f + u - v <= 0, cost lambda v^2` (gusto.jl:972-995).  The
    slack u is REDUNDANT -- v = max(f, 0) at the optimum either way, and for an inactive constraint every u in [0, -f] is optimal:
    a flat direction that makes the interior-point iterations degenerate (measured on Monte-Carlo free-flyer instances: 4 of 5
    first subproblems stop at reduced accuracy after hundreds of dynamic regularisations with it, 2 of 5 without, same optimal
    values).  The product formulates the equivalent `f - v <= 0, cost lambda v^2` (same x, u, p, v); literal_slack=True
    reproduces the reference's variable set (the oracle's literal program, tests/test_template_cpu.py)."""
    if not getattr(mr, "gusto_ok", False):
        raise NotImplementedError("GuSTO needs s(t, k, x, p) independent of the input (gusto.jl:757-792); model '%s' does not "
                                  "qualify (scp_model_info.s_input_free)" % mr.name)
    f = _Formulation(mr, N, scale, nscal=2)
    P, S = f.P, f.S
    w = trapz_weights(N)
    eta, lam = S.ref("scal")[0:1], S.ref("scal")[1:2]
    nx = mr.nx
    f.add_dynamics(relaxed=False)
    one = np.ones((1, 1))

    def soft(terms, const, k, name):
        if pen == "softplus":
            # gusto.jl:996-1031: (-w, 1, u) in EXP, (hom f - w, 1, v) in EXP, u + v <= 1, cost lambda w / hom, i.e.
            # exp(-w) + exp(hom f - w) <= 1  <=>  w >= log(1 + exp(hom f)); the penalty variable of the template is w
            ww = P.var(1, name)
            uu, vv = P.var(1, name + "_eu"), P.var(1, name + "_ev")
            e0, e2 = np.array([[1.0], [0.0], [0.0]]), np.array([[0.0], [0.0], [1.0]])
            P.add_exp([(ww, -e0), (uu, e2)], np.array([0.0, 1.0, 0.0]))
            c3 = Aff.vstack([Aff.lift(const).reshape(1, 1) * hom, np.ones((1, 1)), np.zeros((1, 1))]).reshape(3)
            P.add_exp([(idx, Aff.vstack([Aff.lift(M).reshape(1, -1) * hom, np.zeros((2, Aff.lift(M).c0.size))])) for idx, M in terms] +
                      [(ww, -e0), (vv, e2)], c3)
            P.add_nonpos([(uu, one), (vv, one)], np.array([-1.0]))
            P.add_cost_lin(ww, lam * (w[k] / hom))
            if name == "v_st":
                st_nodes[k].append(int(ww[0]))
            return ww
        vv = P.var(1, name)
        if literal_slack:
            uu = P.var(1, name + "_u")
            P.add_nonpos([(uu, -one)], np.zeros(1))
            P.add_nonpos(list(terms) + [(uu, one), (vv, -one)], const)
        else:
            P.add_nonpos(list(terms) + [(vv, -one)], const)
        P.add_cost_quad_diag(vv, lam * w[k])
        if name == "v_st":
            st_nodes[k].append(int(vv[0]))
        return vv
    st_nodes = [[] for _ in range(N)]

    def indicator(rows_terms_consts, k, soc=False):
        """cone indicator q of define_conic_constraint! (src/parser/problem.jl:705-781) + its soft penalty (gusto.jl:883-934):
        NONPOS rows (a LINF cone lowered to rows shares ONE q): row - q <= 0; SOC: [z0 + q; z1..] in Q."""
        qv = P.var(1, "q_ind")
        if soc:
            terms, const = rows_terms_consts
            e0 = np.zeros((4, 1)); e0[0, 0] = 1.0
            P.add_soc(list(terms) + [(qv, e0)], const)
        else:
            for terms, const in rows_terms_consts:
                P.add_nonpos(list(terms) + [(qv, -one)], const)
        soft([(qv, one)], np.zeros(1), k, "v_st")
    # convex sets: U hard, X soft through its cone indicators.  Parameter-only rows: hard once when they belong to U (the
    # reference's quadrotor definition), soft at every node when they belong to X (mr.global_rows_in_X: free-flyer)
    glob_in_X = bool(getattr(mr, "global_rows_in_X", False))
    for k in range(N):
        L, Lp, l, Mm, m, xrows, urows = split_state_rows(mr, N, k + 1)
        xcones = state_cones(mr, Mm)
        for i in urows:
            terms, const = f.phys(Mx=L[i:i + 1, :nx], kx=k, Mu=L[i:i + 1, nx:], ku=k, Mp=Lp[i:i + 1] if mr.np else None,
                                  const=l[i:i + 1])
            P.add_nonpos(terms, const)
        for c in range(mr.nsoc):
            rows = slice(4 * c, 4 * c + 4)
            if c in xcones:
                indicator(f.phys(Mx=Mm[rows, :nx], kx=k, const=m[rows]), k, soc=True)
            else:
                terms, const = f.phys(Mx=Mm[rows, :nx], kx=k, Mu=Mm[rows, nx:], ku=k, const=m[rows])
                P.add_soc(terms, const)
        groups = getattr(mr, "linf_groups", None)
        groups = [[i] for i in xrows] if groups is None else [g for g in groups(N, k + 1) if set(g) <= set(xrows)] + \
            [[i] for i in xrows if not any(i in g for g in groups(N, k + 1))]
        if glob_in_X and mr.ng > 0:
            Lg, lg = mr.global_rows(N)
            for i in range(mr.ng):
                indicator([f.phys(Mp=Lg[i:i + 1], const=lg[i:i + 1])], k)
        for g in groups:
            indicator([f.phys(Mx=L[i:i + 1, :nx], kx=k, Mp=Lp[i:i + 1] if mr.np else None, const=l[i:i + 1]) for i in g], k)
    if mr.ng > 0 and not glob_in_X:
        Lg, lg = mr.global_rows(N)
        terms, const = f.phys(Mp=Lg, const=lg)
        P.add_nonpos(terms, const)
    # linearised non-convex constraints, soft (s must not depend on u in GuSTO: D is not a source here)
    if mr.ns > 0:
        C, G, rs = S.ref("C"), S.ref("Gs"), S.ref("rs")
        for k in range(N):
            for i in range(mr.ns):
                Gi = scatter_param_columns(G[i:i + 1, :, k], s_param_cols(mr, N, k + 1), mr.np) if mr.np else None
                terms, const = f.phys(Mx=C[i:i + 1, :, k], kx=k, Mp=Gi, const=rs[i:i + 1, k])
                soft(terms, const, k, "v_st")
    f.add_bcs(relaxed=False)
    # soft trust region
    xr, ur, pr = f.scaled_refs()
    tr = P.var(N, "tr"); dx_lq = P.var(N, "dx_lq"); dp_lq = P.var(1, "dp_lq")
    f.add_norm_cone(q_tr, dp_lq, f.ph, mr.np, pr)
    for k in range(N):
        f.add_norm_cone(q_tr, dx_lq[k:k + 1], f.xh[k], mr.nx, xr[:, k])
        if q_tr == 4:      # gusto.jl:1107-1131: (w, dx_lq, dp_lq) in SOC, (w, eta + tr, 1) in GEOM  <=>  dx^2 + dp^2 <= eta + tr
            wv = P.var(1, "w_q4")
            e = lambda i: np.eye(3)[:, i:i + 1]
            P.add_soc([(wv, e(0)), (dx_lq[k:k + 1], e(1)), (dp_lq, e(2))], np.zeros(3))
            P.add_soc([(tr[k:k + 1], np.array([[1.0], [0.0], [1.0]])), (wv, np.array([[0.0], [2.0], [0.0]]))], geom2_const(eta))
        else:
            P.add_nonpos([(dx_lq[k:k + 1], one), (dp_lq, one), (tr[k:k + 1], -one)], -eta)
        soft([(tr[k:k + 1], one)], np.zeros(1), k, "v_tr")
    f.add_original_cost()
    nst = len(st_nodes[0])
    assert all(len(r) == nst for r in st_nodes)
    return f.finish(dict(algo="gusto", q_tr=q_tr, nst=nst, v_st_nodes=np.array(st_nodes, np.int64).reshape(N, nst), pen=pen, hom=hom))


@_cached_template
def build_correct_convex(mr, N, scale):
    """`correct_convex!` (src/solvers/scp.jl:275-361): L1 projection of a guess onto the convex path constraints."""
    f = _Formulation(mr, N, scale, nscal=1)
    P = f.P
    f.add_convex_sets()
    epi_x, epi_u, epi_p = P.var(N, "epi_x"), P.var(N, "epi_u"), P.var(1, "epi_p")
    xr, ur, pr = f.scaled_refs()
    for k in range(N):
        P.add_l1(epi_x[k:k + 1], [(f.xh[k], np.eye(mr.nx))], -xr[:, k])
        P.add_l1(epi_u[k:k + 1], [(f.uh[k], np.eye(mr.nu))], -ur[:, k])
    if mr.np > 16:
        # A LONG parameter vector (free-flyer: np = 1 + 6 N).  The literal L1 cone (MOI's NormOne bridge: |dp_i| <= y_i, sum y <= epi_p,
        # cost epi_p) ends in ONE row over all np auxiliaries; the solver eliminates cone rows first, so that row makes P + Gt'Gt dense
        # on them: 1 201 x 1 201 at N = 200, 2.9e8 multiply-adds per factorisation against 3.7e6 for the whole GuSTO subproblem of the same
        # grid -- the projection took 17.8 s of a launch that otherwise lasts 1.7 s (round-5 bench line).  epi_p appears in that row and in
        # the cost only, so at every optimum epi_p = sum y: the cost sum_i y_i WITHOUT the row is the same program in (x, u, p, y) -- same
        # minimisers, same optimal value -- and stays sparse.  (ECOS keeps the literal row sparse because its KKT matrix carries z explicitly.)
        y = P.var(mr.np, "abs_p")
        I = np.eye(mr.np)
        P.add_nonpos([(f.ph, I), (y, -I)], -pr)
        P.add_nonpos([(f.ph, -I), (y, -I)], pr)
        P.add_nonpos([(epi_p, -np.ones((1, 1)))], np.zeros(1))       # (the variable stays in the layout, pinned at 0 by its cost)
        P.add_cost_lin(y, np.ones(mr.np))
    elif mr.np > 0:
        P.add_l1(epi_p, [(f.ph, np.eye(mr.np))], -pr)
    else:
        P.add_nonpos([(epi_p, -np.ones((1, 1)))], np.zeros(1))
    P.add_cost_lin(epi_x, np.ones(N)); P.add_cost_lin(epi_u, np.ones(N)); P.add_cost_lin(epi_p, np.ones(1))
    return f.finish(dict(algo="correct_convex"))
