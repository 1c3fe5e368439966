"""Shared by the template tests: the per-problem SOURCE VECTOR of a conic template (scptoolbox.jl_amd/subproblem.py)
built from ORACLE data (oracle discretize! + the oracle's model Jacobians), i.e. what csrc/scp_generic.hpp fills on the
device -- so that a template can be checked against the oracle's literal conic program on the CPU."""
import numpy as np
import scipy.sparse as sp

from oracle.models import linrange


def make_src(T, mdl, ref, pp, scal=0.0, Fcols=None):
    S, N = T.sources, T.N
    src = np.zeros(S.n)

    def put(name, arr):
        off, shape = S.segs[name]
        a = np.asarray(arr, float).reshape(shape)
        src[off:off + a.size] = a.reshape(-1, order="F")
    t = linrange(0, 1, N)
    put("xref", ref.xd.T); put("uref", ref.ud.T); put("pref", ref.p)
    tr = lambda a: np.transpose(a, (1, 2, 0))
    npF = S.segs["F"][1][1]
    Fcols = list(range(npF)) if Fcols is None else list(Fcols)    # structurally non-zero columns of F
    put("A", tr(ref.A)); put("Bm", tr(ref.Bm)); put("Bp", tr(ref.Bp)); put("F", tr(ref.F[:, :, Fcols])); put("r", ref.r.T)
    put("E", tr(ref.E))
    ns = mdl.ns
    C = np.zeros((ns, mdl.nx, N)); D = np.zeros((ns, mdl.nu, N)); G = np.zeros((ns, mdl.np, N)); rs = np.zeros((ns, N))
    for k in range(N):
        if ns == 0:
            break
        a = (t[k], k + 1, ref.xd[k], ref.ud[k], ref.p)
        s, Ck, Dk, Gk = mdl.s(*a), mdl.C(*a), mdl.D(*a), mdl.G(*a)
        C[:, :, k] = Ck; D[:, :, k] = Dk
        if mdl.np:
            G[:, :, k] = np.asarray(Gk).reshape(ns, mdl.np)
        rs[:, k] = s - Ck @ ref.xd[k] - Dk @ ref.ud[k] - (Gk @ ref.p if mdl.np else 0)
    from scptoolbox_jl_amd.subproblem import bc_param_cols, s_param_cols
    mr = getattr(T, "mr", None)
    if mr is not None and S.segs["Gs"][1][1] != mdl.np:          # compact parameter columns (free-flyer)
        Gc = np.zeros((ns, S.segs["Gs"][1][1], N))
        for k in range(N):
            cols = s_param_cols(mr, N, k + 1)
            assert not np.delete(G[:, :, k], cols, axis=1).any()  # the declared columns are the only non-zero ones
            Gc[:, :, k] = G[:, cols, k]
        G = Gc
    put("C", C); put("D", D); put("Gs", G); put("rs", rs)
    kc = bc_param_cols(mr) if mr is not None else np.arange(mdl.np)
    for tag, xb, g, H, K in (("0", ref.xd[0], mdl.gic, mdl.H0, mdl.K0), ("f", ref.xd[-1], mdl.gtc, mdl.Hf, mdl.Kf)):
        gv, Hv, Kv = g(xb, ref.p, pp), H(xb, ref.p, pp), K(xb, ref.p, pp)
        Kv = np.asarray(Kv).reshape(len(gv), mdl.np)
        assert not np.delete(Kv, kc, axis=1).any()
        put("H" + tag, Hv); put("K" + tag, Kv[:, kc])
        put("l" + tag, gv - Hv @ xb - (Kv @ ref.p if mdl.np else 0))
    put("scal", np.atleast_1d(scal))
    return src


def template_matrices(T, src):
    v = T.values(src)
    G = sp.csc_matrix((v["Gx"], T.G.indices, T.G.indptr), shape=T.G.shape)
    A = sp.csc_matrix((v["Ax"], T.A.indices, T.A.indptr), shape=T.A.shape)
    P = sp.csc_matrix((v["Px"], T.P.indices, T.P.indptr), shape=T.P.shape)
    return v, G, A, P


class OracleRows:
    """The `ModelRows` interface of scptoolbox.jl_amd/subproblem.py served from an ORACLE model instead of the compiled one
    (scp_model_rows): lets the host-side template builder be checked on problems whose device model does not exist yet --
    the free-flyer with its N-dependent parameter vector (np = 1 + 6 N, LINF room cones lowered to NONPOS rows with a
    per-node parameter column).  Rows are cached per node."""

    def __init__(self, mdl, N):
        self.mdl, self.N, self.name = mdl, N, mdl.name
        self.nx, self.nu, self.np = mdl.nx, mdl.nu, mdl.np
        npd = getattr(mdl, "np_dyn", mdl.np)
        self.npF, self.Fcols = npd, list(range(npd))
        self.ns, self.nic, self.ntc = mdl.ns, mdl.nic, mdl.ntc
        self._cache = {}
        L, Lp, l, Mm, m, Lg, lg = self._node(1)
        self.nl, self.nsoc, self.ng = L.shape[0], Mm.shape[0] // 4, Lg.shape[0]

    def _node(self, k):
        if k in self._cache:
            return self._cache[k]
        from oracle.ptr_ref import lower_linf
        mdl, nx, nu, np_ = self.mdl, self.nx, self.nu, self.np
        t = linrange(0, 1, self.N)[k - 1]
        nz = nx + nu
        L, Lp, l, Mm, m, Lg, lg = [], [], [], [], [], [], []
        groups = []
        for is_x, rows in ((True, mdl.X(t, k)), (False, mdl.U(t, k))):
            for kind, M, Mpar, m0 in rows:
                was_linf = kind == "LINF"
                kind, M, Mpar, m0 = lower_linf(kind, M, Mpar, m0)
                if was_linf:         # the rows of one LINF cone share one cone indicator under GuSTO
                    groups.append(list(range(len(L), len(L) + M.shape[0])))
                Mz = np.zeros((M.shape[0], nz))
                Mz[:, :nx] = M if is_x else 0.0
                if not is_x:
                    Mz[:, nx:] = M
                if kind == "NONPOS":
                    for i in range(M.shape[0]):
                        if not Mz[i].any():          # parameter-only row: global (kept once)
                            Lg.append(Mpar[i]); lg.append(m0[i])
                        else:
                            L.append(Mz[i]); Lp.append(Mpar[i]); l.append(m0[i])
                else:
                    assert kind == "SOC" and M.shape[0] == 4 and not Mpar.any()
                    Mm.append(Mz); m.append(m0)
        out = (np.array(L).reshape(-1, nz), np.array(Lp).reshape(-1, np_), np.array(l), np.vstack(Mm) if Mm else np.zeros((0, nz)),
               np.concatenate(m) if m else np.zeros(0), np.array(Lg).reshape(-1, np_), np.array(lg))
        self._cache[k] = out
        self._groups = getattr(self, "_groups", {})
        self._groups[k] = groups
        return out

    global_rows_in_X = True      # the free-flyer's t_f bounds are members of X (definition.jl:318-331): soft under GuSTO
    gusto_ok = True

    def linf_groups(self, N, k):
        self._node(k)
        return self._groups[k]

    sparse_params = True

    def s_param_cols(self, N, k):
        """free-flyer: s at node k sees its own six room slacks only (definition.jl:412-428)"""
        if self.sparse_params and hasattr(self.mdl, "id_delta"):
            return self.mdl.id_delta(k)
        return np.arange(self.np)

    def bc_param_cols(self):
        return np.zeros(0, np.int64) if self.sparse_params and hasattr(self.mdl, "id_delta") else np.arange(self.np)

    def rows(self, N, k):
        return self._node(k)[:5]

    def global_rows(self, N):
        return self._node(1)[5:]

    def cost_terms(self, N):
        return self.mdl.cost_terms()
