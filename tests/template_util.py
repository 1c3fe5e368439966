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
    put("C", C); put("D", D); put("Gs", G); put("rs", rs)
    for tag, xb, g, H, K in (("0", ref.xd[0], mdl.gic, mdl.H0, mdl.K0), ("f", ref.xd[-1], mdl.gtc, mdl.Hf, mdl.Kf)):
        gv, Hv, Kv = g(xb, ref.p, pp), H(xb, ref.p, pp), K(xb, ref.p, pp)
        put("H" + tag, Hv); put("K" + tag, np.asarray(Kv).reshape(len(gv), mdl.np))
        put("l" + tag, gv - Hv @ xb - (Kv @ ref.p if mdl.np else 0))
    put("scal", np.atleast_1d(scal))
    return src


def template_matrices(T, src):
    v = T.values(src)
    G = sp.csc_matrix((v["Gx"], T.G.indices, T.G.indptr), shape=T.G.shape)
    A = sp.csc_matrix((v["Ax"], T.A.indices, T.A.indptr), shape=T.A.shape)
    P = sp.csc_matrix((v["Px"], T.P.indices, T.P.indptr), shape=T.P.shape)
    return v, G, A, P
