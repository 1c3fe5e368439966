"""ctypes front-end of the CPU ORACLE (test infrastructure, not product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It wraps oracle/_build/libscp_oracle.so (built from
oracle/scp_oracle.c by oracle/Makefile), the plain-C restatement of the
reference's `discretize!` (src/solvers/discretization.jl:160-406).

Parity status: **parity unpinned** (the reference ships no golden vectors and
cannot be run here -- SURVEY.md F4/F5); see tests/test_oracle_discretize.py for
the mathematical pins (LTI closed form, Jacobian finite differences, the
reference's independent FOH discretiser).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libscp_oracle.so")

MODEL_IDS = {"double_integrator": 0, "quadrotor": 1, "rocket_landing": 2, "starship": 3, "freeflyer": 4}
MODEL_DIMS = {"double_integrator": (2, 1, 0), "quadrotor": (6, 4, 1), "rocket_landing": (7, 4, 1), "starship": (8, 3, 10),
              "freeflyer": (13, 6, 1)}

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "scp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_discretize_batch.restype = ctypes.c_int
        _lib.oracle_discretize.restype = ctypes.c_int
        _lib.oracle_discretize_impulse.restype = ctypes.c_int
        _lib.oracle_model_eval.restype = ctypes.c_int
        _lib.oracle_propagate.restype = ctypes.c_int
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(_dp)


def default_params(model):
    """Nominal model parameter blobs (doubles), from the reference's examples."""
    if model == "double_integrator":
        # test/examples/double_integrator/parameters.jl:58-61 (choice 1), T = 10
        return np.array([0.1, 10.0])
    if model == "quadrotor":
        # test/examples/quadrotor/parameters.jl:109 (g = 9.81)
        return np.array([9.81])
    if model == "freeflyer":
        return np.array([7.2, 0.1083, 0.1083, 0.1083])      # [m, J1, J2, J3], test/examples/freeflyer/parameters.jl:140-141
    if model == "starship":
        return np.array([31.0, 100.0])      # [N, hs]: only s(.) / the cost use them (phase-switch node, starship_flip/definition.jl:705-712)
    if model == "rocket_landing":
        # test/examples/rocket_landing/parameters.jl:78-106
        g = np.array([0.0, 0.0, -3.7114])
        th = 30 * np.pi / 180
        T_sid = 24.6229 * 3600
        w = (2 * np.pi / T_sid) * np.array([np.cos(th), 0.0, np.sin(th)])
        Isp, phi, ge = 225.0, 27 * np.pi / 180, 9.807
        alpha = 1 / (Isp * ge * np.cos(phi))
        return np.concatenate([g, w, [alpha]])
    raise KeyError(model)


def discretize(model, par, N, Nsub, xd, ud, p, iSx_diag, feas_tol, method="foh"):
    """Batched `discretize!` (method "foh" or "impulse": discretization.jl:184-193).

    xd[B,N,nx], ud[B,N,nu], p[B,np] in C order (== Julia [nx,N,B] column-major).
    Returns dict of A[B,N-1,nx,nx]^T-layout arrays **in Julia memory order**:
    out["A"][b,k] is the column-major nx*nx block, exposed as shape
    (B, N-1, ncols, nx) so that out["A"][b,k].T is the math matrix.
    """
    nx, nu, np_ = MODEL_DIMS[model]
    xd, ud, p = _c(xd), _c(ud), _c(p)
    B = xd.shape[0]
    assert xd.shape == (B, N, nx) and ud.shape == (B, N, nu) and p.shape == (B, np_)
    M = N - 1
    out = dict(
        A=np.zeros((B, M, nx, nx)), Bm=np.zeros((B, M, nu, nx)), Bp=np.zeros((B, M, nu, nx)),
        F=np.zeros((B, M, np_, nx)), r=np.zeros((B, M, nx)), E=np.zeros((B, M, nx, nx)),
        defect=np.zeros((B, M, nx)),
    )
    feas = np.zeros(B, dtype=np.int32)
    par = _c(par)
    iSx = _c(iSx_diag)
    if method == "impulse":
        M_ = N - 1
        for b in range(B):
            fb = np.zeros(1, dtype=np.int32)
            sl = lambda a: a[b:b + 1]
            rc = lib().oracle_discretize_impulse(
                ctypes.c_int(MODEL_IDS[model]), _ptr(par), ctypes.c_int(N), ctypes.c_int(Nsub),
                _ptr(_c(xd[b])), _ptr(_c(ud[b])), _ptr(_c(p[b])), _ptr(iSx), ctypes.c_double(feas_tol),
                *[ctypes.cast(out[k][b].ctypes.data, _dp) for k in ("A", "Bm", "Bp", "F", "r", "E", "defect")],
                fb.ctypes.data_as(_ip))
            if rc:
                raise RuntimeError("oracle_discretize_impulse rc=%d" % rc)
            feas[b] = fb[0]
        out["feas"] = feas.astype(bool)
        return out
    rc = lib().oracle_discretize_batch(
        ctypes.c_int(MODEL_IDS[model]), _ptr(par), ctypes.c_int(N), ctypes.c_int(Nsub), ctypes.c_int(B),
        _ptr(xd), _ptr(ud), _ptr(p), _ptr(iSx), ctypes.c_double(feas_tol),
        _ptr(out["A"]), _ptr(out["Bm"]), _ptr(out["Bp"]), _ptr(out["F"]), _ptr(out["r"]), _ptr(out["E"]),
        _ptr(out["defect"]), feas.ctypes.data_as(_ip))
    if rc:
        raise RuntimeError("oracle_discretize_batch rc=%d" % rc)
    out["feas"] = feas.astype(bool)
    return out


def model_eval(model, par, t, k, x, u, p):
    """f, A, B, F of the oracle's model (math-layout matrices)."""
    nx, nu, np_ = MODEL_DIMS[model]
    f = np.zeros(nx)
    A = np.zeros((nx, nx))
    Bm = np.zeros((nu, nx))
    F = np.zeros((max(np_, 1), nx))
    x, u, p, par = _c(x), _c(u), _c(p), _c(par)
    rc = lib().oracle_model_eval(ctypes.c_int(MODEL_IDS[model]), _ptr(par), ctypes.c_double(t), ctypes.c_int(k),
                                 _ptr(x), _ptr(u), _ptr(p), _ptr(f), _ptr(A), _ptr(Bm), _ptr(F))
    if rc:
        raise RuntimeError("oracle_model_eval rc=%d" % rc)
    return f, A.T.copy(), Bm.T.copy(), F[:np_].T.copy()


def propagate(model, par, N, xd, ud, p, res=1000):
    """`propagate(sol, pbm; res)` (FOH) for ONE problem: xd[N,nx], ud[N,nu], p[np] -> (tc[res], xc[res,nx])."""
    nx, nu, np_ = MODEL_DIMS[model]
    xd, ud, p, par = _c(xd), _c(ud), _c(p), _c(par)
    xc = np.zeros((res, nx))
    rc = lib().oracle_propagate(ctypes.c_int(MODEL_IDS[model]), _ptr(par), ctypes.c_int(N), _ptr(xd), _ptr(ud), _ptr(p),
                                ctypes.c_int(res), _ptr(xc))
    if rc:
        raise RuntimeError("oracle_propagate rc=%d" % rc)
    tc = np.array([(1 - j / (res - 1)) * 0.0 + (j / (res - 1)) * 1.0 for j in range(res)])
    return tc, xc


def propagate_impulse(model, par, N, xd, ud, p, res=1000):
    """`propagate(sol, pbm; res)` for the IMPULSE method, src/solvers/discretization.jl:542-560, for ONE problem (numpy
    restatement over the C model evaluation): returns (tc, xc[len(tc), nx]).  Every interval restarts from
    xd[k] + f(t_k, -k, xd[k], ud[k], p) and coasts with idle inputs over LinRange(t_k, t_{k+1}, ceil(res / (N - 1)))."""
    nx, nu, _ = MODEL_DIMS[model]
    td = np.array([(1 - j / (N - 1)) * 0.0 + (j / (N - 1)) * 1.0 for j in range(N)])
    sub = -(-int(res) // (N - 1))
    tcs, xcs = [np.array([0.0])], [np.asarray(xd[0], float)[None, :]]
    for k in range(N - 1):
        tg = np.array([(1 - j / (sub - 1)) * td[k] + (j / (sub - 1)) * td[k + 1] for j in range(sub)])
        x = np.asarray(xd[k], float) + model_eval(model, par, td[k], -(k + 1), xd[k], ud[k], p)[0]
        f = lambda t, xx: model_eval(model, par, t, N, xx, np.zeros(nu), p)[0]
        rows = [x.copy()]
        for j in range(1, sub):
            h = tg[j] - tg[j - 1]
            k1 = f(tg[j - 1], x); k2 = f(tg[j - 1] + h / 2, x + h / 2 * k1); k3 = f(tg[j - 1] + h / 2, x + h / 2 * k2)
            k4 = f(tg[j - 1] + h, x + h * k3)
            x = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
            rows.append(x.copy())
        tg = tg.copy(); tg[0] += np.sqrt(np.finfo(float).eps)
        tcs.append(tg); xcs.append(np.array(rows))
    return np.concatenate(tcs), np.vstack(xcs)
