"""ctypes front-end of the CPU BASELINE (oracle/cpu_ptr.cpp: C++/OpenMP restatement of the batched PTR iteration the HIP
library runs).  Test / bench infrastructure only -- the product path never imports this."""
import ctypes
import os
import subprocess

import numpy as np

from .models import MODELS
from .oracle import MODEL_IDS
from .ptr_ref import Scaling

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libscp_cpu.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = ctypes.CDLL(_SO)
        _lib.cpu_ptr_solve_batch.restype = ctypes.c_int
        _lib.cpu_ptr_max_threads.restype = ctypes.c_int
    return _lib


def max_threads():
    return int(lib().cpu_ptr_max_threads())


def effective_cpus():
    """CPUs this process may really use: min(affinity mask, cgroup v2/v1 CPU quota) -- a container often SEES every core of
    the host (os.cpu_count()) while its quota is a few CPUs; oversubscribing OpenMP threads there is pathological."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def solve_batch(model, N, Nsub, iters, pp, wvc=1e3, wtr=0.1, feas_tol=1e-3, threads=0, guess=None, want_hist=False, deadline_s=0.0):
    """Batched PTR (fixed iteration count) on the host.  pp[B, npp].  Returns dict(xd[B,N,nx], ud, p, stats[B,8], seconds, hist)."""
    L = lib()
    mdl = MODELS[model]()
    pp = np.ascontiguousarray(np.atleast_2d(pp), dtype=np.float64)
    B = pp.shape[0]
    sc = Scaling(*mdl.bbox())
    if guess is None:
        g = [mdl.guess(N, pp[b]) for b in range(B)]
        xd = np.ascontiguousarray(np.stack([a[0] for a in g])); ud = np.ascontiguousarray(np.stack([a[1] for a in g]))
        p = np.ascontiguousarray(np.stack([a[2] for a in g]).reshape(B, -1))
    else:
        xd, ud, p = [np.ascontiguousarray(a, dtype=np.float64).copy() for a in guess]
    par = np.ascontiguousarray(mdl.par(), dtype=np.float64)
    stats = np.zeros((B, 8)); hist = np.zeros((B, iters, 6)) if want_hist else None
    sec = ctypes.c_double(0.0); ndone = ctypes.c_int(0)
    dp = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None and a.size else None
    c = [np.ascontiguousarray(v, dtype=np.float64) for v in (sc.Sx, sc.cx, sc.Su, sc.cu, sc.Sp, sc.cp)]
    rc = L.cpu_ptr_solve_batch(ctypes.c_int(MODEL_IDS[model]), dp(par), ctypes.c_int(N), ctypes.c_int(Nsub), ctypes.c_int(iters),
                               ctypes.c_double(wvc), ctypes.c_double(wtr), ctypes.c_double(feas_tol), dp(c[0]), dp(c[1]), dp(c[2]),
                               dp(c[3]), dp(c[4]), dp(c[5]), ctypes.c_int(B), dp(pp), dp(xd), dp(ud), dp(p), ctypes.c_int(threads),
                               dp(stats), dp(hist), ctypes.byref(sec), ctypes.c_double(deadline_s), ctypes.byref(ndone))
    if rc != 0:
        raise RuntimeError("cpu_ptr_solve_batch rc=%d" % rc)
    return dict(xd=xd, ud=ud, p=p, stats=stats, seconds=sec.value, hist=hist, scale=sc, n_done=ndone.value)
