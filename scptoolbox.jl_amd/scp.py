"""Host-side mirror of the shared SCP scaffolding (src/solvers/scp.jl) for the
hot path only: SCPScaling, SCPProblem (owning the native handle), the batched
reference/solution container and `discretize_` (== `discretize!`).

Array convention: numpy arrays are C-ordered with the batch index FIRST and the
Julia dimensions reversed, e.g. xd[B, N, nx] has exactly the memory layout of a
Julia Array{Float64,3} of size (nx, N, B).  Matrices therefore appear
transposed: dyn.A[b, k] is the column-major nx-by-nx block, `dyn.A[b, k].T` is
the math matrix A_k.
"""
import ctypes
import math

import numpy as np

from . import _lib
from .models import MODEL_IDS, NativeModel, linrange

FOH, IMPULSE = 0, 1  # DiscretizationType, src/parser/problem.jl:52


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class SCPScaling:
    """src/solvers/scp.jl:39-49; built from bounding boxes as in scp.jl:479-511."""

    def __init__(self, x_bbox, u_bbox, p_bbox):
        zero_intvl_tol = math.sqrt(np.finfo(np.float64).eps)  # scp.jl:388

        def one(bbox):
            bbox = np.asarray(bbox, dtype=np.float64).reshape(-1, 2)
            lo, hi = bbox[:, 0], bbox[:, 1]
            S = (hi - lo) / 1.0  # interval [0,1] -> width 1  (scp.jl:479-487)
            S = np.where(S < zero_intvl_tol, 1.0, S)
            c = lo - S * 0.0
            return np.ascontiguousarray(S), np.ascontiguousarray(c)

        self.Sx, self.cx = one(x_bbox)
        self.Su, self.cu = one(u_bbox)
        self.Sp, self.cp = one(p_bbox)
        self.iSx, self.iSu, self.iSp = 1.0 / self.Sx, 1.0 / self.Su, 1.0 / self.Sp


class DLTV:
    """Batched discrete LTV system, src/solvers/discretization.jl:28-84."""

    def __init__(self, nx, nu, npF, N, B, method=FOH):
        M = N - 1
        self.A = np.empty((B, M, nx, nx))
        self.B = [np.empty((B, M, nu, nx)), np.empty((B, M, nu, nx))]  # B[0]=B^-, B[1]=B^+
        self.F = np.empty((B, M, max(npF, 1), nx))[:, :, :npF]
        self._F_store = np.empty((B, M, max(npF, 1), nx))
        self.r = np.empty((B, M, nx))
        self.E = np.empty((B, M, nx, nx))
        self.method = method
        self.timing = 0.0


class SCPProblem:
    """src/solvers/scp.jl:63-157: parameters + trajectory problem + common terms.
    Owns the native handle (device scratch + HIP stream)."""

    def __init__(self, pars, traj, batch_capacity=1, device=0):
        self.pars = pars
        self.traj = traj
        mdl = traj.mdl
        assert isinstance(mdl, NativeModel)
        if hasattr(mdl, "bind"):
            mdl.bind(pars)      # models whose constraints depend on the grid (Starship: phase-switch node)
        self.t_grid = linrange(0.0, 1.0, pars.N)  # scp.jl:147
        self.scale = SCPScaling(*mdl.scale_advice())
        L = _lib.lib()
        info = _lib.ScpModelInfo()
        _lib.check(L.scp_model_query(MODEL_IDS[mdl.name], ctypes.byref(info)))
        self.info = info
        # p = [global parameters (info.np); node parameters (info.np_node, N)] (include/scp_mi355x.h, scp_model_info)
        self.nx, self.nu, self.np, self.npF = info.nx, info.nu, info.np + info.np_node * pars.N, info.npF
        self.np_glob, self.np_node = info.np, info.np_node
        assert self.scale.Sp.size == self.np, "scaling advice does not cover the parameter vector"
        self._par = np.ascontiguousarray(mdl.par(), dtype=np.float64)
        assert self._par.size == info.npar
        d = _lib.ScpProblemDesc()
        d.model_id = MODEL_IDS[mdl.name]
        d.model_par = self._par.ctypes.data_as(_lib.c_double_p)
        d.N, d.Nsub, d.disc_method = pars.N, pars.Nsub, pars.disc_method
        d.feas_tol = pars.feas_tol
        s = self.scale
        for nm in ("Sx", "cx", "Su", "cu", "Sp", "cp"):
            setattr(d.scale, nm, getattr(s, nm).ctypes.data_as(_lib.c_double_p))
        d.batch_capacity = batch_capacity
        d.device = device
        self.batch_capacity = batch_capacity
        h = ctypes.c_void_p()
        rc = L.scp_problem_create(ctypes.byref(d), ctypes.byref(h))
        self.handle = h
        if rc != 0:
            msg = L.scp_last_error(h).decode() if h else ""
            if h:
                L.scp_problem_destroy(h)
            self.handle = None
            raise _lib.ScpError(rc, msg)

    def set_discretize_precision(self, bits):
        """arithmetic of discretize! on this handle: 64 (reference) or 32 (tolerance-check variant of K1; Starship, FOH)."""
        _lib.check(_lib.lib().scp_set_discretize_precision(self.handle, int(bits)), self.handle)

    def close(self):
        # dependants (generic subproblem handles, scp_sub_*) hold a pointer to this handle: destroy them first
        for child in list(getattr(self, "_children", [])):
            child.close()
        self._children = []
        if getattr(self, "handle", None):
            _lib.lib().scp_problem_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SubproblemSolutionBatch:
    """Batched analogue of SCPSubproblemSolution's trajectory part
    (src/solvers/ptr.jl:73-101): xd, ud, p, dyn, defect, feas."""

    def __init__(self, xd, ud, p, pbm):
        self.xd = np.ascontiguousarray(xd, dtype=np.float64)
        self.ud = np.ascontiguousarray(ud, dtype=np.float64)
        self.p = np.ascontiguousarray(p, dtype=np.float64)
        B, N, nx = self.xd.shape
        assert N == pbm.pars.N and nx == pbm.nx
        assert self.ud.shape == (B, N, pbm.nu) and self.p.shape == (B, pbm.np)
        self.dyn = DLTV(pbm.nx, pbm.nu, pbm.npF, N, B, pbm.pars.disc_method)
        self.defect = np.full((B, N - 1, nx), np.nan)  # ptr.jl:334
        self.feas = np.zeros(B, dtype=bool)            # ptr.jl:333


def discretize_(ref, pbm):
    """`discretize!(ref, pbm)` (src/solvers/discretization.jl:160-217) for a batch,
    through the C ABI (scp_discretize_batch_host).  Mutates ref.dyn, ref.defect,
    ref.feas, ref.dyn.timing; returns None like the reference."""
    L = _lib.lib()
    B = ref.xd.shape[0]
    feas = np.zeros(B, dtype=np.uint8)
    sec = ctypes.c_double(0.0)
    dyn = ref.dyn
    rc = L.scp_discretize_batch_host(
        pbm.handle, B, _ptr(ref.xd), _ptr(ref.ud), _ptr(ref.p) if pbm.np > 0 else None,
        _ptr(dyn.A), _ptr(dyn.B[0]), _ptr(dyn.B[1]), _ptr(dyn._F_store) if pbm.npF > 0 else None,
        _ptr(dyn.r), _ptr(dyn.E), _ptr(ref.defect), _ptr(feas), ctypes.byref(sec))
    _lib.check(rc, pbm.handle)
    dyn.F = dyn._F_store[:, :, :pbm.npF]
    ref.feas = feas.astype(bool)
    dyn.timing = sec.value
    return None


def propagate(sol, pbm, res=1000):
    """`propagate(sol, pbm; res)` (src/solvers/discretization.jl:515-562) for a batch through the C ABI
    (scp_propagate_batch_host): `sol` carries xd[B,N,nx], ud[B,N,nu], p[B,np].  Returns (tc, xc[B,len(tc),nx]): the sample
    times and the state values of the reference's continuous-time `Trajectory`.  FOH: tc = LinRange(0,1,res).  IMPULSE
    (:542-560): 1 + (N-1) ceil(res/(N-1)) samples -- node 1, then every interval's own grid from the post-impulse state, its
    first time shifted by sqrt(eps) like the reference's."""
    L = _lib.lib()
    xd = np.ascontiguousarray(sol.xd, dtype=np.float64)
    ud = np.ascontiguousarray(sol.ud, dtype=np.float64)
    p = np.ascontiguousarray(sol.p, dtype=np.float64)
    B, N = xd.shape[0], pbm.pars.N
    res = int(res)
    if pbm.pars.disc_method == IMPULSE:
        sub = -(-res // (N - 1))
        tc = [np.array([0.0])]
        for k in range(N - 1):
            t = linrange(pbm.t_grid[k], pbm.t_grid[k + 1], sub)
            t[0] += math.sqrt(np.finfo(np.float64).eps)
            tc.append(t)
        tc = np.concatenate(tc)
    else:
        tc = np.array([(1 - j / (res - 1)) * 0.0 + (j / (res - 1)) * 1.0 for j in range(res)])
    xc = np.zeros((B, tc.size, pbm.nx))
    rc = L.scp_propagate_batch_host(pbm.handle, B, _ptr(xd), _ptr(ud), _ptr(p) if pbm.np > 0 else None, res, _ptr(xc))
    _lib.check(rc, pbm.handle)
    return tc, xc


class LinearTrajectory:
    """`Trajectory(td, ud, :linear)` (src/utils/trajectory.jl): first-order-hold interpolation of nodal values, the `uc` of
    an SCPSolution for the FOH method; sample(t) -> [B, n]."""

    def __init__(self, td, values):
        self.td, self.values = np.asarray(td, float), np.asarray(values, float)     # values [B, N, n]

    def sample(self, t):
        t = min(max(float(t), self.td[0]), self.td[-1])
        k = int(min(max(np.searchsorted(self.td, t, side="left"), 1), self.td.size - 1))
        c = (self.td[k] - t) / (self.td[k] - self.td[k - 1])
        return c * self.values[:, k - 1] + (1.0 - c) * self.values[:, k]


class ImpulseTrajectory:
    """`Trajectory(td, ud, :impulse)` (src/utils/trajectory.jl, diracinterp helper.jl:166-186): the nodal value when t lands
    exactly on a grid node (or beyond the last one), zero otherwise -- the `uc` of an SCPSolution for the IMPULSE method."""

    def __init__(self, td, values):
        self.td, self.values = np.asarray(td, float), np.asarray(values, float)     # values [B, N, n]

    def sample(self, t):
        t = float(t)
        if t >= self.td[-1]:
            return self.values[:, -1].copy()
        t = max(self.td[0], t)
        k = np.nonzero(self.td == t)[0]
        return self.values[:, k[0]].copy() if k.size else np.zeros_like(self.values[:, 0])


def continuous_time(sol, pbm):
    """The continuous-time part of `SCPSolution(history)` (src/solvers/scp.jl:227-237) for a batch solution of any of the
    three algorithms: xc = propagate(last_sol, pbm; res = 2 Nsub (N - 1)) on the device, uc = the first-order-hold input
    trajectory.  Failed problems (status != SCP_SOLVED) get NaN samples (`missing` in the reference).  Attaches and returns
    (tc, xc, uc)."""
    res = 2 * pbm.pars.Nsub * (pbm.pars.N - 1)
    tc, xc = propagate(sol, pbm, res=res)
    ok = np.array([str(st).startswith("SCP_SOLVED") for st in sol.status])
    xc[~ok] = np.nan
    sol.tc, sol.xc = tc, xc
    sol.uc = (LinearTrajectory if pbm.pars.disc_method == FOH else ImpulseTrajectory)(pbm.t_grid, sol.ud)   # scp.jl:233-237
    return tc, xc, sol.uc


def device_guess(pbm, pp):
    """`traj.guess(N)` for a Monte-Carlo batch evaluated on the device (scp_guess_batch_host): pp[B,npp] ->
    (xd[B,N,nx], ud[B,N,nu], p[B,np])."""
    pp = np.ascontiguousarray(np.atleast_2d(pp), dtype=np.float64)
    B, N = pp.shape[0], pbm.pars.N
    xd = np.zeros((B, N, pbm.nx)); ud = np.zeros((B, N, pbm.nu)); p = np.zeros((B, pbm.np))
    _lib.check(_lib.lib().scp_guess_batch_host(pbm.handle, B, _ptr(pp) if pbm.info.npp else None, _ptr(xd), _ptr(ud),
                                               _ptr(p) if pbm.np else None), pbm.handle)
    return xd, ud, p


def device_guess_failures(pbm):
    """instances of the last `device_guess` call for which the model's own guess rule failed (Starship: no velocity crossing or
    no feasible descent duration -- the reference raises an error there) and the straight-line guess was returned instead"""
    return int(_lib.lib().scp_guess_failures(pbm.handle))
