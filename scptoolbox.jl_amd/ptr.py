"""Host-side mirror of the PTR algorithm interface (src/solvers/ptr.jl): Parameters,
create, solve -- batched over independent problem instances (Monte-Carlo initial
conditions).  All arithmetic of an iteration runs in the HIP library through the C
ABI; this module only marshals arrays and mirrors the reference's result types."""
import ctypes
import math
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .scp import FOH, SCPProblem

SOLVER_STATUS = {0: "OPTIMAL", 1: "ALMOST_OPTIMAL", 2: "ITERATION_LIMIT", 3: "NUMERICAL_ERROR"}


@dataclass
class Parameters:
    """`PTR.Parameters`, src/solvers/ptr.jl:57-71 (same field order).  `solver`
    selected ECOS in the reference; here the only backend is the native structured
    interior-point solver and `solver_opts` carries its options
    (maxit, feastol, abstol, reltol, reg, nref, ref_gap, ref_tol, stall, warm, warm_mu, warm_mu_coarse, warm_dev, warm_min_cold -- ECOS
    option names where they exist)."""
    N: int
    Nsub: int
    iter_max: int
    disc_method: int = FOH
    wvc: float = 1e3
    wtr: float = 0.1
    eps_abs: float = 1e-5   # ε_abs
    eps_rel: float = 1e-4   # ε_rel
    feas_tol: float = 1e-3
    q_tr: float = float("inf")
    q_exit: float = float("inf")
    solver: object = None
    solver_opts: dict = field(default_factory=dict)

    def c_struct(self):
        o = self.solver_opts
        c = _lib.ScpPtrParams()
        c.iter_max, c.wvc, c.wtr, c.eps_abs, c.eps_rel = self.iter_max, self.wvc, self.wtr, self.eps_abs, self.eps_rel
        c.q_tr, c.q_exit = self.q_tr, self.q_exit
        c.ipm_max_iter = int(o.get("maxit", 100))
        c.ipm_feastol = float(o.get("feastol", 1e-8))
        c.ipm_abstol = float(o.get("abstol", 1e-8))
        c.ipm_reltol = float(o.get("reltol", 1e-8))
        c.ipm_reg = float(o.get("reg", 1e-12))   # a factorisation that breaks down is repeated with 10x (up to 1e-8); >= 3e-10 stalls the end-game
        # one step of iterative refinement per Newton solve once relgap < ref_gap.  Measured on the rocket bench batch (round 3):
        # nref = 0 is 14 % faster at the same SCP outcomes, but 96 % of the subproblems then stop at ECOS's REDUCED accuracy
        # (ALMOST_OPTIMAL, gap stalling at 1e-6 ... 5e-5) instead of 40 % -- the default keeps the accuracy
        c.ipm_nref = int(o.get("nref", 1))
        c.ipm_ref_gap = float(o.get("ref_gap", 1e-2))
        c.ipm_split_step = int(o.get("split_step", 0))   # 1: separate primal/dual steps when P = 0 (-10 % iterations, less robust)
        c.ipm_ref_tol = float(o.get("ref_tol", 0.0))   # > 0: skip refinement when the residual is below ref_tol*feastol
        c.ipm_stall = int(o.get("stall", 3))
        # warm start of the subproblem solver from a snapshot of the previous solve (header: scp_ptr_params.ipm_warm): the fine
        # snapshot (mu <= warm_mu) when the reference moved less than warm_dev, else the coarse one (mu <= warm_mu_coarse)
        c.ipm_warm = int(o.get("warm", 1))
        # four snapshot levels since round 6 (two before: 1e-1 and 1e-7): the level is chosen by the deviation of the previous solution
        # (include/scp_mi355x.h).  Once a PTR run has converged its reference moves by ~1e-8 and a solve restarts from the previous
        # solve's iterate at mu <= 1e-10: 1 ... 5 IPM iterations per launch instead of 9 ... 16 in the second half of a 15-iteration
        # run; the mid level saves ~10 iterations in the launch after the last large move.  Same optima (levels and bounds swept on
        # the CPU twin -- oracle/cpu_ptr.cpp, SCP_CPU_LVL_MU / SCP_CPU_LVL_DEV -- then on the 4096 batch).
        c.ipm_warm_mu = float(o.get("warm_mu", 1e-8))
        c.ipm_warm_mu_coarse = float(o.get("warm_mu_coarse", 1e-1))
        c.ipm_warm_dev = float(o.get("warm_dev", 1e-3))
        c.ipm_warm_min_cold = int(o.get("warm_min_cold", 25))   # gates the COARSE level only
        c.ipm_warm_mu_mid = float(o.get("warm_mu_mid", 1e-5)); c.ipm_warm_dev_mid = float(o.get("warm_dev_mid", 1e-1))
        c.ipm_warm_mu_vfine = float(o.get("warm_mu_vfine", 1e-10)); c.ipm_warm_dev_vfine = float(o.get("warm_dev_vfine", 1e-6))
        c.ipm_wpe = int(o.get("wpe", 0))
        return c


def create(pars, traj, batch_capacity=1, device=0):
    """`PTR.create(pars, traj)`, src/solvers/ptr.jl:148-195."""
    traj.scp = pars
    return SCPProblem(pars, traj, batch_capacity=batch_capacity, device=device)


@dataclass
class SCPSolutionBatch:
    """Batched `SCPSolution` (src/solvers/scp.jl:105-119): status strings per problem as in
    scp.jl:213,222, discrete trajectories, cost, iterations."""
    status: list
    algo: str
    iterations: np.ndarray
    cost: np.ndarray      # J_aug of the last subproblem (scp.jl:238; Inf on failure, scp.jl:219)
    J: np.ndarray         # original cost J of the last subproblem
    td: np.ndarray
    xd: np.ndarray        # [B, N, nx]
    ud: np.ndarray        # [B, N, nu]
    p: np.ndarray         # [B, np]
    J_aug: np.ndarray
    feas: np.ndarray
    defect: np.ndarray


@dataclass
class SCPHistoryBatch:
    """Per-iteration records (the reference keeps whole Subproblem objects, scp.jl:122-124; here the
    scalars its progress table prints): arrays [iter, B]."""
    J: np.ndarray
    J_tr: np.ndarray
    J_vc: np.ndarray
    J_aug: np.ndarray
    deviation: np.ndarray
    improv_rel: np.ndarray
    feas: np.ndarray
    solver_status: np.ndarray
    solver_iters: np.ndarray
    active: np.ndarray
    gap: np.ndarray
    pres: np.ndarray
    dres: np.ndarray


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _guess_batch(pbm, pp):
    xs, us, ps = [], [], []
    for b in range(pp.shape[0]):
        x, u, p = pbm.traj.guess(pbm.pars.N, pp[b])
        xs.append(x); us.append(u); ps.append(p)
    return (np.ascontiguousarray(np.stack(xs)), np.ascontiguousarray(np.stack(us)),
            np.ascontiguousarray(np.stack(ps).reshape(pp.shape[0], -1)))


def solve(pbm, pp=None, warm=None, all_reduce=None, device_guess=False):
    """`PTR.solve(pbm[, warm])` (src/solvers/ptr.jl:448-532) for a batch.

    pp: [B, npp] per-problem data (None = one nominal problem).  warm: optional (xd, ud, p)
    arrays replacing traj.guess (scp.jl:532-539).  all_reduce: optional callable n -> global n
    (the per-iteration convergence all-reduce across GPUs; identity on one GPU).  device_guess: generate the
    initial guesses on the device (scp_ptr_init_guess_host).
    Returns (SCPSolutionBatch, SCPHistoryBatch)."""
    L = _lib.lib()
    pars = pbm.pars
    mdl = pbm.traj.mdl
    pp = np.ascontiguousarray(mdl.nominal_pp()[None] if pp is None else pp, dtype=np.float64)
    B = pp.shape[0]
    if pars.q_tr != math.inf or pars.q_exit != math.inf or not pbm.info.structured:
        # trust-region norms 1, 2, 4 (ptr.jl:582-739) and models without a stage-structured fast path (Starship):
        # generic conic path
        return _solve_generic(pbm, pp, warm, all_reduce)
    upload(pbm, pp, warm, device_guess)   # device_guess: traj.guess runs on the device, only pp is uploaded
    na = ctypes.c_int(B)
    while True:
        _lib.check(L.scp_ptr_iterate(pbm.handle, ctypes.byref(na)), pbm.handle)
        n = na.value if all_reduce is None else all_reduce(na.value)
        if n <= 0:
            break
    return _collect(pbm, B)


def _qnorm(v, q):
    return np.abs(v).max(axis=-1) if q == math.inf else (np.abs(v) ** q).sum(axis=-1) ** (1.0 / q)


def _generic_sub(pbm):
    """the PTR subproblem of this problem as a conic template bound to the device (built once per SCPProblem)"""
    from .generic import GenericSubproblem
    from .subproblem import ModelRows, build_ptr
    if getattr(pbm, "_generic_sub", None) is None:
        pars = pbm.pars
        T = build_ptr(ModelRows(pbm.traj.mdl), pars.N, pbm.scale, pars.wvc, pars.wtr, pars.q_tr)
        pbm._generic_sub = GenericSubproblem(pbm, T)
    return pbm._generic_sub


# ECOS option names of PTR.Parameters.solver_opts -> options of the generic conic backend (include/scp_conic.h); the
# remaining keys belong to the structured fast path only and are refused here instead of being dropped silently
_GENERIC_OPTS = {"maxit": "max_iter", "max_iter": "max_iter", "feastol": "feastol", "abstol": "abstol", "reltol": "reltol",
                 "reg": "reg", "nref": "nref", "ref_tol": "ref_tol", "dyn_eps": "dyn_eps", "dyn_delta": "dyn_delta", "step": "step"}
_STRUCTURED_ONLY = ("ref_gap", "stall", "split_step", "warm", "warm_mu", "warm_mu_coarse", "warm_dev", "warm_min_cold", "wpe")


def generic_solver_options(solver_opts):
    out = {}
    for k, v in solver_opts.items():
        if k in _GENERIC_OPTS:
            out[_GENERIC_OPTS[k]] = v
        elif k in _STRUCTURED_ONLY:
            raise _lib.ScpError(7, "solver option %r belongs to the stage-structured PTR solver; this problem runs on the "
                                   "generic conic backend (model without a fast path, or q_tr != Inf)" % k)
        else:
            raise _lib.ScpError(1, "unknown solver option %r" % k)
    return out


def _solve_generic(pbm, pp, warm=None, all_reduce=None):
    """PTR loop (ptr.jl:448-532) on the generic subproblem pipeline, RESIDENT on the device (scp_ptr_generic_*): every
    iteration is discretize! + formulate + conic solve + discretize! + the cost split, the stopping rule (ptr.jl:908-932,
    solution_deviation scp.jl:909-931 in the q_exit norm) and ref = sol; only the active count comes back per iteration."""
    from .conic import default_options
    pars = pbm.pars
    if not pars.q_exit >= 1:
        raise _lib.ScpError(1, "q_exit must be >= 1 or Inf")
    L = _lib.lib()
    sub = _generic_sub(pbm)
    B = pp.shape[0]
    xd, ud, p = _guess_batch(pbm, pp) if warm is None else [np.ascontiguousarray(a, dtype=np.float64) for a in warm]
    cp = _lib.ScpPtrGenericParams()
    cp.iter_max, cp.wvc, cp.wtr, cp.eps_abs, cp.eps_rel, cp.q_exit = pars.iter_max, pars.wvc, pars.wtr, pars.eps_abs, pars.eps_rel, pars.q_exit
    cp.cost_const = sub.T.cost_const
    cp.solver = default_options(**generic_solver_options(pars.solver_opts))
    sub._check(L.scp_ptr_generic_init_host(sub._h, B, ctypes.byref(cp), _vp(xd), _vp(ud), _vp(p) if pbm.np else None,
                                           _vp(pp) if pbm.info.npp else None))
    na = ctypes.c_int(1)
    k, n = 0, 1
    while k < pars.iter_max and n > 0:
        sub._check(L.scp_ptr_generic_iterate(sub._h, ctypes.byref(na)))
        n = na.value if all_reduce is None else all_reduce(na.value)
        k += 1
    N = pars.N
    x = np.zeros((B, N, pbm.nx)); u = np.zeros((B, N, pbm.nu)); po = np.zeros((B, pbm.np))
    status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); cost = np.zeros((B, 4)); feas = np.zeros(B, np.uint8)
    defect = np.zeros((B, N - 1, pbm.nx)); hist = np.zeros((pars.iter_max, B, _lib.HIST_WIDTH))
    sub._check(L.scp_ptr_generic_get_host(sub._h, _vp(x), _vp(u), _vp(po) if pbm.np else None, _vp(status), _vp(iters), _vp(cost),
                                          _vp(feas), _vp(defect), _vp(hist)))
    keys = ("J", "J_tr", "J_vc", "J_aug", "deviation", "improv_rel", "feas", "solver_status", "solver_iters", "active", "gap",
            "pres", "dres")
    H = {kk: hist[:, :, j] for j, kk in enumerate(keys)}
    last_st = np.array([int(H["solver_status"][max(int(iters[b]) - 1, 0), b]) for b in range(B)])
    failed = status != 0
    st = ["SCP_FAILED (%s)" % SOLVER_STATUS.get(int(last_st[b]), "?") if failed[b] else "SCP_SOLVED" for b in range(B)]
    sol = SCPSolutionBatch(status=st, algo="PTR (backend: MI355X generic conic IPM)", iterations=iters,
                           cost=np.where(failed, math.inf, cost[:, 3]), J=cost[:, 0].copy(), td=pbm.t_grid.copy(), xd=x, ud=u, p=po,
                           J_aug=cost[:, 3].copy(), feas=feas.astype(bool), defect=defect)
    hb = SCPHistoryBatch(**{kk: (H[kk] > 0 if kk in ("feas", "active") else (H[kk].astype(int) if kk.startswith("solver_") else H[kk]))
                            for kk in keys})
    return sol, hb


def upload(pbm, pp=None, warm=None, device_guess=False):
    """Upload guesses + per-problem data and discretise the guess (start of PTR.solve).
    device_guess=True: only pp is uploaded, the model's guess rule runs on the device (scp_ptr_init_guess_host)."""
    L = _lib.lib()
    mdl = pbm.traj.mdl
    pp = np.ascontiguousarray(mdl.nominal_pp()[None] if pp is None else pp, dtype=np.float64)
    B = pp.shape[0]
    if device_guess and warm is None:
        cp = pbm.pars.c_struct()
        _lib.check(L.scp_ptr_init_guess_host(pbm.handle, B, ctypes.byref(cp), _vp(pp)), pbm.handle)
        return B
    xd, ud, p = _guess_batch(pbm, pp) if warm is None else [np.ascontiguousarray(a, dtype=np.float64) for a in warm]
    cp = pbm.pars.c_struct()
    _lib.check(L.scp_ptr_init_host(pbm.handle, B, ctypes.byref(cp), _vp(xd), _vp(ud), _vp(p) if pbm.np else None,
                                   _vp(pp)), pbm.handle)
    return B


def restart(pbm):
    """Reset the resident batch to its uploaded guesses on the device (no host traffic)."""
    _lib.check(_lib.lib().scp_ptr_restart(pbm.handle), pbm.handle)


def iterate(pbm):
    """One batched PTR iteration; returns the number of still-active local problems."""
    na = ctypes.c_int(0)
    _lib.check(_lib.lib().scp_ptr_iterate(pbm.handle, ctypes.byref(na)), pbm.handle)
    return na.value


def run_resident(pbm, all_reduce=None):
    """Iterate the resident batch until no problem (on any rank) is active; returns #iterations."""
    n_it = 0
    while True:
        n = iterate(pbm)
        n_it += 1
        if all_reduce is not None:
            n = all_reduce(n)
        if n <= 0:
            return n_it


def kernel_timing(pbm, reset=False):
    """(seconds[4], launches[4]) per kernel: discretize, assemble, ipm, extract+update."""
    sec = (ctypes.c_double * 4)()
    cnt = (ctypes.c_long * 4)()
    _lib.check(_lib.lib().scp_get_kernel_timing(pbm.handle, sec, cnt, 1 if reset else 0), pbm.handle)
    return list(sec), list(cnt)


def collect(pbm, B):
    return _collect(pbm, B)


def _collect(pbm, B):
    L = _lib.lib()
    pars = pbm.pars
    N, nx, nu, np_ = pars.N, pbm.nx, pbm.nu, pbm.np
    xd = np.empty((B, N, nx)); ud = np.empty((B, N, nu)); p = np.empty((B, np_))
    status = np.zeros(B, dtype=np.int32); iters = np.zeros(B, dtype=np.int32)
    cost = np.empty((B, 4)); feas = np.zeros(B, dtype=np.uint8); defect = np.empty((B, N - 1, nx))
    hist = np.zeros((pars.iter_max, B, _lib.HIST_WIDTH))
    _lib.check(L.scp_ptr_get_host(pbm.handle, _vp(xd), _vp(ud), _vp(p) if np_ else None, _vp(status), _vp(iters),
                                  _vp(cost), _vp(feas), _vp(defect), _vp(hist)), pbm.handle)
    # final status string of the LAST subproblem solve (scp.jl:211-222)
    st = []
    J = cost[:, 3].copy()   # SCPSolution.cost = last_sol.J_aug (scp.jl:238)
    for b in range(B):
        if status[b] == 0:
            st.append("SCP_SOLVED")
        else:
            last = int(hist[max(iters[b] - 1, 0), b, 7])
            st.append("SCP_FAILED (%s)" % SOLVER_STATUS.get(last, "?"))
            J[b] = math.inf
    sol = SCPSolutionBatch(status=st, algo="PTR (backend: MI355X structured IPM)", iterations=iters, cost=J,
                           J=cost[:, 0].copy(), td=pbm.t_grid.copy(), xd=xd, ud=ud, p=p, J_aug=cost[:, 3].copy(), feas=feas.astype(bool),
                           defect=defect)
    h = hist
    history = SCPHistoryBatch(J=h[:, :, 0], J_tr=h[:, :, 1], J_vc=h[:, :, 2], J_aug=h[:, :, 3], deviation=h[:, :, 4],
                              improv_rel=h[:, :, 5], feas=h[:, :, 6] > 0, solver_status=h[:, :, 7].astype(int),
                              solver_iters=h[:, :, 8].astype(int), active=h[:, :, 9] > 0, gap=h[:, :, 10],
                              pres=h[:, :, 11], dres=h[:, :, 12])
    return sol, history


def solve_subproblem_(pbm, xd_ref, ud_ref, p_ref, pp=None):
    """`solve_subproblem!` (src/solvers/scp.jl:942-950) for a batch of reference trajectories:
    formulate + solve the PTR subproblem about them and discretise the new point.  Returns a dict."""
    L = _lib.lib()
    pars = pbm.pars
    mdl = pbm.traj.mdl
    if pars.q_tr != math.inf or not pbm.info.structured:
        return _generic_sub(pbm).solve(xd_ref, ud_ref, p_ref, pp=pp, want_conic=True)
    xd_ref = np.ascontiguousarray(xd_ref, dtype=np.float64); ud_ref = np.ascontiguousarray(ud_ref, dtype=np.float64)
    B, N, nx = xd_ref.shape
    nu, np_ = pbm.nu, pbm.np
    p_ref = np.ascontiguousarray(p_ref, dtype=np.float64).reshape(B, np_)
    pp = np.ascontiguousarray(np.repeat(mdl.nominal_pp()[None], B, 0) if pp is None else pp, dtype=np.float64)
    out = dict(x=np.empty((B, N, nx)), u=np.empty((B, N, nu)), p=np.empty((B, np_)), cost=np.empty((B, 4)),
               eta=np.empty((B, 2 * N + 1)), status=np.zeros(B, dtype=np.int32), iters=np.zeros(B, dtype=np.int32),
               info=np.empty((B, 8)), defect=np.empty((B, N - 1, nx)), feas=np.zeros(B, dtype=np.uint8))
    sec = ctypes.c_double(0.0)
    cp = pars.c_struct()
    rc = L.scp_ptr_solve_subproblem_batch_host(
        pbm.handle, B, ctypes.byref(cp), _vp(xd_ref), _vp(ud_ref), _vp(p_ref) if np_ else None, _vp(pp),
        _vp(out["x"]), _vp(out["u"]), _vp(out["p"]) if np_ else None, _vp(out["cost"]), _vp(out["eta"]),
        _vp(out["status"]), _vp(out["iters"]), _vp(out["info"]), _vp(out["defect"]), _vp(out["feas"]), ctypes.byref(sec))
    _lib.check(rc, pbm.handle)
    out["seconds"] = sec.value
    out["feas"] = out["feas"].astype(bool)
    out["J"], out["J_tr"], out["J_vc"], out["J_aug"] = (out["cost"][:, i] for i in range(4))
    out.update(virtual_controls(pbm, B))
    return out


def virtual_controls(pbm, B):
    """vd, vs, vic, vtc, P, Pf of the last solved subproblem (`SubproblemSolution(spbm)`, src/solvers/ptr.jl:399-432)
    through scp_ptr_get_virtual_controls_host: dict of arrays vd[B,N-1,nx], vs[B,N,ns], vic[B,nic], vtc[B,ntc],
    P[B,N], Pf[B,2]."""
    L = _lib.lib()
    N, nx, info = pbm.pars.N, pbm.nx, pbm.info
    out = dict(vd=np.zeros((B, N - 1, nx)), vs=np.zeros((B, N, info.ns)), vic=np.zeros((B, info.nic)),
               vtc=np.zeros((B, info.ntc)), P=np.zeros((B, N)), Pf=np.zeros((B, 2)))
    _lib.check(L.scp_ptr_get_virtual_controls_host(pbm.handle, _vp(out["vd"]), _vp(out["vs"]) if info.ns else None,
                                                   _vp(out["vic"]) if info.nic else None,
                                                   _vp(out["vtc"]) if info.ntc else None, _vp(out["P"]), _vp(out["Pf"])),
               pbm.handle)
    return out


def debug_stage_problem(pbm, b=0):
    """Diagnostic: raw stage-form slab of problem b (layout csrc/stage_problem.hpp)."""
    L = _lib.lib()
    n = ctypes.c_long(0)
    _lib.check(L.scp_debug_get_stage_problem(pbm.handle, b, None, ctypes.byref(n)), pbm.handle)
    buf = np.empty(n.value)
    _lib.check(L.scp_debug_get_stage_problem(pbm.handle, b, _vp(buf), ctypes.byref(n)), pbm.handle)
    return buf


# ------------------------------------------------------------------------------------------------------------------
# Sub-batches on separate streams (scp_ptr_iterate_async / scp_ptr_poll)
# ------------------------------------------------------------------------------------------------------------------

class SCPProblemGroup:
    """A batch split into `streams` contiguous sub-batches, each with its own handle (device scratch + HIP stream).
    The sub-batches' kernels overlap on the GPU and several PTR iterations can be enqueued ahead, so the slowest
    problems of one launch (the subproblem solver's iteration count varies about 3x between problems) do not idle the
    rest of the chip.  Results are identical to one handle: problems are independent."""

    def __init__(self, pars, traj, batch_capacity, streams=8, device=0):
        from .dist import shard_range
        streams = max(1, min(int(streams), int(batch_capacity)))
        self.pars, self.traj, self.batch_capacity, self.streams = pars, traj, batch_capacity, streams
        self.ranges = [shard_range(batch_capacity, g, streams) for g in range(streams)]
        if streams > 1 and batch_capacity > 1024 and "wpe" not in pars.solver_opts:
            # the sub-batches share the chip: together they are a large batch -> two-waves-per-SIMD solver variant
            import copy
            pars = copy.copy(pars)
            pars.solver_opts = dict(pars.solver_opts, wpe=2)
            self.pars = pars
        self.parts = [create(pars, traj, batch_capacity=hi - lo, device=device) for lo, hi in self.ranges]
        # sub-batches at DIFFERENT stream priorities (scp_set_stream_priority): no lockstep between their K3 launches, the short
        # kernels of the high-priority sub-batch overtake the other's pending K3 workgroups.  SCP_STREAM_PRIORITIES=0 switches it off.
        import os
        if streams > 1 and os.environ.get("SCP_STREAM_PRIORITIES", "1") != "0":
            for i, p in enumerate(self.parts):
                _lib.check(_lib.lib().scp_set_stream_priority(p.handle, 1 if i == 0 else (-1 if i == streams - 1 else 0)), p.handle)
        p0 = self.parts[0]
        self.scale, self.info, self.t_grid = p0.scale, p0.info, p0.t_grid
        self.nx, self.nu, self.np = p0.nx, p0.nu, p0.np

    def close(self):
        for p in self.parts:
            p.close()


def group_upload(grp, pp, device_guess=False):
    pp = np.ascontiguousarray(pp, dtype=np.float64)
    assert pp.shape[0] == grp.batch_capacity
    for p, (lo, hi) in zip(grp.parts, grp.ranges):
        upload(p, pp[lo:hi], device_guess=device_guess)
    return pp.shape[0]


def group_restart(grp):
    for p in grp.parts:
        restart(p)


def group_run_resident(grp, all_reduce=None, lookahead=1, pipelined=False):
    """Iterate every sub-batch until no problem (on any rank) is active.  `lookahead` PTR iterations are enqueued on every
    stream between two convergence checks (lookahead = 1: one all-reduce per iteration as in `run_resident`; a fixed
    iteration count, eps_abs = eps_rel = 0, can enqueue all of them).  Returns the number of iterations executed.

    pipelined (the multi-GPU loop): window k + 1 is enqueued BEFORE the active count of window k is read
    (scp_ptr_poll_iteration waits for that iteration only), so no stream drains at a window boundary and, with a lagged
    all-reduce (dist.make_lagged_all_reduce), the host never waits for the collective either.  Every rank sees the same sequence
    of counts, so all ranks enqueue the same number of windows; windows enqueued after global convergence (one for the
    pipeline, one for the lag) only launch kernels that skip stopped problems, and nothing at all beyond iter_max."""
    L = _lib.lib()
    na = ctypes.c_int(0)
    iter_max = grp.pars.iter_max

    def enqueue():
        for _ in range(lookahead):
            for p in grp.parts:
                _lib.check(L.scp_ptr_iterate_async(p.handle), p.handle)
    if not pipelined:
        n_it = 0
        while True:
            enqueue()
            n_it += lookahead
            n = 0
            for p in grp.parts:
                _lib.check(L.scp_ptr_poll(p.handle, ctypes.byref(na)), p.handle)
                n += na.value
            if all_reduce is not None:
                n = all_reduce(n)
            if n <= 0:
                return n_it
    enqueue()
    n_enq = lookahead
    while True:
        enqueue()
        n_enq += lookahead
        n = 0
        for p in grp.parts:
            _lib.check(L.scp_ptr_poll_iteration(p.handle, n_enq - lookahead, ctypes.byref(na)), p.handle)
            n += na.value
        if all_reduce is not None:
            n = all_reduce(n)
        if n <= 0:
            #  the iterations up to the window whose count was 0 -- with a lagged all-reduce `n` is the count of the
            # window before, so one more window is subtracted; the no-op windows enqueued behind it are not counted
            lag = 1 if getattr(all_reduce, "flush", None) is not None else 0
            return max(lookahead, min(n_enq - lookahead * (1 + lag), iter_max))


def group_run_sharded(grp, comm=None, lookahead=1):
    """The multi-GPU loop BEHIND the C ABI (scp_ptr_run_sharded): windows of `lookahead` iterations on every sub-batch stream, the
    ranks' active counts summed by RCCL on the device (comm = dist.Communicator; None: this process only), window w + 1 enqueued
    before the global count of window w is read.  Returns (iterations executed until no problem was active on any rank,
    collectives issued)."""
    L = _lib.lib()
    if comm is not None and not comm._h:
        raise ValueError("group_run_sharded: the communicator is closed (pass comm=None for the single-process loop)")
    arr = (ctypes.c_void_p * len(grp.parts))(*[p.handle for p in grp.parts])
    it, nc = ctypes.c_int(0), ctypes.c_int(0)
    rc = L.scp_ptr_run_sharded(comm._h if comm is not None else None, arr, len(grp.parts), int(lookahead), ctypes.byref(it), ctypes.byref(nc))
    if rc != 0:
        msg = L.scp_comm_last_error(comm._h if comm is not None else None).decode(errors="replace")
        raise _lib.ScpError(rc, msg)
    return it.value, nc.value


def group_sync(grp):
    """wait for everything enqueued on the sub-batches' streams (and fold the kernel time stamps into the handles' totals)"""
    L = _lib.lib()
    na = ctypes.c_int(0)
    for p in grp.parts:
        _lib.check(L.scp_ptr_poll(p.handle, ctypes.byref(na)), p.handle)


def group_collect(grp):
    sols, hists = zip(*[_collect(p, hi - lo) for p, (lo, hi) in zip(grp.parts, grp.ranges)])
    cat = lambda xs, ax=0: np.concatenate(xs, axis=ax)
    sol = SCPSolutionBatch(status=sum((s.status for s in sols), []), algo=sols[0].algo, iterations=cat([s.iterations for s in sols]),
                           cost=cat([s.cost for s in sols]), J=cat([s.J for s in sols]), td=sols[0].td, xd=cat([s.xd for s in sols]),
                           ud=cat([s.ud for s in sols]), p=cat([s.p for s in sols]), J_aug=cat([s.J_aug for s in sols]),
                           feas=cat([s.feas for s in sols]), defect=cat([s.defect for s in sols]))
    hist = SCPHistoryBatch(**{f: cat([getattr(h, f) for h in hists], 1) for f in SCPHistoryBatch.__dataclass_fields__})
    return sol, hist


def group_kernel_timing(grp, reset=False):
    sec, cnt = np.zeros(4), np.zeros(4, dtype=np.int64)
    for p in grp.parts:
        s_, c_ = kernel_timing(p, reset=reset)
        sec += s_; cnt += c_
    return sec.tolist(), cnt.tolist()
