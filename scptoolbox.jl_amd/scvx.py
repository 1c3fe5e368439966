"""SCvx on the MI355X behind the reference's solver contract (src/solvers/scvx.jl).

    pars = SCvx.Parameters(N=30, Nsub=15, iter_max=15, lam=30, rho_0=0, rho_1=0.1, rho_2=0.7, beta_sh=2, beta_gr=2,
                           eta_init=1, eta_lb=1e-3, eta_ub=10, eps_abs=0, eps_rel=0, feas_tol=1e-3)
    pbm = SCvx.create(pars, traj, batch_capacity=B)        # scvx.jl:160-206
    sol, history = SCvx.solve(pbm, pp)                     # scvx.jl:459-540

The subproblem (hard trust region ||dx||_q + ||du||_q + ||dp||_q <= eta, scvx.jl:578-678; cost L + lambda (trapz P +
sum Pf), :804-901) and the guess projection `correct_convex!` (scp.jl:275-361) are formulated once as conic templates
(subproblem.py); the whole loop -- discretize!, formulate, solve, check_stopping_criterion! (:711-734),
update_trust_region! (:753-769, 1000-1045) -- runs on the device (csrc/scp_generic.hpp)."""
import ctypes

import numpy as np

from . import _lib
from .conic import default_options
from .generic import GenericSubproblem, _ptr
from .scp import FOH, SCPProblem
from .subproblem import ModelRows, build_correct_convex, build_scvx

H_NAMES = ("L", "L_pen", "L_aug", "J_ref", "J_sol", "pre_improv", "act_improv", "rho", "eta", "eta_next", "accepted", "stop",
           "deviation", "feas", "solver_status", "solver_iters")


class Parameters:
    """SCvx.Parameters, src/solvers/scvx.jl:60-81 (lam = λ, rho_i = ρ_i, beta_* = β_*, eta_* = η_*)."""

    def __init__(self, N, Nsub, iter_max, lam, rho_0, rho_1, rho_2, beta_sh, beta_gr, eta_init, eta_lb, eta_ub, eps_abs=0.0,
                 eps_rel=0.0, feas_tol=1e-3, q_tr=np.inf, q_exit=np.inf, disc_method=FOH, solver_opts=None):
        if not q_exit >= 1:
            raise ValueError("q_exit must be >= 1 or Inf (norm of solution_deviation, scp.jl:909-931)")
        self.N, self.Nsub, self.iter_max, self.lam = N, Nsub, iter_max, lam
        self.rho_0, self.rho_1, self.rho_2, self.beta_sh, self.beta_gr = rho_0, rho_1, rho_2, beta_sh, beta_gr
        self.eta_init, self.eta_lb, self.eta_ub = eta_init, eta_lb, eta_ub
        self.eps_abs, self.eps_rel, self.feas_tol, self.q_tr, self.q_exit = eps_abs, eps_rel, feas_tol, q_tr, q_exit
        self.disc_method = disc_method
        self.solver_opts = dict(solver_opts or {})

    def c_struct(self):
        c = _lib.ScpScvxParams()
        for k in ("iter_max", "lam", "rho_0", "rho_1", "rho_2", "beta_sh", "beta_gr", "eta_init", "eta_lb", "eta_ub",
                  "eps_abs", "eps_rel", "q_exit"):
            setattr(c, k, getattr(self, k))
        c.solver = default_options(**self.solver_opts)
        return c


class SCvxProblem(SCPProblem):
    def __init__(self, pars, traj, batch_capacity=1, device=0):
        super().__init__(pars, traj, batch_capacity, device)
        mr = ModelRows(traj.mdl)
        self.template = build_scvx(mr, pars.N, self.scale, pars.lam, pars.q_tr)
        self.sub = GenericSubproblem(self, self.template)
        self.proj = GenericSubproblem(self, build_correct_convex(mr, pars.N, self.scale))

    def close(self):
        for s in ("sub", "proj"):
            if getattr(self, s, None) is not None:
                getattr(self, s).close()
                setattr(self, s, None)
        super().close()


def create(pars, traj, batch_capacity=1, device=0):
    return SCvxProblem(pars, traj, batch_capacity, device)


class SCPSolutionBatch:
    pass


def solve(pbm, pp=None, guess=None, project_guess=True, all_reduce=None):
    """`SCvx.solve(pbm[, warm])` for a Monte-Carlo batch (pp[B,npp]); guess = (xd, ud, p) arrays or None (traj.guess).
    The guess is projected onto the convex sets first (correct_convex!, scvx.jl:555-565) unless project_guess=False."""
    L = _lib.lib()
    mdl = pbm.traj.mdl
    pp = np.ascontiguousarray(np.atleast_2d(mdl.nominal_pp() if pp is None else pp), np.float64)
    B = pp.shape[0]
    if guess is None:
        g = [pbm.traj.guess(pbm.pars.N, pp[b]) for b in range(B)]
        xd, ud, p = (np.stack([gi[j] for gi in g]) for j in range(3))
    else:
        xd, ud, p = guess
    xd = np.ascontiguousarray(xd, np.float64); ud = np.ascontiguousarray(ud, np.float64); p = np.ascontiguousarray(p, np.float64)
    cp = pbm.pars.c_struct()
    s = pbm.sub
    s._check(L.scp_scvx_init_host(s._h, pbm.proj._h if project_guess else None, B, ctypes.byref(cp), _ptr(xd), _ptr(ud),
                                  _ptr(p) if pbm.np else None, _ptr(pp) if pbm.info.npp else None))
    na = ctypes.c_int(1)
    k = 0
    n = 1
    while k < pbm.pars.iter_max and n > 0:      # all_reduce: n -> global n (the per-iteration convergence all-reduce of a
        s._check(L.scp_scvx_iterate(s._h, ctypes.byref(na)))  # batch sharded over GPUs, dist.py; identity on one GPU)
        n = na.value if all_reduce is None else all_reduce(na.value)
        k += 1
    N = pbm.pars.N
    sol = SCPSolutionBatch()
    sol.xd = np.zeros((B, N, pbm.nx)); sol.ud = np.zeros((B, N, pbm.nu)); sol.p = np.zeros((B, pbm.np))
    status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); cost = np.zeros((2, B)); feas = np.zeros(B, np.uint8)
    sol.defect = np.zeros((B, N - 1, pbm.nx))
    hist = np.zeros((pbm.pars.iter_max, B, _lib.SCVX_HIST_WIDTH))
    s._check(L.scp_scvx_get_host(s._h, _ptr(sol.xd), _ptr(sol.ud), _ptr(sol.p) if pbm.np else None, _ptr(status), _ptr(iters),
                                 _ptr(cost), _ptr(feas), _ptr(sol.defect), _ptr(hist)))
    names = {0: "SCP_SOLVED", 1: "SCP_FAILED", 2: "SCP_GUESS_PROJECTION_FAILED"}
    sol.status = [names[int(v)] for v in status]
    sol.iterations, sol.feas = iters, feas.astype(bool)
    sol.J_ref, sol.cost = cost[0], cost[1]      # cost = nonlinear augmented cost of the last solution (sol.J_aug)
    history = {nm: hist[:, :, j] for j, nm in enumerate(H_NAMES)}
    return sol, history
