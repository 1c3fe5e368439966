"""GuSTO on the MI355X behind the reference's solver contract (src/solvers/gusto.jl).

    pars = GuSTO.Parameters(N=30, Nsub=15, iter_max=15, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.9, beta_sh=2,
                            beta_gr=2, gamma_fail=5, eta_init=10, eta_lb=1e-3, eta_ub=10, mu=0.8, iter_mu=6,
                            eps_abs=0, eps_rel=0, feas_tol=1e-3)                    # test/examples/quadrotor/tests.jl:86-130
    pbm = GuSTO.create(pars, traj, batch_capacity=B)        # gusto.jl:169-201
    sol, history = GuSTO.solve(pbm, pp)                     # gusto.jl:425-502

The subproblem (un-relaxed dynamics / boundary conditions, U hard, soft quadratic penalties lambda v^2 on the state
constraints and on the trust-region excess, gusto.jl:534-550, 725-995, 1056-1170) is one conic template with the two
per-problem scalars (eta, lambda) as sources -- lambda weights the diagonal of P, so the quadratic cost VALUES are per
problem while the pattern and the symbolic factorisation are shared.  The loop (discretize!, formulate, solve, the
solution costs :391-407, check_stopping_criterion! :1203-1230, update_trust_region! :1245-1427) runs on the device
(csrc/scp_generic.hpp).  Both penalties of the reference: `pen = "quad"` and, since round 4, `pen = "softplus"` (exponential cones in the
conic solver, gusto.jl:996-1031).  Restriction (subproblem.build_gusto): s(t, k, x, p) independent of the input."""
import ctypes

import numpy as np

from . import _lib
from .conic import default_options
from .generic import GenericSubproblem, _ptr
from .scp import FOH, SCPProblem
from .scvx import SCPSolutionBatch
from .subproblem import ModelRows, build_correct_convex, build_gusto

H_NAMES = ("L", "L_st", "L_tr", "J_aug", "J_st", "rho", "eta", "lam", "eta_next", "lam_next", "flags", "deviation",
           "solver_status", "solver_iters", "dyn_error", "dyn_nrml")
FLAG_ACCEPTED, FLAG_STOP, FLAG_TRUST_VIOLATED, FLAG_CONSTRAINTS_FEASIBLE, FLAG_DYN_FEASIBLE = 1, 2, 4, 8, 16


class Parameters:
    """GuSTO.Parameters, src/solvers/gusto.jl:59-85 (lam = λ, rho = ρ, beta = β, gamma = γ, eta = η, mu = μ)."""

    def __init__(self, N, Nsub, iter_max, lam_init, lam_max, rho_0, rho_1, beta_sh, beta_gr, gamma_fail, eta_init, eta_lb,
                 eta_ub, mu, iter_mu, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3, pen="quad", hom=500.0, q_tr=np.inf,
                 q_exit=np.inf, disc_method=FOH, solver_opts=None):
        if not q_exit >= 1:
            raise ValueError("q_exit must be >= 1 or Inf (norm of solution_deviation, scp.jl:909-931)")
        if pen not in ("quad", "softplus"):
            raise ValueError("pen must be 'quad' or 'softplus' (gusto.jl:79-80)")
        if pen == "softplus" and not hom > 0:
            raise ValueError("pen = :softplus needs hom > 0")
        self.N, self.Nsub, self.iter_max = N, Nsub, iter_max
        self.lam_init, self.lam_max, self.rho_0, self.rho_1 = lam_init, lam_max, rho_0, rho_1
        self.beta_sh, self.beta_gr, self.gamma_fail = beta_sh, beta_gr, gamma_fail
        self.eta_init, self.eta_lb, self.eta_ub, self.mu, self.iter_mu = eta_init, eta_lb, eta_ub, mu, iter_mu
        self.eps_abs, self.eps_rel, self.feas_tol, self.q_tr, self.q_exit = eps_abs, eps_rel, feas_tol, q_tr, q_exit
        self.pen, self.hom, self.disc_method = pen, hom, disc_method
        self.solver_opts = dict(solver_opts or {})

    def c_struct(self, nst):
        c = _lib.ScpGustoParams()
        for k in ("iter_max", "lam_init", "lam_max", "rho_0", "rho_1", "beta_sh", "beta_gr", "gamma_fail", "eta_init",
                  "eta_lb", "eta_ub", "mu", "iter_mu", "eps_abs", "eps_rel", "q_tr", "q_exit"):
            setattr(c, k, getattr(self, k))
        c.nst = nst
        c.solver = default_options(**self.solver_opts)
        c.pen, c.hom = (1 if self.pen == "softplus" else 0), float(self.hom)
        return c


class GuSTOProblem(SCPProblem):
    def __init__(self, pars, traj, batch_capacity=1, device=0):
        super().__init__(pars, traj, batch_capacity, device)
        mr = ModelRows(traj.mdl)
        self.template = build_gusto(mr, pars.N, self.scale, pars.q_tr, pen=pars.pen, hom=pars.hom)
        self.sub = GenericSubproblem(self, self.template)
        self.proj = GenericSubproblem(self, build_correct_convex(mr, pars.N, self.scale))

    def close(self):
        for s in ("sub", "proj"):
            if getattr(self, s, None) is not None:
                getattr(self, s).close()
                setattr(self, s, None)
        super().close()


def create(pars, traj, batch_capacity=1, device=0):
    return GuSTOProblem(pars, traj, batch_capacity, device)


def solve(pbm, pp=None, guess=None, project_guess=True, all_reduce=None):
    """`GuSTO.solve(pbm[, warm])` for a Monte-Carlo batch (pp[B,npp]); guess = (xd, ud, p) arrays or None (traj.guess).
    The guess is projected onto the convex sets first (correct_convex!, gusto.jl:516-521) unless project_guess=False."""
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
    cp = pbm.pars.c_struct(pbm.template.nst)
    s = pbm.sub
    s._check(L.scp_gusto_init_host(s._h, pbm.proj._h if project_guess else None, B, ctypes.byref(cp), _ptr(xd), _ptr(ud),
                                   _ptr(p) if pbm.np else None, _ptr(pp) if pbm.info.npp else None))
    na = ctypes.c_int(1)
    k = 0
    n = 1
    while k < pbm.pars.iter_max and n > 0:      # all_reduce: n -> global n (the per-iteration convergence all-reduce of a
        s._check(L.scp_gusto_iterate(s._h, ctypes.byref(na)))  # batch sharded over GPUs, dist.py; identity on one GPU)
        n = na.value if all_reduce is None else all_reduce(na.value)
        k += 1
    N = pbm.pars.N
    sol = SCPSolutionBatch()
    sol.xd = np.zeros((B, N, pbm.nx)); sol.ud = np.zeros((B, N, pbm.nu)); sol.p = np.zeros((B, pbm.np))
    status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); cost = np.zeros((2, B)); feas = np.zeros(B, np.uint8)
    sol.defect = np.zeros((B, N - 1, pbm.nx))
    hist = np.zeros((pbm.pars.iter_max, B, _lib.SCVX_HIST_WIDTH))
    s._check(L.scp_gusto_get_host(s._h, _ptr(sol.xd), _ptr(sol.ud), _ptr(sol.p) if pbm.np else None, _ptr(status),
                                  _ptr(iters), _ptr(cost), _ptr(feas), _ptr(sol.defect), _ptr(hist)))
    names = {0: "SCP_SOLVED", 1: "SCP_FAILED", 2: "SCP_GUESS_PROJECTION_FAILED"}
    sol.status = [names[int(v)] for v in status]
    sol.iterations, sol.feas = iters, feas.astype(bool)
    sol.J_ref, sol.cost = cost[0], cost[1]      # cost = J_aug of the last solution
    history = {nm: hist[:, :, j] for j, nm in enumerate(H_NAMES)}
    history["accepted"] = (hist[:, :, 10].astype(int) & FLAG_ACCEPTED) != 0
    history["stop"] = (hist[:, :, 10].astype(int) & FLAG_STOP) != 0
    return sol, history
