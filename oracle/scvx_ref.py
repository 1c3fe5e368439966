"""CPU ORACLE (test infrastructure, NOT product code): literal restatement of the SCvx algorithm of the
reference, ahead of the device implementation.

Follows, line by line:
  parameters            src/solvers/scvx.jl:60-81
  subproblem            src/solvers/scvx.jl:225-303 (variables), 578-678 (trust region: LINF cones for q_tr = Inf and the
                        bound dx_lq[k] + du_lq[k] + dp_lq <= eta), 688-698 + 804-901 (cost L + lambda (trapz(P) + sum Pf))
                        -- assembled by oracle/ptr_ref.py::solve_subproblem(algo="scvx"), which shares dynamics, convex
                        sets, non-convex rows and boundary conditions with PTR exactly as the reference shares scp.jl
  initial guess         src/solvers/scvx.jl:555-565 -> correct_convex!, src/solvers/scp.jl:275-361
  stopping criterion    src/solvers/scvx.jl:711-734  (NB: the "linear" cost of the solution is the ORIGINAL cost only,
                        scvx.jl:972-973 -- SURVEY App. D quirk 3 -- reproduced here)
  actual cost penalty   src/solvers/scvx.jl:924-952
  trust-region update   src/solvers/scvx.jl:753-769, 1000-1045
  loop                  src/solvers/scvx.jl:459-540
The conic solves are oracle/ipm.py.  Parity status: unpinned (no golden data in the reference); pinned here on the
algorithm's own invariants (tests/test_oracle_scvx.py): accepted steps have rho >= rho_0, eta follows the update rule,
the converged trajectory is dynamically feasible and matches the PTR optimum of the same problem.
"""
import numpy as np

from . import ipm
from . import ptr_ref
from .models import MODELS, linrange


class SCvxParameters:
    """SCvx.Parameters, src/solvers/scvx.jl:60-81 (field lam = the reference's λ)."""

    def __init__(self, N, Nsub, iter_max, lam, rho_0, rho_1, rho_2, beta_sh, beta_gr, eta_init, eta_lb, eta_ub,
                 eps_abs, eps_rel, feas_tol, q_tr=np.inf, q_exit=np.inf):
        assert q_tr in (1, 2, 4, np.inf) and q_exit >= 1, "q_tr in {1, 2, 4, Inf} (scvx.jl:593-594)"
        self.N, self.Nsub, self.iter_max, self.lam = N, Nsub, iter_max, lam
        self.rho_0, self.rho_1, self.rho_2, self.beta_sh, self.beta_gr = rho_0, rho_1, rho_2, beta_sh, beta_gr
        self.eta_init, self.eta_lb, self.eta_ub = eta_init, eta_lb, eta_ub
        self.eps_abs, self.eps_rel, self.feas_tol, self.q_tr, self.q_exit = eps_abs, eps_rel, feas_tol, q_tr, q_exit


def quadrotor_test_parameters(N=30, Nsub=15, iter_max=15):
    """test/examples/quadrotor/tests.jl:32-75."""
    return SCvxParameters(N, Nsub, iter_max, lam=30.0, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0,
                          eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)


def compute_original_cost(mdl, pars, x, u, p):
    """compute_original_cost, src/solvers/scp.jl:617-643: phi(x_N, p) + trapz Gamma."""
    t = linrange(0.0, 1.0, pars.N)
    ct = mdl.cost_terms()
    gam = np.array([ct["Qu"] @ (u[k] * u[k]) + ct["lu"] @ u[k] + ct["lx"] @ x[k] for k in range(pars.N)])
    J = ct["tx"] @ x[-1] + ptr_ref._trapz(gam, t)
    if mdl.np:
        J += ct["tp"] @ p + ct["Qp"] @ (p * p)
    return float(J)


def actual_cost_penalty(mdl, pars, sol, pp):
    """actual_cost_penalty!, src/solvers/scvx.jl:924-952: the subproblem's penalty evaluated on the defects of the
    nonlinear propagation and the true non-convex constraint values."""
    N, lam = pars.N, pars.lam
    t = linrange(0.0, 1.0, N)
    P = np.zeros(N)
    for k in range(N):
        dk = sol.defect[k] if k < N - 1 else np.zeros(mdl.nx)
        sk = mdl.s(t[k], k + 1, sol.xd[k], sol.ud[k], sol.p) if mdl.ns else np.zeros(1)
        P[k] = lam * (np.abs(dk).sum() + np.maximum(sk, 0.0).sum())
    gic = mdl.gic(sol.xd[0], sol.p, pp)
    gtc = mdl.gtc(sol.xd[-1], sol.p, pp)
    return float(ptr_ref._trapz(P, t) + lam * (np.abs(gic).sum() + np.abs(gtc).sum()))


def solution_cost(mdl, pars, sol, kind, pp):
    """solution_cost!, src/solvers/scvx.jl:955-984 (values cached on the solution like the reference)."""
    if getattr(sol, "L", None) is None or np.isnan(sol.L):
        sol.L = compute_original_cost(mdl, pars, sol.xd, sol.ud, sol.p)
    if kind == "linear":
        return sol.L
    if getattr(sol, "J_nl", None) is None or np.isnan(sol.J_nl):
        sol.J_nl = sol.L + actual_cost_penalty(mdl, pars, sol, pp)
    return sol.J_nl


def correct_convex(mdl, pars, scale, x, u, p, ipm_opts=None):
    """correct_convex!, src/solvers/scp.jl:275-361: L1-closest trajectory satisfying the convex path constraints."""
    N, nx, nu, np_ = pars.N, mdl.nx, mdl.nu, mdl.np
    t = linrange(0.0, 1.0, N)
    Sx, cx, Su, cu, Sp, cp = scale.Sx, scale.cx, scale.Su, scale.cu, scale.Sp, scale.cp
    P = ptr_ref._Prog()
    xh = [P.var(nx) for _ in range(N)]
    uh = [P.var(nu) for _ in range(N)]
    ph = P.var(np_)

    def phys(M, idx, S, c, Mp, m0):
        terms = [(idx, np.atleast_2d(M) * S[None, :])]
        const = np.array(m0, float) + np.atleast_2d(M) @ c
        if Mp is not None and np_ > 0:
            terms.append((ph, np.atleast_2d(Mp) * Sp[None, :])); const = const + np.atleast_2d(Mp) @ cp
        return terms, const
    for k in range(N):
        for kind, M, Mp, m0 in mdl.X(t[k], k + 1):
            kind, M, Mp, m0 = ptr_ref.lower_linf(kind, M, Mp, m0)
            tr, c0 = phys(M, xh[k], Sx, cx, Mp, m0)
            (P.add_nonpos if kind == "NONPOS" else P.add_soc)(tr, c0)
        for kind, M, Mp, m0 in mdl.U(t[k], k + 1):
            tr, c0 = phys(M, uh[k], Su, cu, Mp, m0)
            (P.add_nonpos if kind == "NONPOS" else P.add_soc)(tr, c0)
    epi_x, epi_u, epi_p = P.var(N), P.var(N), P.var(1)
    xr, ur = (x - cx) / Sx, (u - cu) / Su
    for k in range(N):
        P.add_l1(epi_x[k:k + 1], [(xh[k], np.eye(nx))], -xr[k])     # iSx (x_k - x_ref_k) in scaled variables
        P.add_l1(epi_u[k:k + 1], [(uh[k], np.eye(nu))], -ur[k])
    if np_ > 0:
        P.add_l1(epi_p, [(ph, np.eye(np_))], -(p - cp) / Sp)
    else:
        P.add_nonpos([(epi_p, -np.ones((1, 1)))], np.zeros(1))
    P.add_cost_lin(epi_x, np.ones(N)); P.add_cost_lin(epi_u, np.ones(N)); P.add_cost_lin(epi_p, np.ones(1))
    res = P.solve(**(ipm_opts or {}))
    if res["status"] not in (ipm.OPTIMAL, ipm.ALMOST_OPTIMAL):
        raise RuntimeError("SCP_GUESS_PROJECTION_FAILED (%s)" % res["status"])
    z = res["x"]
    return (np.stack([Sx * z[i] + cx for i in xh]), np.stack([Su * z[i] + cu for i in uh]),
            Sp * z[ph] + cp if np_ else np.zeros(0))


def update_rule(pars, rho, eta):
    """update_rule, src/solvers/scvx.jl:1000-1045 -> (accept, next_eta, tag)."""
    if rho < pars.rho_0:
        return False, max(pars.eta_lb, eta / pars.beta_sh), "S"
    if rho < pars.rho_1:
        return True, max(pars.eta_lb, eta / pars.beta_sh), "S"
    if rho < pars.rho_2:
        return True, eta, ""
    return True, min(pars.eta_ub, pars.beta_gr * eta), "G"


def scvx_solve(model, pars, pp=None, guess=None, ipm_opts=None, verbose=False):
    """`SCvx.solve(pbm)` (src/solvers/scvx.jl:459-540) for one problem.  Returns (status, history)."""
    mdl = MODELS[model]() if isinstance(model, str) else model
    pp = mdl.nominal_pp() if pp is None else np.asarray(pp, float)
    scale = ptr_ref.Scaling(*mdl.bbox())
    x, u, p = mdl.guess(pars.N, pp) if guess is None else guess
    x, u, p = correct_convex(mdl, pars, scale, x, u, p, ipm_opts)        # generate_initial_guess, scvx.jl:555-565
    ref = ptr_ref.discretize(mdl, pars, scale, x, u, p)
    ref.L = np.nan; ref.J_nl = np.nan
    eta = pars.eta_init
    hist = []
    status = "SCP_SOLVED"
    k = 1
    while True:
        sub = ptr_ref.solve_subproblem(mdl, pars, scale, ref, pp, ipm_opts, algo="scvx", eta=eta)
        sol = ptr_ref.discretize(mdl, pars, scale, sub["x"], sub["u"], sub["p"])
        sol.L = sub["L"]; sol.J_nl = np.nan
        if sub["status"] not in (ipm.OPTIMAL, ipm.ALMOST_OPTIMAL):          # unsafe_solution, scp.jl:965-980
            status = "SCP_FAILED (%s)" % sub["status"]
            hist.append(dict(sub=sub, sol=sol, ref=ref, eta=eta, stop=False)); break
        # ---- check_stopping_criterion!, scvx.jl:711-734 ----
        dev = ptr_ref.solution_deviation(scale, pars, ref, sol)
        J_ref = solution_cost(mdl, pars, ref, "nonlinear", pp)
        L_sol = solution_cost(mdl, pars, sol, "linear", pp)
        pre_improv = J_ref - L_sol
        pre_rel = pre_improv / abs(J_ref)
        stop = k > 1 and (sol.feas and (pre_rel <= pars.eps_rel or dev <= pars.eps_abs))
        rec = dict(sub=sub, sol=sol, ref=ref, eta=eta, stop=stop, deviation=dev, J_ref=J_ref, pre_improv=pre_improv)
        if stop:
            hist.append(rec); break
        # ---- update_trust_region!, scvx.jl:753-769 ----
        J_sol = solution_cost(mdl, pars, sol, "nonlinear", pp)
        act_improv = J_ref - J_sol
        rho = act_improv / pre_improv
        accept, eta_next, tag = update_rule(pars, rho, eta)
        rec.update(J_sol=J_sol, act_improv=act_improv, rho=rho, accept=accept, tr_update=tag, eta_next=eta_next)
        hist.append(rec)
        if verbose:
            print("k=%2d %s L=% .6e Lpen=%.3e J_nl=% .6e eta=%.3g rho=% .3f %s%s dev=%.2e feas=%s" % (
                k, sub["status"][:8], sub["L"], sub["L_pen"], J_sol, eta, rho, tag or "-", "" if accept else " (rejected)",
                dev, sol.feas))
        if accept:
            ref = sol
        eta = eta_next
        k += 1
        if k > pars.iter_max:
            break
    return status, hist
