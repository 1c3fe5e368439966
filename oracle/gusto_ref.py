"""CPU ORACLE (test infrastructure, NOT product code): literal restatement of the GuSTO algorithm of the reference
(`pen = :quad` and, since round 4, `pen = :softplus` through exponential cones, gusto.jl:996-1031).

Follows, line by line:
  parameters            src/solvers/gusto.jl:59-85
  subproblem            gusto.jl:218-287 (variables, kappa), 534-550 (cost = L + L_st + L_tr),
                        570-707 (original cost), 725-867 + 883-934 (soft penalties on the convex state set X and on
                        the linearised non-convex constraints s), 936-995 (soft_penalty, quadratic: u >= 0,
                        f + u - v <= 0, cost lambda v^2), 1056-1170 (trust region: norm cones + dx_lq + dp_lq <= eta + tr,
                        tr penalised); dynamics and boundary conditions are NOT relaxed (gusto.jl:452-454), U is hard
  solution costs        gusto.jl:391-407 (J, J_st nonlinear; J_tr = value(L_tr); L, L_st, L_aug)
  stopping criterion    gusto.jl:1203-1230
  trust-region update   gusto.jl:1245-1293 (model-error ratio rho from the cost error and the dynamics error),
                        1310-1427 (update rule incl. the kappa shrink)
  loop                  gusto.jl:425-502; initial guess projected by correct_convex! (:516-521, scp.jl:275-361)
The conic solves are oracle/ipm.py.  Restriction: s must not depend on u (GuSTO's s(t, k, x, p) signature).  The convex state set X
enters through the cone indicators of define_conic_constraint! (src/parser/problem.jl:686-806): NONPOS, SOC and LINF.  Parity status: unpinned (no golden data in the reference)."""
import numpy as np

from . import ipm
from . import ptr_ref
from . import scvx_ref
from .models import MODELS, linrange


class GuSTOParameters:
    def __init__(self, N, Nsub, iter_max, lam_init, lam_max, rho_0, rho_1, beta_sh, beta_gr, gamma_fail, eta_init, eta_lb,
                 eta_ub, mu, iter_mu, eps_abs, eps_rel, feas_tol, q_tr=np.inf, q_exit=np.inf, pen="quad", hom=500.0):
        assert q_tr in (1, 2, 4, np.inf) and q_exit >= 1      # gusto.jl:1078-1079
        assert pen in ("quad", "softplus")                   # gusto.jl:69-70 (pen, hom)
        self.pen, self.hom = pen, hom
        self.N, self.Nsub, self.iter_max = N, Nsub, iter_max
        self.lam_init, self.lam_max, self.rho_0, self.rho_1 = lam_init, lam_max, rho_0, rho_1
        self.beta_sh, self.beta_gr, self.gamma_fail = beta_sh, beta_gr, gamma_fail
        self.eta_init, self.eta_lb, self.eta_ub, self.mu, self.iter_mu = eta_init, eta_lb, eta_ub, mu, iter_mu
        self.eps_abs, self.eps_rel, self.feas_tol, self.q_tr, self.q_exit = eps_abs, eps_rel, feas_tol, q_tr, q_exit


def quadrotor_test_parameters(N=30, Nsub=15, iter_max=15):
    """test/examples/quadrotor/tests.jl:86-130."""
    return GuSTOParameters(N, Nsub, iter_max, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.9, beta_sh=2.0, beta_gr=2.0,
                           gamma_fail=5.0, eta_init=10.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=6, eps_abs=0.0,
                           eps_rel=0.0, feas_tol=1e-3)


def kappa(pars, it):
    return 1.0 if it < pars.iter_mu else pars.mu ** (1 + it - pars.iter_mu)      # gusto.jl:264


def _indicators(mdl, t, k, x, p):
    """numerical mode of define_conic_constraint! (src/parser/problem.jl:783-803): the cone indicators q of the convex
    state set X at node k -- q <= 0 iff the point is in the cone.  NONPOS: q = z (one per row); SOC / LINF: the scalar
    q = |z[1:]|_2 or |z[1:]|_inf minus z[0]."""
    out = []
    for kind, M, Mp, m0 in mdl.X(t, k):
        z = M @ x + (Mp @ p if mdl.np else 0.0) + m0
        if kind == "NONPOS":
            out.extend(z.tolist())
        elif kind == "SOC":
            out.append(float(np.linalg.norm(z[1:]) - z[0]))
        elif kind == "LINF":
            out.append(float(np.abs(z[1:]).max() - z[0]))
        else:
            raise NotImplementedError(kind)
    return out


def solve_subproblem(mdl, pars, scale, ref, pp, lam, eta, ipm_opts=None):
    N, nx, nu, np_ = pars.N, mdl.nx, mdl.nu, mdl.np
    t = linrange(0.0, 1.0, N)
    w = ptr_ref._trapz_weights(t)
    Sx, cx, Su, cu, Sp, cp = scale.Sx, scale.cx, scale.Su, scale.cu, scale.Sp, scale.cp
    P = ptr_ref._Prog()
    xh = [P.var(nx) for _ in range(N)]
    uh = [P.var(nu) for _ in range(N)]
    ph = P.var(np_)

    def phys(Mx=None, kx=None, Mu=None, ku=None, Mp=None, const=None):
        terms = []
        const = np.array(const, float).copy()
        if Mx is not None:
            Mx = np.atleast_2d(Mx); terms.append((xh[kx], Mx * Sx[None, :])); const += Mx @ cx
        if Mu is not None:
            Mu = np.atleast_2d(Mu); terms.append((uh[ku], Mu * Su[None, :])); const += Mu @ cu
        if Mp is not None and np_ > 0:
            Mp = np.atleast_2d(Mp); terms.append((ph, Mp * Sp[None, :])); const += Mp @ cp
        return terms, const
    # ---- dynamics, un-relaxed (gusto.jl:452; discretization.jl:458-467 without E v) ----
    for k in range(N - 1):
        t1, c1 = phys(Mx=np.eye(nx), kx=k + 1, const=np.zeros(nx))
        t2, c2 = phys(Mx=-ref.A[k], kx=k, Mu=-ref.Bm[k], ku=k, Mp=-ref.F[k] if np_ else None, const=-ref.r[k])
        t3, c3 = phys(Mu=-ref.Bp[k], ku=k + 1, const=np.zeros(nx))
        P.add_zero(t1 + t2 + t3, c1 + c2 + c3)
    # ---- U hard (scp.jl:717-734) ----
    for k in range(N):
        for kind, M, Mp, m0 in mdl.U(t[k], k + 1):
            terms, const = phys(Mu=M, ku=k, Mp=Mp, const=m0)
            (P.add_nonpos if kind == "NONPOS" else P.add_soc)(terms, const)
    # ---- boundary conditions, un-relaxed (gusto.jl:454; scp.jl:808-895) ----
    for xb, kx, g, H, K in ((ref.xd[0], 0, mdl.gic, mdl.H0, mdl.K0), (ref.xd[-1], N - 1, mdl.gtc, mdl.Hf, mdl.Kf)):
        gv, Hv, Kv = g(xb, ref.p, pp), H(xb, ref.p, pp), K(xb, ref.p, pp)
        l0 = gv - Hv @ xb - (Kv @ ref.p if np_ else 0.0)
        terms, const = phys(Mx=Hv, kx=kx, Mp=Kv if np_ else None, const=l0)
        P.add_zero(terms, const)

    # ---- soft penalties (gusto.jl:936-995, quadratic): u >= 0, f + u - v <= 0, cost lambda v^2 ----
    pen, hom = getattr(pars, "pen", "quad"), getattr(pars, "hom", 500.0)

    def soft(terms, const, weight):
        if pen == "softplus":
            # gusto.jl:996-1031: (-w, 1, u) in EXP, (hom f - w, 1, v) in EXP, u + v <= 1, cost lambda w / hom
            # (exp(-w) + exp(hom f - w) <= 1  <=>  w >= log(1 + exp(hom f)))
            uu, vv, ww = P.var(1), P.var(1), P.var(1)
            e0, e2 = np.array([[1.0], [0.0], [0.0]]), np.array([[0.0], [0.0], [1.0]])
            P.add_exp([(ww, -e0), (uu, e2)], np.array([0.0, 1.0, 0.0]))
            P.add_exp([(idx, hom * e0 @ np.atleast_2d(M)) for idx, M in terms] + [(ww, -e0), (vv, e2)],
                      np.array([hom * float(np.asarray(const).reshape(-1)[0]), 1.0, 0.0]))
            P.add_nonpos([(uu, np.ones((1, 1))), (vv, np.ones((1, 1)))], np.array([-1.0]))
            P.add_cost_lin(ww, weight / hom)
            return ww
        uu, vv = P.var(1), P.var(1)
        P.add_nonpos([(uu, -np.ones((1, 1)))], np.zeros(1))
        P.add_nonpos(terms + [(uu, np.ones((1, 1))), (vv, -np.ones((1, 1)))], const)
        P.add_cost_quad_diag(vv, weight)
        return vv
    v_st = [[] for _ in range(N)]
    for k in range(N):          # convex state constraints: cone indicators q (optimisation mode of define_conic_constraint!,
        for kind, M, Mp, m0 in mdl.X(t[k], k + 1):      # problem.jl:705-781), each softly penalised (gusto.jl:883-934)
            terms, const = phys(Mx=M, kx=k, Mp=Mp if np_ else None, const=m0)
            d = M.shape[0]
            if kind == "NONPOS":                          # z - q <= 0, one q per row
                q_ = P.var(d)
                P.add_nonpos(terms + [(q_, -np.eye(d))], const)
                qs = [q_[i:i + 1] for i in range(d)]
            else:                                         # [z[0] + q; z[1:]] in the cone, scalar q
                q_ = P.var(1)
                e0 = np.zeros((d, 1)); e0[0, 0] = 1.0
                if kind == "SOC":
                    P.add_soc(terms + [(q_, e0)], const)
                elif kind == "LINF":
                    Dm = np.vstack([np.hstack([-np.ones((d - 1, 1)), np.eye(d - 1)]), np.hstack([-np.ones((d - 1, 1)), -np.eye(d - 1)])])
                    P.add_nonpos([(idx, Dm @ Mt) for idx, Mt in terms] + [(q_, Dm @ e0)], Dm @ const)
                else:
                    raise NotImplementedError(kind)
                qs = [q_]
            for qi in qs:
                v_st[k].append(soft([(qi, np.ones((1, 1)))], np.zeros(1), lam * w[k]))
    for k in range(N):          # linearised non-convex constraints (gusto.jl:757-792)
        if mdl.ns == 0:
            break
        a = (t[k], k + 1, ref.xd[k], ref.ud[k], ref.p)
        s, C, G = mdl.s(*a), mdl.C(*a), mdl.G(*a)
        assert not np.any(mdl.D(*a)), "GuSTO: s must not depend on u"
        for i in range(mdl.ns):
            r = s[i] - C[i] @ ref.xd[k] - (G[i] @ ref.p if np_ else 0.0)
            terms, const = phys(Mx=C[i][None, :], kx=k, Mp=G[i][None, :] if np_ else None, const=np.array([r]))
            v_st[k].append(soft(terms, const, lam * w[k]))
    # ---- trust region (gusto.jl:1056-1170), q = Inf ----
    xh_ref = (ref.xd - cx) / Sx
    ph_ref = (ref.p - cp) / Sp if np_ else np.zeros(0)
    tr = P.var(N); dx_lq = P.var(N); dp_lq = P.var(1)
    q_tr = getattr(pars, "q_tr", np.inf)

    def add_norm(t_idx, idx, n_, const):
        """(t, expr) in the cone of the q_tr-norm: q2cone = {1: L1, 2: SOC, 4: SOC, Inf: LINF} (gusto.jl:1078-1079)"""
        if q_tr == np.inf:
            P.add_linf(t_idx, [(idx, np.eye(n_))], const)
        elif q_tr == 1:
            P.add_l1(t_idx, [(idx, np.eye(n_))], const)
        else:
            P.add_soc([(t_idx, np.vstack([np.ones((1, 1)), np.zeros((n_, 1))])), (idx, np.vstack([np.zeros((1, n_)), np.eye(n_)]))],
                      np.concatenate([[0.0], const]))
    if np_ > 0:
        add_norm(dp_lq, ph, np_, -ph_ref)
    else:
        P.add_nonpos([(dp_lq, -np.ones((1, 1)))], np.zeros(1))
    one = np.ones((1, 1))
    v_tr = []
    for k in range(N):
        add_norm(dx_lq[k:k + 1], xh[k], nx, -xh_ref[k])
        if q_tr == 4:       # gusto.jl:1107-1131: (w, dx_lq, dp_lq) in SOC, (w, eta + tr, 1) in GEOM -- the geometric-mean cone as a
            wv = P.var(1)   # second-order-cone solver receives it: w^2 <= eta + tr  <=>  (eta + tr + 1, 2 w, eta + tr - 1) in Q^3
            e = lambda i: np.eye(3)[:, i:i + 1]
            P.add_soc([(wv, e(0)), (dx_lq[k:k + 1], e(1)), (dp_lq, e(2))], np.zeros(3))
            P.add_soc([(tr[k:k + 1], np.array([[1.0], [0.0], [1.0]])), (wv, np.array([[0.0], [2.0], [0.0]]))],
                      np.array([float(eta) + 1.0, 0.0, float(eta) - 1.0]))
        else:
            P.add_nonpos([(dx_lq[k:k + 1], one), (dp_lq, one), (tr[k:k + 1], -one)], np.array([-float(eta)]))
        v_tr.append(soft([(tr[k:k + 1], one)], np.zeros(1), lam * w[k]))
    # ---- original cost (gusto.jl:570-663 with convex S, l, g) ----
    ct = mdl.cost_terms()
    cost_const = 0.0
    for k in range(N):
        P.add_cost_quad_diag(uh[k], w[k] * ct["Qu"] * Su * Su)
        P.add_cost_lin(uh[k], w[k] * (2 * ct["Qu"] * cu * Su + ct["lu"] * Su))
        P.add_cost_lin(xh[k], w[k] * ct["lx"] * Sx)
        cost_const += w[k] * (ct["Qu"] @ (cu * cu) + ct["lu"] @ cu + ct["lx"] @ cx)
    P.add_cost_lin(xh[N - 1], ct["tx"] * Sx)
    cost_const += ct["tx"] @ cx
    if np_ > 0:
        P.add_cost_lin(ph, ct["tp"] * Sp + 2 * ct["Qp"] * cp * Sp)
        P.add_cost_quad_diag(ph, ct["Qp"] * Sp * Sp)
        cost_const += ct["tp"] @ cp + ct["Qp"] @ (cp * cp)
    res = P.solve(**(ipm_opts or {}))
    z = res["x"]
    x = np.stack([Sx * z[i] + cx for i in xh]); u = np.stack([Su * z[i] + cu for i in uh])
    p = Sp * z[ph] + cp if np_ else np.zeros(0)
    L = scvx_ref.compute_original_cost(mdl, pars, x, u, p)
    if pen == "softplus":      # the penalty variable is w, its cost lambda w / hom (gusto.jl:1029)
        L_st = lam * sum(w[k] * sum(float(z[v][0]) / hom for v in v_st[k]) for k in range(N))
        L_tr = lam * sum(w[k] * float(z[v_tr[k]][0]) / hom for k in range(N))
    else:
        L_st = lam * sum(w[k] * sum(float(z[v][0]) ** 2 for v in v_st[k]) for k in range(N))
        L_tr = lam * sum(w[k] * float(z[v_tr[k]][0]) ** 2 for k in range(N))
    return dict(x=x, u=u, p=p, status=res["status"], ipm=res, L=L, L_st=L_st, L_tr=L_tr, L_aug=L + L_st + L_tr,
                pcost=res["pcost"] + cost_const, sizes=P.sizes)


def state_penalty_nonconvex(mdl, pars, x, p, lam):
    """state_penalty_cost(x, p, spbm, :nonconvex), gusto.jl:835-865."""
    t = linrange(0.0, 1.0, pars.N)
    pen = np.zeros(pars.N)
    h = penalty_fn(pars, lam)
    for k in range(pars.N):
        for f in _indicators(mdl, t[k], k + 1, x[k], p):
            pen[k] += h(f)
        if mdl.ns:
            s = mdl.s(t[k], k + 1, x[k], np.zeros(mdl.nu), p)
            pen[k] += sum(h(float(si)) for si in s)
    return float(ptr_ref._trapz(pen, t))


def penalty_fn(pars, lam):
    """numerical mode of soft_penalty (gusto.jl:966-1000): lambda max(0, f)^2, or lambda logsumexp([0, f]; t = hom) =
    lambda log(1 + exp(hom f)) / hom (src/utils/helper.jl logsumexp)"""
    if getattr(pars, "pen", "quad") == "softplus":
        hom = pars.hom
        return lambda f: lam * float(np.logaddexp(0.0, hom * f)) / hom
    return lambda f: lam * max(0.0, f) ** 2


def model_error(mdl, pars, ref, x, u, p):
    """dynamics part of update_trust_region!, gusto.jl:1269-1287: (dyn_error, dynamics_nrml)."""
    from . import oracle as orc
    t = linrange(0.0, 1.0, pars.N)
    df, dn = np.zeros(pars.N), np.zeros(pars.N)
    npd = getattr(mdl, "np_dyn", mdl.np)        # parameters the dynamics see (ptr_ref.discretize)
    for k in range(pars.N):
        f, A, B, F = orc.model_eval(mdl.name, mdl.par(), t[k], k + 1, ref.xd[k], ref.ud[k], ref.p[:npd])
        r = f - A @ ref.xd[k] - B @ ref.ud[k] - (F @ ref.p[:npd] if npd else 0.0)
        f_lin = A @ x[k] + B @ u[k] + (F @ p[:npd] if npd else 0.0) + r
        f_nl = orc.model_eval(mdl.name, mdl.par(), t[k], k + 1, x[k], u[k], p[:npd])[0]
        df[k] = np.linalg.norm(f_nl - f_lin); dn[k] = np.linalg.norm(f_lin)
    return float(ptr_ref._trapz(df, t)), float(ptr_ref._trapz(dn, t))


def update_rule(mdl, pars, scale, ref, sol, sub, rho, lam, eta, it):
    """update_rule!, gusto.jl:1310-1427 -> (accept, next_eta, next_lam, tags)."""
    N = pars.N
    t = linrange(0.0, 1.0, N)
    xh, xr = (sol.xd - scale.cx) / scale.Sx, (ref.xd - scale.cx) / scale.Sx
    q = getattr(pars, "q_tr", np.inf)
    wq = 2 if q == 4 else 1                                                       # gusto.jl:1180
    dp = np.linalg.norm((sol.p - ref.p) / scale.Sp, q) if mdl.np else 0.0
    tr = np.array([np.linalg.norm(xh[k] - xr[k], q) ** wq + dp ** wq - eta for k in range(N)])   # trust_region_cost(:nonconvex), :1172-1185
    trust_viol = bool(np.any(tr > 1e-3))
    feasible = True
    if not trust_viol:
        for k in range(N):
            if any(q > 1e-3 for q in _indicators(mdl, t[k], k + 1, sol.xd[k], sol.p)):
                feasible = False
            if mdl.ns and np.any(mdl.s(t[k], k + 1, sol.xd[k], np.zeros(mdl.nu), sol.p) > 1e-3):
                feasible = False
    if trust_viol:
        accept, eta_n, lam_n = False, eta, pars.gamma_fail * lam
    elif rho < pars.rho_1:
        accept = True
        eta_n = min(pars.eta_ub, pars.beta_gr * eta) if rho < pars.rho_0 else eta
        lam_n = pars.lam_init if feasible else pars.gamma_fail * lam
    else:
        accept, eta_n, lam_n = False, max(pars.eta_lb, eta / pars.beta_sh), lam
    kap = kappa(pars, it)
    if kap < 1:
        eta_n *= kap
    return accept, eta_n, lam_n, dict(trust_viol=trust_viol, feasible=feasible)


def gusto_solve(model, pars, pp=None, guess=None, ipm_opts=None, verbose=False):
    """`GuSTO.solve(pbm)` (gusto.jl:425-502) for one problem.  Returns (status, history)."""
    mdl = MODELS[model]() if isinstance(model, str) else model
    pp = mdl.nominal_pp() if pp is None else np.asarray(pp, float)
    scale = ptr_ref.Scaling(*mdl.bbox())
    x, u, p = mdl.guess(pars.N, pp) if guess is None else guess
    x, u, p = scvx_ref.correct_convex(mdl, pars, scale, x, u, p, ipm_opts)       # generate_initial_guess, :516-521
    ref = ptr_ref.discretize(mdl, pars, scale, x, u, p)
    ref.J_aug = np.nan
    lam, eta = pars.lam_init, pars.eta_init
    hist = []
    status = "SCP_SOLVED"
    for k in range(1, pars.iter_max + 1):
        sub = solve_subproblem(mdl, pars, scale, ref, pp, lam, eta, ipm_opts)
        sol = ptr_ref.discretize(mdl, pars, scale, sub["x"], sub["u"], sub["p"])
        if sub["status"] not in (ipm.OPTIMAL, ipm.ALMOST_OPTIMAL):
            status = "SCP_FAILED (%s)" % sub["status"]
            hist.append(dict(sub=sub, sol=sol, ref=ref, eta=eta, lam=lam, stop=False)); break
        J = sub["L"]                                                               # original_cost(:nonconvex) == convex cost here
        J_st = state_penalty_nonconvex(mdl, pars, sol.xd, sol.p, lam)
        sol.J_aug = J + J_st + sub["L_tr"]                                         # gusto.jl:399-402
        dev = ptr_ref.solution_deviation(scale, pars, ref, sol)
        dJ = abs(ref.J_aug - sol.J_aug) / abs(ref.J_aug)
        stop = k > 1 and ((sol.feas and (dJ <= pars.eps_rel or dev <= pars.eps_abs)) or lam > pars.lam_max)
        rec = dict(sub=sub, sol=sol, ref=ref, eta=eta, lam=lam, stop=stop, deviation=dev, J_aug=sol.J_aug, J_st=J_st)
        if stop:
            hist.append(rec); break
        cost_error = abs(sol.J_aug - sub["L_aug"])
        dyn_error, dyn_nrml = model_error(mdl, pars, ref, sol.xd, sol.ud, sol.p)
        rho = (cost_error + dyn_error) / (abs(sub["L_aug"]) + dyn_nrml)
        accept, eta_n, lam_n, tags = update_rule(mdl, pars, scale, ref, sol, sub, rho, lam, eta, k)
        rec.update(rho=rho, accept=accept, eta_next=eta_n, lam_next=lam_n, cost_error=cost_error, dyn_error=dyn_error, **tags)
        hist.append(rec)
        if verbose:
            print("k=%2d %s L=% .6e Lst=%.3e Ltr=%.3e J_aug=% .6e eta=%.3g lam=%.1e rho=%.4f %s dev=%.2e feas=%s" % (
                k, sub["status"][:8], sub["L"], sub["L_st"], sub["L_tr"], sol.J_aug, eta, lam, rho,
                "acc" if accept else "REJ", dev, sol.feas))
        if accept:
            ref = sol
        eta, lam = eta_n, lam_n
    return status, hist
