"""Generic subproblem pipeline and SCvx on the MI355X (-m gpu): discretize! -> linearise -> gather -> conic solve ->
read-out -> discretize! through the C ABI (scp_sub_*), against the oracle's literal restatements
(oracle/ptr_ref.py, oracle/scvx_ref.py) and against the structured fast path."""
import numpy as np
import pytest

from oracle import ptr_ref, scvx_ref
from oracle.models import MODELS

pytestmark = pytest.mark.gpu


def make_pbm(pkg, model, N, Nsub, B, **kw):
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=3, wvc=1e3, wtr=0.1, **kw)
    return traj, pars, pkg.PTR.create(pars, traj, batch_capacity=B)


@pytest.mark.parametrize("model,N,Nsub", [("quadrotor", 12, 8), ("rocket_landing", 10, 8), ("double_integrator", 10, 6)])
@pytest.mark.parametrize("q_tr", [np.inf, 1, 2])
def test_generic_ptr_subproblem_matches_oracle(pkg, model, N, Nsub, q_tr):
    mdl = MODELS[model]()
    traj, pars, pbm = make_pbm(pkg, model, N, Nsub, 3)
    scale = ptr_ref.Scaling(*mdl.bbox())
    opars = ptr_ref.PTRParameters(N, Nsub, 3, 1e3, 0.1, 0, 0, 1e-3, q_tr=q_tr)
    rng = np.random.default_rng(0)
    refs, pps = [], []
    for b in range(3):
        pp = mdl.nominal_pp() * (1 + 0.05 * rng.uniform(-1, 1, mdl.nominal_pp().size)) if b else mdl.nominal_pp()
        x, u, p = mdl.guess(N, pp)
        x = x + 0.02 * scale.Sx * rng.standard_normal(x.shape)
        refs.append((x, u, p)); pps.append(pp)
    T = pkg.subproblem.build_ptr(pkg.subproblem.ModelRows(traj.mdl), N, pbm.scale, 1e3, 0.1, q_tr)
    sub = pkg.generic.GenericSubproblem(pbm, T)
    g = sub.solve(np.stack([r[0] for r in refs]), np.stack([r[1] for r in refs]), np.stack([r[2] for r in refs]),
                  pp=np.stack(pps), want_conic=True)
    for b in range(3):
        ref = ptr_ref.discretize(mdl, opars, scale, *refs[b])
        o = ptr_ref.solve_subproblem(mdl, opars, scale, ref, pps[b])
        assert g["status"][b] in (0, 1)
        assert abs(g["pcost"][b] - o["J_aug"]) <= 2e-7 * max(1.0, abs(o["J_aug"]))
        assert np.abs((g["u"][b] - o["u"]) / scale.Su).max() < 5e-5
        J_vc = 1e3 * g["fun"][b, 0]; J_tr = 0.1 * g["fun"][b, 1]
        assert abs(J_vc - o["J_vc"]) <= 1e-6 * max(1.0, abs(o["J_vc"])) and abs(J_tr - o["J_tr"]) <= 1e-6 * max(1.0, abs(o["J_tr"]))
        # the new point was discretised on the device: defects match the oracle's discretize! of the oracle's solution
        sol = ptr_ref.discretize(mdl, opars, scale, g["x"][b], g["u"][b], g["p"][b])
        assert np.abs(g["defect"][b] - sol.defect).max() < 1e-9 * max(1.0, np.abs(sol.defect).max())
    if q_tr == np.inf:      # both product paths solve the same problem
        s2 = pkg.PTR.solve_subproblem_(pbm, np.stack([r[0] for r in refs]), np.stack([r[1] for r in refs]),
                                       np.stack([r[2] for r in refs]), pp=np.stack(pps))
        assert np.abs(s2["J_aug"] - g["pcost"]).max() <= 5e-6 * max(1.0, np.abs(g["pcost"]).max())
    sub.close(); pbm.close()


def test_correct_convex_on_device_matches_oracle(pkg):
    model, N, Nsub = "quadrotor", 12, 8
    mdl = MODELS[model]()
    traj, pars, pbm = make_pbm(pkg, model, N, Nsub, 2)
    scale = ptr_ref.Scaling(*mdl.bbox())
    opars = ptr_ref.PTRParameters(N, Nsub, 3, 1e3, 0.1, 0, 0, 1e-3)
    rng = np.random.default_rng(1)
    x, u, p = mdl.guess(N, mdl.nominal_pp())
    us = np.stack([u + 0.6 * scale.Su * rng.standard_normal(u.shape) for _ in range(2)])
    T = pkg.subproblem.build_correct_convex(pkg.subproblem.ModelRows(traj.mdl), N, pbm.scale)
    sub = pkg.generic.GenericSubproblem(pbm, T)
    g = sub.solve(np.stack([x, x]), us, np.stack([p, p]))
    for b in range(2):
        xo, uo, po = scvx_ref.correct_convex(mdl, opars, scale, x, us[b], p)
        assert g["status"][b] in (0, 1)
        assert np.abs((g["u"][b] - uo) / scale.Su).max() < 1e-5 and np.abs((g["x"][b] - xo) / scale.Sx).max() < 1e-5
    sub.close(); pbm.close()


def test_scvx_loop_matches_oracle_on_the_reference_config(pkg):
    """SCvx on the quadrotor with the reference's own test parameters (test/examples/quadrotor/tests.jl:32-75: N = 30,
    Nsub = 15, iter_max = 15, lambda = 30, rho = (0, 0.1, 0.7), beta = 2, eta in [1e-3, 10], eta_init = 1): the device
    loop follows the oracle's literal loop decision by decision."""
    N, Nsub, iters = 30, 15, 15
    op = scvx_ref.quadrotor_test_parameters(N, Nsub, iters)
    mdl = MODELS["quadrotor"]()
    rng = np.random.default_rng(3)
    pps = [mdl.nominal_pp()]
    for _ in range(2):
        q = mdl.nominal_pp().copy(); q[6:9] *= 1 + 0.1 * rng.uniform(-1, 1, 3); pps.append(q)
    traj = pkg.TrajectoryProblem("quadrotor")
    pars = pkg.SCvx.Parameters(N=N, Nsub=Nsub, iter_max=iters, lam=op.lam, rho_0=op.rho_0, rho_1=op.rho_1, rho_2=op.rho_2,
                               beta_sh=op.beta_sh, beta_gr=op.beta_gr, eta_init=op.eta_init, eta_lb=op.eta_lb, eta_ub=op.eta_ub,
                               eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
    pbm = pkg.SCvx.create(pars, traj, batch_capacity=3)
    sol, hist = pkg.SCvx.solve(pbm, np.stack(pps))
    scale = ptr_ref.Scaling(*mdl.bbox())
    for b in range(3):
        st, oh = scvx_ref.scvx_solve("quadrotor", op, pp=pps[b])
        assert st == "SCP_SOLVED" and sol.status[b] == "SCP_SOLVED"
        assert sol.iterations[b] == len(oh)
        forked = False
        for k, rec in enumerate(oh):
            assert hist["eta"][k, b] == pytest.approx(rec["eta"], rel=1e-12)          # same trust-region sequence
            if bool(hist["accepted"][k, b]) != bool(rec["accept"]):
                # a decision may differ only where rho = dJ / dL is not determined within the solvers' accuracy: (a) the oracle loop has
                # stopped moving (its solution cost repeats to 10 digits: 0 / 0, e.g. instance 2 from iteration 12 on -- both outcomes
                # shrink the radius the same way), or (b) the oracle's rho sits within 0.03 of the acceptance threshold rho_0 and the
                # device's rho agrees with it to 0.02; the comparison of this instance ends there
                still = k > 0 and abs(rec["J_sol"] - oh[k - 1]["J_sol"]) <= 1e-10 * max(1.0, abs(rec["J_sol"]))
                near = abs(rec["rho"] - op.rho_0) <= 0.03 and abs(hist["rho"][k, b] - rec["rho"]) <= 0.02
                assert still or near, (b, k, float(hist["rho"][k, b]), rec["rho"], float(hist["J_sol"][k, b]), rec["J_sol"])
                forked = True
                break
            # costs of the iterates: the subproblem optimum is unique in cost, not in trajectory (flat faces of the L1 /
            # Linf epigraphs), and the nonlinear cost amplifies the difference by lambda = 30: 1e-4 relative
            assert abs(hist["L"][k, b] - rec["sub"]["L"]) <= 2e-5 * max(1.0, abs(rec["sub"]["L"]))
            assert abs(hist["J_sol"][k, b] - rec["J_sol"]) <= 1e-4 * max(1.0, abs(rec["J_sol"]))
        if forked:
            continue
        fin = oh[-1]["sol"]
        assert np.abs((sol.xd[b] - fin.xd) / scale.Sx).max() < 2e-4
        assert np.abs((sol.ud[b] - fin.ud) / scale.Su).max() < 2e-4
        assert abs(sol.p[b, 0] - fin.p[0]) < 2e-4 * scale.Sp[0]
        assert sol.feas[b] == fin.feas
    pbm.close()


def test_scvx_stopping_and_batch_independence(pkg):
    """with a stopping tolerance the problems stop at their own iteration; a batch member does not depend on its peers"""
    N, Nsub = 16, 10
    traj = pkg.TrajectoryProblem("quadrotor")
    mdl = traj.mdl
    mk = lambda: pkg.SCvx.Parameters(N=N, Nsub=Nsub, iter_max=12, lam=30.0, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0,
                                     beta_gr=2.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, eps_abs=1e-4, eps_rel=1e-3)
    rng = np.random.default_rng(5)
    pps = np.stack([mdl.nominal_pp() * (1 + 0.03 * rng.uniform(-1, 1, 12)) for _ in range(70)])
    pbm = pkg.SCvx.create(mk(), traj, batch_capacity=70)
    sol, hist = pkg.SCvx.solve(pbm, pps)
    pbm.close()
    assert all(s == "SCP_SOLVED" for s in sol.status)
    stopped = sol.iterations < 12
    assert stopped.any() and sol.feas[stopped].all()          # the stopping criterion requires feasibility (scvx.jl:724-727)
    pb1 = pkg.SCvx.create(mk(), traj, batch_capacity=1)
    for b in (0, 37, 69):
        s1, h1 = pkg.SCvx.solve(pb1, pps[b:b + 1])
        assert s1.iterations[0] == sol.iterations[b]
        assert np.abs(s1.xd[0] - sol.xd[b]).max() < 1e-9
    pb1.close()


@pytest.mark.parametrize("model", ["double_integrator", "quadrotor"])
def test_impulse_discretize_matches_oracle(pkg, orc, model):
    """IMPULSE discretisation (discretization.jl:186-193, 304-340, 384-390) on the device vs the C restatement."""
    N, Nsub, B = 9, 7, 5
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=1, disc_method=pkg.IMPULSE)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    rng = np.random.default_rng(2)
    xs, us, ps = [], [], []
    for b in range(B):
        x, u, p = traj.guess(N, traj.mdl.nominal_pp())
        xs.append(x + 0.1 * rng.standard_normal(x.shape)); us.append(u + 0.3 * rng.standard_normal(u.shape)); ps.append(p)
    ref = pkg.SubproblemSolutionBatch(np.stack(xs), np.stack(us), np.stack(ps).reshape(B, -1), pbm)
    pkg.discretize_(ref, pbm)
    o = orc.discretize(model, orc.default_params(model), N, Nsub, ref.xd, ref.ud, ref.p, pbm.scale.iSx, pars.feas_tol,
                       method="impulse")
    for nm, got in (("A", ref.dyn.A), ("Bm", ref.dyn.B[0]), ("F", ref.dyn.F), ("r", ref.dyn.r), ("E", ref.dyn.E),
                    ("defect", ref.defect)):
        err = np.max(np.abs(got - o[nm])) / max(1.0, np.max(np.abs(o[nm]))) if o[nm].size else 0.0
        assert err < 1e-10, (nm, err)
    assert np.abs(ref.dyn.B[1]).max() == 0.0          # IMPULSE has a single input matrix (DLTV.B, discretization.jl:31)
    assert (ref.feas == o["feas"]).all()
    # the discrete model reproduces the impulse-then-coast propagation at the reference: x_{k+1} - (A x + B u + F p + r) = defect
    b, k = 1, 3
    pred = ref.dyn.A[b, k].T @ ref.xd[b, k] + ref.dyn.B[0][b, k].T @ ref.ud[b, k] + ref.dyn.r[b, k]
    if pbm.npF:
        pred = pred + ref.dyn.F[b, k].T @ ref.p[b]
    assert np.abs((ref.xd[b, k + 1] - pred) - ref.defect[b, k]).max() < 1e-10
    pbm.close()


def test_ptr_loop_with_two_norm_trust_region_matches_oracle(pkg):
    """PTR with q_tr = 2 (SOC trust regions, ptr.jl:582-599) through the generic conic path: same loop as the oracle's."""
    model, N, Nsub, iters = "quadrotor", 12, 8, 6
    mdl = MODELS[model]()
    opars = ptr_ref.PTRParameters(N, Nsub, iters, 1e3, 0.1, 0, 0, 1e-3, q_tr=2)
    st, oh = ptr_ref.ptr_solve(model, opars)
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0, q_tr=2.0)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=2)
    sol, hist = pkg.PTR.solve(pbm, np.stack([mdl.nominal_pp(), mdl.nominal_pp()]))
    scale = ptr_ref.Scaling(*mdl.bbox())
    assert sol.status[0] == "SCP_SOLVED" and st == "SCP_SOLVED"
    for k, rec in enumerate(oh):
        assert abs(hist.J_aug[k, 0] - rec["sub"]["J_aug"]) <= 2e-5 * max(1.0, abs(rec["sub"]["J_aug"]))
    fin = oh[-1]["sol"]
    assert np.abs((sol.xd[0] - fin.xd) / scale.Sx).max() < 2e-4 and np.abs((sol.ud[0] - fin.ud) / scale.Su).max() < 2e-4
    assert np.array_equal(sol.xd[0], sol.xd[1])
    pbm.close()


def test_compute_scaling_on_device_matches_the_analytic_boxes(pkg):
    """compute_scaling (scp.jl:376-517) as two batched conic solves: the LP optima over the quadrotor's U set are the
    analytic boxes the model advises; unbounded directions (no X set) keep the default [0, 1] box via the
    DUAL_INFEASIBLE certificate (scp.jl:470-477)."""
    mr = pkg.subproblem.ModelRows(pkg.REGISTRY["quadrotor"]())
    from scptoolbox_jl_amd import scaling
    sc, info = scaling.compute_scaling(mr, 10)
    adv = pkg.REGISTRY["quadrotor"]().scale_advice()
    np.testing.assert_allclose(info["bbox"]["u"], adv[1], atol=1e-6)
    np.testing.assert_allclose(info["bbox"]["p"], adv[2], atol=1e-6)
    np.testing.assert_allclose(info["bbox"]["x"], adv[0], atol=0)
    assert set(info["status"].values()) <= {0, 1, 5}
    mr = pkg.subproblem.ModelRows(pkg.REGISTRY["rocket_landing"]())
    sc, info = scaling.compute_scaling(mr, 10)
    v_max = 500 * 1e3 / 3600
    np.testing.assert_allclose(info["bbox"]["x"][3:6], [[-v_max, v_max]] * 3, rtol=1e-7)
    np.testing.assert_allclose(info["bbox"]["p"], [[40.0, 120.0]], atol=1e-6)


@pytest.mark.parametrize("q_exit", [1.0, 2.0])
def test_device_resident_generic_ptr_loop_with_q_exit_norms(pkg, q_exit):
    """PTR through the generic path RESIDENT on the device (scp_ptr_generic_*): the cost split, the stopping rule with the
    deviation in the q_exit norm (scp.jl:909-931: 1, 2) and ref = sol -- same iteration count, deviations and costs as the
    oracle's literal loop (ptr.jl:467-524), stopping on the absolute tolerance."""
    model, N, Nsub, iters = "quadrotor", 12, 8, 12
    mdl = MODELS[model]()
    opars = ptr_ref.PTRParameters(N, Nsub, iters, 1e3, 0.1, 2e-3, 0.0, 1e-3, q_tr=1, q_exit=q_exit)
    st, oh = ptr_ref.ptr_solve(model, opars)
    assert st == "SCP_SOLVED" and 2 < len(oh) < iters            # stopped by the deviation test
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=2e-3, eps_rel=0.0, q_tr=1.0, q_exit=q_exit,
                              solver_opts={"maxit": 80})
    pbm = pkg.PTR.create(pars, traj, batch_capacity=2)
    pp2 = mdl.nominal_pp().copy(); pp2[6:9] *= 1.02
    sol, hist = pkg.PTR.solve(pbm, np.stack([mdl.nominal_pp(), pp2]))
    assert sol.status == ["SCP_SOLVED", "SCP_SOLVED"] and sol.iterations[0] == len(oh)
    for k, rec in enumerate(oh):
        assert abs(hist.J_aug[k, 0] - rec["sub"]["J_aug"]) <= 2e-5 * max(1.0, abs(rec["sub"]["J_aug"]))
        assert abs(hist.deviation[k, 0] - rec["deviation"]) <= 1e-3 * max(rec["deviation"], 1e-3) + 2e-5
    assert hist.active[:, 0].sum() == len(oh) and not hist.active[len(oh):, 0].any()
    scale = ptr_ref.Scaling(*mdl.bbox())
    fin = oh[-1]["sol"]
    assert np.abs((sol.xd[0] - fin.xd) / scale.Sx).max() < 2e-4
    # options: ECOS names are mapped, options of the structured solver are refused on this path instead of being dropped
    pbm.pars.solver_opts = {"warm": 1}
    with pytest.raises(pkg._lib.ScpError):
        pkg.PTR.solve(pbm, mdl.nominal_pp()[None])
    pbm.close()


@pytest.mark.parametrize("model", ["double_integrator", "quadrotor"])
def test_impulse_propagate_matches_oracle(pkg, orc, model):
    """propagate of an IMPULSE solution (discretization.jl:542-560): every interval restarts from its node with the model's
    impulse response applied and coasts with idle inputs; 1 + (N-1) ceil(res/(N-1)) samples, the first time of every interval
    shifted by sqrt(eps); uc is the impulse trajectory (diracinterp)."""
    N, B, res = 7, 3, 50
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=6, iter_max=1, disc_method=pkg.IMPULSE)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    rng = np.random.default_rng(4)
    xs, us, ps = [], [], []
    for b in range(B):
        x, u, p = traj.guess(N, traj.mdl.nominal_pp())
        xs.append(x + 0.1 * rng.standard_normal(x.shape)); us.append(u + 0.3 * rng.standard_normal(u.shape)); ps.append(p)
    ref = pkg.SubproblemSolutionBatch(np.stack(xs), np.stack(us), np.stack(ps).reshape(B, -1), pbm)
    tc, xc = pkg.propagate(ref, pbm, res=res)
    sub = -(-res // (N - 1))
    assert tc.size == 1 + (N - 1) * sub and xc.shape == (B, tc.size, pbm.nx)
    for b in range(B):
        to, xo = orc.propagate_impulse(model, orc.default_params(model), N, ref.xd[b], ref.ud[b], ref.p[b], res=res)
        assert np.array_equal(to, tc)
        assert np.abs(xc[b] - xo).max() <= 1e-10 * max(1.0, np.abs(xo).max())
    ref.status = ["SCP_SOLVED"] * B
    _, _, uc = pkg.continuous_time(ref, pbm)
    assert isinstance(uc, pkg.ImpulseTrajectory)
    assert np.array_equal(uc.sample(pbm.t_grid[2]), ref.ud[:, 2]) and not uc.sample(0.5 * (pbm.t_grid[2] + pbm.t_grid[3])).any()
    pbm.close()


@pytest.mark.parametrize("q_tr", [2, 4])
def test_scvx_loop_other_trust_region_norms(pkg, q_tr):
    """SCvx with q_tr = 2 and q_tr = 4 (scvx.jl:593-675; q = 4: SOC + GEOM cones -- dx_lq^2 + du_lq^2 + dp_lq^2 <= eta) on the
    device against the oracle's literal loop: same radii and accept / reject decisions, same costs."""
    N, Nsub, iters = 16, 10, 5
    op = scvx_ref.quadrotor_test_parameters(N, Nsub, iters)
    op.q_tr = q_tr
    mdl = MODELS["quadrotor"]()
    traj = pkg.TrajectoryProblem("quadrotor")
    pars = pkg.SCvx.Parameters(N=N, Nsub=Nsub, iter_max=iters, lam=op.lam, rho_0=op.rho_0, rho_1=op.rho_1, rho_2=op.rho_2,
                               beta_sh=op.beta_sh, beta_gr=op.beta_gr, eta_init=op.eta_init, eta_lb=op.eta_lb, eta_ub=op.eta_ub,
                               eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3, q_tr=q_tr)
    pbm = pkg.SCvx.create(pars, traj, batch_capacity=1)
    sol, hist = pkg.SCvx.solve(pbm, mdl.nominal_pp()[None])
    pbm.close()
    st, oh = scvx_ref.scvx_solve("quadrotor", op, pp=mdl.nominal_pp())
    assert st == "SCP_SOLVED" and sol.status[0] == "SCP_SOLVED" and sol.iterations[0] == len(oh)
    same = 0
    for k, rec in enumerate(oh):
        if k > 0 and abs(hist["eta"][k, 0] - rec["eta"]) > 1e-12 * rec["eta"]:
            # The nonlinear cost of a solution sees WHICH minimiser of a flat optimal face the solver returned, amplified by
            # lambda = 30 (measured with q_tr = 2: 6e-3 relative at iteration 3), and with it the ratio rho: once it falls on the
            # other side of a threshold of the update rule the two loops carry different radii and are no longer comparable step by step
            break
        same += 1
        assert hist["eta"][k, 0] == pytest.approx(rec["eta"], rel=1e-12)
        assert bool(hist["accepted"][k, 0]) == bool(rec["accept"])
        assert abs(hist["L"][k, 0] - rec["sub"]["L"]) <= (5e-5 if k == 0 else 5e-3) * max(1.0, abs(rec["sub"]["L"]))
        assert abs(hist["J_sol"][k, 0] - rec["J_sol"]) <= (2e-4 if k == 0 else 2e-2) * max(1.0, abs(rec["J_sol"]))
    assert same >= 2       # the first subproblem is the same program for both solvers; measured: 2-3 (q_tr = 2), 5 (q_tr = 4) of 5
    if same == iters:       # loops on different radii after a threshold decision are not comparable at a fixed iteration count
        assert abs(hist["L"][iters - 1, 0] - oh[-1]["sub"]["L"]) <= 5e-3 * max(1.0, abs(oh[-1]["sub"]["L"]))


def test_gusto_loop_with_the_four_norm_trust_region(pkg):
    """GuSTO with q_tr = 4 (gusto.jl:1107-1131: dx_lq^2 + dp_lq^2 <= eta + tr through SOC + GEOM cones; trust_region_cost(:nonconvex)
    with squared norms, :1172-1185) on the device against the oracle's literal loop: same (eta, lambda) sequence and decisions.
    Two scenarios: eta_init = 50 (steps accepted), and the reference's eta_init = 10, where the squared 4-norm of the first step
    exceeds the radius -- trust-region violation, rejection, lambda x gamma_fail (:1349-1353) -- in both loops (one iteration: the
    NEXT subproblem, lambda = 5e4 about a reference 3 radii away, is one the product's solver ends NUMERICAL_ERROR on while the
    oracle's pivoting solver still solves it -- reproduced on the host build)."""
    from oracle import gusto_ref
    N, Nsub = 16, 10
    mdl = MODELS["quadrotor"]()
    traj = pkg.TrajectoryProblem("quadrotor")
    for eta_init, eta_ub, iters in ((50.0, 100.0, 6), (10.0, 10.0, 1)):
        op = gusto_ref.quadrotor_test_parameters(N, Nsub, iters)
        op.q_tr, op.eta_init, op.eta_ub = 4, eta_init, eta_ub
        gp = pkg.GuSTO.Parameters(N=N, Nsub=Nsub, iter_max=iters, lam_init=op.lam_init, lam_max=op.lam_max, rho_0=op.rho_0,
                                  rho_1=op.rho_1, beta_sh=op.beta_sh, beta_gr=op.beta_gr, gamma_fail=op.gamma_fail, eta_init=op.eta_init,
                                  eta_lb=op.eta_lb, eta_ub=op.eta_ub, mu=op.mu, iter_mu=op.iter_mu, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3,
                                  q_tr=4)
        pbm = pkg.GuSTO.create(gp, traj, batch_capacity=1)
        sol, hist = pkg.GuSTO.solve(pbm, mdl.nominal_pp()[None])
        pbm.close()
        # the oracle's solver with its objective normalised, like the product's (lambda-weighted costs of 1e7)
        st, oh = gusto_ref.gusto_solve("quadrotor", op, pp=mdl.nominal_pp(), ipm_opts=dict(normalise_objective=True))
        assert st.split()[0] == sol.status[0] == "SCP_SOLVED" and sol.iterations[0] == len(oh)
        for k, rec in enumerate(oh):
            assert hist["eta"][k, 0] == pytest.approx(rec["eta"], rel=1e-12) and hist["lam"][k, 0] == pytest.approx(rec["lam"], rel=1e-12)
            la = hist["L"][k, 0] + hist["L_st"][k, 0] + hist["L_tr"][k, 0]
            assert abs(la - rec["sub"]["L_aug"]) <= (2e-5 if k == 0 else 2e-2) * max(1.0, abs(rec["sub"]["L_aug"]))
            if "accept" in rec:
                assert bool(hist["accepted"][k, 0]) == bool(rec["accept"])
                assert bool(int(hist["flags"][k, 0]) & pkg.GuSTO.FLAG_TRUST_VIOLATED) == bool(rec["trust_viol"])
        if eta_init == 10.0:
            assert not oh[0]["accept"] and oh[0]["trust_viol"]
