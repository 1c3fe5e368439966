"""6-DoF free-flyer on the MI355X (-m gpu): discretize! with a 13-dimensional state-dependent Jacobian and the first real
integration action (quaternion renormalisation, freeflyer/definition.jl:69-82), propagate, the guess, and the SUBPROBLEM side
with the reference's N-dependent parameter vector p = [t_f; delta(6, N)] (one global + six node parameters,
csrc/models/freeflyer.hpp): SCvx and GuSTO (quadratic penalty on the cone indicators of X) at the reference's own test
parameters (freeflyer/tests.jl:25-140) against the oracle's literal loops and their committed golden runs."""
import os

import numpy as np
import pytest

from oracle.models import MODELS

pytestmark = pytest.mark.gpu


def _batch(pkg, N, B, seed=0):
    traj = pkg.TrajectoryProblem("freeflyer")
    rng = np.random.default_rng(seed)
    x, u, p = traj.guess(N, traj.mdl.nominal_pp())
    xs = np.stack([x + np.concatenate([0.05 * rng.standard_normal((N, 6)), 0.02 * rng.standard_normal((N, 4)),
                                       2e-3 * rng.standard_normal((N, 3))], axis=1) for _ in range(B)])
    xs[:, :, 6:10] /= np.linalg.norm(xs[:, :, 6:10], axis=2, keepdims=True)
    us = np.stack([u + np.concatenate([5e-3 * rng.standard_normal((N, 3)), 3e-5 * rng.standard_normal((N, 3))], axis=1)
                   for _ in range(B)])
    ps = np.stack([p * (1 + 0.1 * rng.uniform(-1, 1)) for _ in range(B)])
    return traj, xs, us, ps


@pytest.mark.parametrize("N,Nsub", [(50, 15), (12, 40)])
def test_discretize_matches_oracle(pkg, orc, N, Nsub):
    """reference test sizes (freeflyer/tests.jl:29-36: N = 50, Nsub = 15), 1e-10 relative"""
    B = 3
    traj, xs, us, ps = _batch(pkg, N, B)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=1, feas_tol=1e-3)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    assert pbm.info.has_subproblem == 1 and pbm.info.structured == 0 and (pbm.nx, pbm.nu, pbm.np) == (13, 6, 1 + 6 * N)
    ref = pkg.SubproblemSolutionBatch(xs, us, ps, pbm)
    pkg.discretize_(ref, pbm)
    # the dynamics see the global parameter t_f only (the oracle's C restatement carries np = 1)
    o = orc.discretize("freeflyer", orc.default_params("freeflyer"), N, Nsub, xs, us, ps[:, :1].copy(), pbm.scale.iSx, pars.feas_tol)
    for nm, got, want in (("A", ref.dyn.A, o["A"]), ("Bm", ref.dyn.B[0], o["Bm"]), ("Bp", ref.dyn.B[1], o["Bp"]),
                          ("F", ref.dyn.F, o["F"]), ("r", ref.dyn.r, o["r"]), ("E", ref.dyn.E, o["E"]),
                          ("defect", ref.defect, o["defect"])):
        err = np.max(np.abs(got - want)) / max(1.0, np.max(np.abs(want)))
        assert err < 1e-10, (nm, err)
    assert np.array_equal(ref.feas, o["feas"].astype(bool))
    # the action ran: the propagated quaternion is of unit norm
    prop = xs[:, 1:] - ref.defect
    assert np.abs(np.linalg.norm(prop[:, :, 6:10], axis=2) - 1.0).max() < 1e-12
    pbm.close()


def test_discretize_at_config_size_uses_both_forms_per_problem(pkg, orc):
    """BASELINE configs[4] size (N = 200, Nsub = 15): problems whose physical RK4 step t_f h is below the model's bound are
    discretised by the variational kernel K1x (no Phi^-1), the others by the reference formulation K1 -- chosen per problem on
    the device (disc_split_kernel).  Both must meet the reference formulation (C oracle) to 1e-10; the batch spans
    t_f = 70 ... 200 s around the switch at 0.047 s / 3.6e-4 = 131 s."""
    N, Nsub, B = 200, 15, 6
    traj, xs, us, ps = _batch(pkg, N, B, seed=5)
    ps[:, 0] = [70.0, 110.0, 129.0, 133.0, 160.0, 200.0]
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=1, feas_tol=1e-3)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    ref = pkg.SubproblemSolutionBatch(xs, us, ps, pbm)
    pkg.discretize_(ref, pbm)
    o = orc.discretize("freeflyer", orc.default_params("freeflyer"), N, Nsub, xs, us, ps[:, :1].copy(), pbm.scale.iSx, pars.feas_tol)
    worst = {}
    for nm, got, want in (("A", ref.dyn.A, o["A"]), ("Bm", ref.dyn.B[0], o["Bm"]), ("Bp", ref.dyn.B[1], o["Bp"]),
                          ("F", ref.dyn.F, o["F"]), ("r", ref.dyn.r, o["r"]), ("E", ref.dyn.E, o["E"]),
                          ("defect", ref.defect, o["defect"])):
        for b in range(B):
            err = np.max(np.abs(got[b] - want[b])) / max(1.0, np.max(np.abs(want[b])))
            worst[(nm, b)] = err
            assert err < 1e-10, (nm, b, ps[b, 0], err)
    # the variational problems are recognisable by their truncation-level (not round-off-level) distance in B
    assert max(worst[("Bm", b)] for b in (0, 1, 2)) > 1e-13 and max(worst[("Bm", b)] for b in (3, 4, 5)) < 1e-12
    assert np.array_equal(ref.feas, o["feas"].astype(bool))
    pbm.close()


def test_propagate_matches_oracle(pkg, orc):
    N, B, res = 20, 2, 101
    traj, xs, us, ps = _batch(pkg, N, B, seed=3)
    pbm = pkg.PTR.create(pkg.PTR.Parameters(N=N, Nsub=5, iter_max=1), traj, batch_capacity=B)
    ref = pkg.SubproblemSolutionBatch(xs, us, ps, pbm)
    tc, xc = pkg.propagate(ref, pbm, res=res)
    for b in range(B):
        to, xo = orc.propagate("freeflyer", orc.default_params("freeflyer"), N, xs[b], us[b], ps[b][:1].copy(), res=res)
        assert np.abs(xc[b] - xo).max() / max(1.0, np.abs(xo).max()) < 1e-10
    assert np.abs(np.linalg.norm(xc[:, :, 6:10], axis=2) - 1.0).max() < 1e-12
    pbm.close()


def test_device_guess_matches_the_oracle_restatement(pkg):
    """traj.guess on the device (scp_guess_batch_host): axis-by-axis path at constant speed, SLERP attitude, constant body
    rate (freeflyer/definition.jl:84-186) for a Monte-Carlo batch of boundary conditions, against oracle/models.py; and the
    straight-line guesses of two structured models against the host mirrors."""
    rng = np.random.default_rng(2)
    N, B = 50, 4
    om = MODELS["freeflyer"](N)
    pps = []
    for b in range(B):
        pp = om.nominal_pp().copy()
        pp[0:3] += 0.3 * rng.standard_normal(3); pp[13:16] += 0.3 * rng.standard_normal(3)
        for o in (6, 19):
            pp[o:o + 4] += 0.2 * rng.standard_normal(4); pp[o:o + 4] /= np.linalg.norm(pp[o:o + 4])
        pps.append(pp)
    pps[1][13] = pps[1][0] - 1.0                      # a leg in the negative direction
    traj = pkg.TrajectoryProblem("freeflyer")
    pbm = pkg.PTR.create(pkg.PTR.Parameters(N=N, Nsub=5, iter_max=1), traj, batch_capacity=B)
    xd, ud, p = pkg.device_guess(pbm, np.stack(pps))
    pbm.close()
    for b in range(B):
        xo, uo, po = om.guess(N, pps[b])
        assert np.abs(xd[b] - xo).max() < 1e-12 and not ud[b].any() and p[b, 0] == po[0]
        assert p.shape[1] == 1 + 6 * N and np.abs(p[b] - po).max() < 1e-12        # the room slacks of the guess (:166-172)
    # (Starship's guess is no straight line: bang-bang flip + descent programs, tests/test_starship_gpu.py)
    for model in ("quadrotor", "rocket_landing"):
        traj = pkg.TrajectoryProblem(model)
        pbm = pkg.PTR.create(pkg.PTR.Parameters(N=20, Nsub=5, iter_max=1), traj, batch_capacity=2)
        pp = np.stack([traj.mdl.nominal_pp(), traj.mdl.nominal_pp() * 1.05])
        xd, ud, p = pkg.device_guess(pbm, pp)
        for b in range(2):
            xh, uh, ph = traj.guess(20, pp[b])
            assert np.abs(xd[b] - xh).max() < 1e-12 * max(1.0, np.abs(xh).max()) and np.abs(ud[b] - uh).max() < 1e-9 * max(1.0, np.abs(uh).max())
            assert np.abs(p[b] - ph).max() < 1e-12 * max(1.0, np.abs(ph).max())
        pbm.close()


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _scvx_pars(pkg, N, Nsub, iter_max):
    # test/examples/freeflyer/tests.jl:25-80
    return pkg.SCvx.Parameters(N=N, Nsub=Nsub, iter_max=iter_max, lam=1e3, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0,
                               eta_init=1.0, eta_lb=1e-6, eta_ub=10.0, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)


def _gusto_pars(pkg, N, Nsub, iter_max):
    # test/examples/freeflyer/tests.jl:84-140 (pen = :quad)
    return pkg.GuSTO.Parameters(N=N, Nsub=Nsub, iter_max=iter_max, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.5, beta_sh=2.0,
                                beta_gr=2.0, gamma_fail=5.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=16,
                                eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)


def test_subproblem_of_the_compiled_model_equals_the_oracle_program(pkg):
    """one SCvx subproblem about a perturbed guess on the device (linearise with the compact parameter Jacobians, gather,
    conic solve, read-out of all 1 + 6 N parameters) against the oracle's literal program"""
    from oracle import ptr_ref, scvx_ref
    N, Nsub, B = 10, 8, 2
    mdl = MODELS["freeflyer"](N)
    scale = ptr_ref.Scaling(*mdl.bbox())
    op = scvx_ref.SCvxParameters(N, Nsub, 3, lam=1e3, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                 eta_lb=1e-6, eta_ub=10.0, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
    traj = pkg.TrajectoryProblem("freeflyer")
    pbm = pkg.SCvx.create(_scvx_pars(pkg, N, Nsub, 3), traj, batch_capacity=B)
    assert np.allclose(pbm.scale.Sp, scale.Sp) and np.allclose(pbm.scale.Sx, scale.Sx)
    pp = mdl.nominal_pp()
    rng = np.random.default_rng(0)
    xs, us, ps = [], [], []
    for b in range(B):
        x, u, p = mdl.guess(N, pp)
        x = x + 0.02 * scale.Sx * rng.standard_normal(x.shape); x[:, 6:10] /= np.linalg.norm(x[:, 6:10], axis=1, keepdims=True)
        xs.append(x); us.append(u); ps.append(p)
    for eta in (1.0, 0.1):
        g = pbm.sub.solve(np.stack(xs), np.stack(us), np.stack(ps), pp=np.tile(pp, (B, 1)), scal=np.full((B, 1), eta))
        for b in range(B):
            ref = ptr_ref.discretize(mdl, op, scale, xs[b], us[b], ps[b])
            o = ptr_ref.solve_subproblem(mdl, op, scale, ref, pp, algo="scvx", eta=eta)
            assert g["status"][b] in (0, 1)
            assert abs(g["pcost"][b] - o["L_aug"]) <= 2e-7 * max(1.0, abs(o["L_aug"]))
            assert np.abs((g["u"][b] - o["u"]) / scale.Su).max() < 1e-4
            assert abs(g["p"][b, 0] - o["p"][0]) < 1e-4 * scale.Sp[0]
    pbm.close()


def test_scvx_follows_the_oracle_loop_and_ends_on_the_golden_run(pkg):
    """SCvx at the reference's own test parameters (freeflyer/tests.jl:25-80: N = 50, Nsub = 15, 15 iterations): same
    trust-region radii and accept / reject decisions as the oracle's literal loop, the same cost at every iteration, and the
    golden run's converged cost (0.23648) to 1e-5; a perturbed boundary condition rides along in the batch."""
    g = np.load(os.path.join(GOLD, "freeflyer_scvx_N50.npz"))
    N, Nsub, iters = int(g["N"]), int(g["Nsub"]), int(g["iters"])
    traj = pkg.TrajectoryProblem("freeflyer")
    pbm = pkg.SCvx.create(_scvx_pars(pkg, N, Nsub, iters), traj, batch_capacity=2)
    pp2 = g["pp"].copy(); pp2[13:16] += [0.05, -0.05, 0.02]
    sol, hist = pkg.SCvx.solve(pbm, np.stack([g["pp"], pp2]))
    pbm.close()
    assert sol.status == ["SCP_SOLVED", "SCP_SOLVED"] and sol.iterations[0] == iters and bool(sol.feas[0]) and bool(sol.feas[1])
    assert np.allclose(hist["eta"][:iters, 0], g["eta"], rtol=1e-12)
    # decisions are compared while the oracle loop still MOVES: from its 13th iteration on the solution cost repeats to 12 digits
    # (0.236475248363 twice), rho = dJ / dL is 0 / 0 and its sign is round-off (the oracle accepts, the device accepted with the
    # static regularisation 1e-8 and rejects with 1e-10); the radius sequence above is the same either way
    J = g["J_sol"][:iters - 1]
    moving = np.concatenate([[True], np.abs(np.diff(J)) > 1e-10 * np.maximum(1.0, np.abs(J[1:]))])
    assert moving.sum() >= 11
    assert np.array_equal((hist["accepted"][:iters - 1, 0] > 0)[moving], g["accept"][:iters - 1][moving])
    assert np.abs(hist["L"][:iters, 0] - g["L"]).max() <= 2e-5 * max(1.0, np.abs(g["L"]).max())
    ok = np.isfinite(g["J_sol"])
    assert np.abs(hist["J_sol"][:iters, 0][ok] - g["J_sol"][ok]).max() <= 1e-4 * max(1.0, np.abs(g["J_sol"][ok]).max())
    assert abs(hist["L"][iters - 1, 0] - 0.23648) < 1e-5 and abs(hist["L"][iters - 1, 0] - g["L"][-1]) < 1e-5 * max(1.0, g["L"][-1])
    from oracle import ptr_ref
    scale = ptr_ref.Scaling(*MODELS["freeflyer"](N).bbox())
    assert np.abs((sol.xd[0] - g["xd"]) / scale.Sx).max() < 2e-4 and np.abs((sol.ud[0] - g["ud"]) / scale.Su).max() < 2e-4
    assert abs(sol.p[0, 0] - g["p"][0]) < 2e-4 * scale.Sp[0]
    assert abs(hist["L"][iters - 1, 1] - g["L"][-1]) < 0.05 * g["L"][-1]       # the neighbour converged to a neighbouring cost


def test_gusto_follows_the_oracle_loop_and_ends_on_the_golden_run(pkg):
    """GuSTO (quadratic penalty) at the reference's own test parameters (freeflyer/tests.jl:84-140): the convex state set
    enters through its cone indicators (2 SOC speed limits, the t_f bounds at every node, 6 LINF rooms: 10 per node + the 4
    rows of s), the same (eta, lambda) sequence and accept / reject decisions as the oracle loop, the golden run's converged
    cost (0.22754) to 1e-5."""
    g = np.load(os.path.join(GOLD, "freeflyer_gusto_N50.npz"))
    N, Nsub, iters = int(g["N"]), int(g["Nsub"]), int(g["iters"])
    traj = pkg.TrajectoryProblem("freeflyer")
    pbm = pkg.GuSTO.create(_gusto_pars(pkg, N, Nsub, iters), traj, batch_capacity=1)
    assert pbm.template.nst == 14
    sol, hist = pkg.GuSTO.solve(pbm, g["pp"][None])
    pbm.close()
    assert sol.status == ["SCP_SOLVED"] and sol.iterations[0] == iters and bool(sol.feas[0])
    assert np.allclose(hist["eta"][:iters, 0], g["eta"], rtol=1e-12) and np.allclose(hist["lam"][:iters, 0], g["lam"], rtol=1e-12)
    assert np.array_equal(hist["accepted"][:iters - 1, 0], g["accept"][:iters - 1])
    assert np.abs(hist["L"][:iters, 0] - g["L"]).max() <= 1e-3 * max(1.0, np.abs(g["L"]).max())
    ok = np.isfinite(g["J_aug"])
    assert np.abs(hist["J_aug"][:iters, 0][ok] - g["J_aug"][ok]).max() <= 1e-3 * max(1.0, np.abs(g["J_aug"][ok]).max())
    assert abs(hist["L"][iters - 1, 0] - 0.22754) < 1e-5 and abs(hist["L"][iters - 1, 0] - g["L"][-1]) < 1e-5
    from oracle import ptr_ref
    scale = ptr_ref.Scaling(*MODELS["freeflyer"](N).bbox())
    assert np.abs((sol.xd[0] - g["xd"]) / scale.Sx).max() < 1e-3 and np.abs((sol.ud[0] - g["ud"]) / scale.Su).max() < 1e-3


def test_gusto_solution_costs_include_the_cone_indicators(pkg):
    """J_st of the device loop (gusto_post_kernel) against oracle/gusto_ref.state_penalty_nonconvex at the device's own
    iterate: an iterate pushed out of a room and over the speed limit must be charged for the violated cone indicators,
    not only for s (round-2 ADVICE: silently wrong rho otherwise)."""
    from oracle import gusto_ref
    N, Nsub = 12, 8
    mdl = MODELS["freeflyer"](N)
    traj = pkg.TrajectoryProblem("freeflyer")
    pbm = pkg.GuSTO.create(_gusto_pars(pkg, N, Nsub, 2), traj, batch_capacity=1)
    sol, hist = pkg.GuSTO.solve(pbm, mdl.nominal_pp()[None])
    pbm.close()
    op = gusto_ref.GuSTOParameters(N, Nsub, 2, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.5, beta_sh=2.0, beta_gr=2.0,
                                   gamma_fail=5.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=16, eps_abs=0.0,
                                   eps_rel=0.0, feas_tol=1e-3)
    k = int(sol.iterations[0]) - 1
    want = gusto_ref.state_penalty_nonconvex(mdl, op, sol.xd[0], sol.p[0], hist["lam"][k, 0])
    assert abs(hist["J_st"][k, 0] - want) <= 1e-9 * max(1.0, want) + 1e-12
    st, oh = gusto_ref.gusto_solve(mdl, op)
    for j in range(len(oh)):
        assert hist["eta"][j, 0] == pytest.approx(oh[j]["eta"], rel=1e-12) and hist["lam"][j, 0] == pytest.approx(oh[j]["lam"], rel=1e-12)
        assert abs(hist["L"][j, 0] + hist["L_st"][j, 0] + hist["L_tr"][j, 0] - oh[j]["sub"]["L_aug"]) <= 2e-5 * max(1.0, abs(oh[j]["sub"]["L_aug"]))
        if "accept" in oh[j]:
            assert bool(hist["accepted"][j, 0]) == bool(oh[j]["accept"])


def test_model_constants_cross_the_abi(pkg):
    """The model is DATA (src/parser/problem.jl:64-121): an override of the vehicle (mass) and of the environment (an
    obstacle moved onto a node of the guess) changes what the DEVICE computes -- discretize!, and the nonlinear cost of the
    same reference trajectory."""
    N, Nsub = 10, 6
    hit = [6.5 + 11.5 / 9.0, -0.2, 5.0]          # second node of the axis-by-axis guess (first leg along x)
    variants = dict(base={}, heavy=dict(m=14.4),
                    blocked=dict(obstacles=[(1.0 / 0.3, hit), (1.0 / 0.3, [11.2, 1.84, 5.0]), (1.0 / 0.3, [11.3, 3.8, 4.8])]))
    out = {}
    for name, kw in variants.items():
        traj = pkg.TrajectoryProblem("freeflyer", **kw)
        pbm = pkg.SCvx.create(_scvx_pars(pkg, N, Nsub, 1), traj, batch_capacity=1)
        x, u, p = traj.guess(N, traj.mdl.nominal_pp())
        ref = pkg.SubproblemSolutionBatch(x[None], u[None] + 1e-3, p[None], pbm)
        pkg.discretize_(ref, pbm)
        sol, hist = pkg.SCvx.solve(pbm, traj.mdl.nominal_pp()[None], guess=(x[None], u[None], p[None]), project_guess=False)
        out[name] = (ref.dyn.B[0].copy(), float(hist["J_ref"][0, 0]))
        pbm.close()
    dvdT = lambda B: B[0, :, 0, 3]                # d v_x / d T_x of every interval = (time) / m
    assert np.abs(dvdT(out["heavy"][0]) - 0.5 * dvdT(out["base"][0])).max() < 1e-12 * np.abs(dvdT(out["base"][0])).max()
    assert np.array_equal(out["blocked"][0], out["base"][0])
    # J = L + lambda (trapz(|defect|_1 + |max(s, 0)|_1) + ...): the blocked guess pays lambda w_1 (1 - 0) more
    assert out["blocked"][1] == pytest.approx(out["base"][1] + 1e3 * (1.0 / 9.0) * 1.0, rel=1e-9)
