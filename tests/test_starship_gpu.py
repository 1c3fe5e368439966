"""Starship landing flip (test/examples/starship_flip) on the MI355X (-m gpu): the first model with state-dependent
Jacobians, np = 10 and 23 non-convex rows.  discretize! runs the reference-form kernel K1 (cooperative LU per RK4
stage, Nsub = 100 as in the reference's tests); the subproblems go through the generic conic path."""
import os

import numpy as np
import pytest

from oracle import ptr_ref
from oracle.models import MODELS

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_discretize_matches_oracle_at_the_reference_config(pkg, orc):
    """N = 31, Nsub = 100 (starship_flip/tests.jl:35-36).  Phi^-1 is ill-conditioned here (gimbal-delay eigenvalue
    -20 * tdil, SURVEY section 7 hard part d): two fp64 evaluation orders agree to ~1e-9, not to round-off."""
    N, Nsub, B = 31, 100, 4
    traj = pkg.TrajectoryProblem("starship")
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=1, feas_tol=5e-3)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    rng = np.random.default_rng(0)
    xs, us, ps = [], [], []
    for b in range(B):
        x, u, p = traj.guess(N, traj.mdl.nominal_pp())
        xs.append(x * (1 + 0.02 * rng.standard_normal(x.shape))); us.append(u * (1 + 0.05 * rng.standard_normal(u.shape)))
        u[:, 1] += 0.05 * rng.standard_normal(N)
        ps.append(p * (1 + 0.05 * rng.standard_normal(p.shape)))
    ref = pkg.SubproblemSolutionBatch(np.stack(xs), np.stack(us), np.stack(ps), pbm)
    pkg.discretize_(ref, pbm)
    o = orc.discretize("starship", orc.default_params("starship"), N, Nsub, ref.xd, ref.ud, ref.p, pbm.scale.iSx, pars.feas_tol)
    for nm, got, want in (("A", ref.dyn.A, o["A"]), ("Bm", ref.dyn.B[0], o["Bm"]), ("Bp", ref.dyn.B[1], o["Bp"]),
                          ("F", ref.dyn.F, o["F"][:, :, :2]), ("r", ref.dyn.r, o["r"]), ("E", ref.dyn.E, o["E"]),
                          ("defect", ref.defect, o["defect"])):
        err = np.max(np.abs(got - want)) / max(1.0, np.max(np.abs(want)))
        assert err < 1e-8, (nm, err)
    assert np.abs(o["F"][:, :, 2:]).max() == 0.0       # only the t1 / t2 columns of F are structurally non-zero
    pbm.close()


def fp32_vs_fp64(pkg, N, Nsub, B, seed=0):
    """discretize! of perturbed Starship guesses in fp64 and in fp32 arithmetic (scp_set_discretize_precision)."""
    traj = pkg.TrajectoryProblem("starship")
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=1, feas_tol=5e-3)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    rng = np.random.default_rng(seed)
    x, u, p = traj.guess(N, traj.mdl.nominal_pp())
    xs = np.stack([x * (1 + 0.02 * rng.standard_normal(x.shape)) for _ in range(B)])
    us = np.stack([u * (1 + 0.05 * rng.standard_normal(u.shape)) for _ in range(B)])
    ps = np.stack([p * (1 + 0.05 * rng.standard_normal(p.shape)) for _ in range(B)])
    out = {}
    for bits in (64, 32, 64):
        pbm.set_discretize_precision(bits)
        ref = pkg.SubproblemSolutionBatch(xs, us, ps, pbm)
        pkg.discretize_(ref, pbm)
        out.setdefault(bits, []).append(ref)
    iSx = pbm.scale.iSx
    pbm.close()
    return out, iSx


def test_fp32_discretize_tolerance_check(pkg):
    """BASELINE.json configs[2]: 'fp64 vs fp32 tolerance check'.  The fp32 variant of K1 integrates the same quantities in
    single precision; Phi^-1 [..] loses digits to the gimbal-delay mode (SURVEY section 7 hard part d), so fp32 is a
    tolerance REPORT (bench.py, generic_path.fp32_discretize_starship), not a product default.  Here: the variant is
    really different arithmetic, its error is far above fp64 round-off and far below O(1), and switching back restores
    the fp64 results bit for bit."""
    out, iSx = fp32_vs_fp64(pkg, 31, 100, 4)
    a, b, c = out[64][0], out[32][0], out[64][1]
    for nm in ("A", "r", "E"):
        assert np.array_equal(getattr(a.dyn, nm), getattr(c.dyn, nm))
    assert np.array_equal(a.defect, c.defect)
    worst = 0.0
    for got, want in ((b.dyn.A, a.dyn.A), (b.dyn.B[0], a.dyn.B[0]), (b.dyn.B[1], a.dyn.B[1]), (b.dyn.r, a.dyn.r), (b.dyn.E, a.dyn.E)):
        worst = max(worst, np.max(np.abs(got - want)) / max(1.0, np.max(np.abs(want))))
    assert 1e-9 < worst < 5e-2, worst
    ddef = np.abs((b.defect - a.defect) * iSx[None, None, :]).max()
    assert ddef < 5e-2, ddef


def test_fp32_discretize_is_refused_where_it_does_not_exist(pkg):
    traj = pkg.TrajectoryProblem("quadrotor")
    pbm = pkg.PTR.create(pkg.PTR.Parameters(N=10, Nsub=5, iter_max=1), traj, batch_capacity=1)
    with pytest.raises(pkg._lib.ScpError):
        pbm.set_discretize_precision(32)
    with pytest.raises(pkg._lib.ScpError):
        pbm.set_discretize_precision(16)
    pbm.set_discretize_precision(64)
    pbm.close()


@pytest.mark.parametrize("N", [11, 31])
def test_ptr_subproblem_matches_oracle(pkg, N):
    mdl = MODELS["starship"](N)
    Nsub = 40
    traj = pkg.TrajectoryProblem("starship")
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=3, wvc=1e3, wtr=0.1, feas_tol=5e-3)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=2)
    scale = ptr_ref.Scaling(*mdl.bbox())
    opars = ptr_ref.PTRParameters(N, Nsub, 3, 1e3, 0.1, 0, 0, 5e-3)
    x, u, p = mdl.guess(N, mdl.nominal_pp())
    g = pkg.PTR.solve_subproblem_(pbm, np.stack([x, x]), np.stack([u, u]), np.stack([p, p]),
                                  pp=np.stack([mdl.nominal_pp()] * 2))
    ref = ptr_ref.discretize(mdl, opars, scale, x, u, p)
    o = ptr_ref.solve_subproblem(mdl, opars, scale, ref, mdl.nominal_pp())
    assert g["status"][0] in (0, 1) and o["status"] in ("OPTIMAL", "ALMOST_OPTIMAL")
    assert abs(g["pcost"][0] - o["J_aug"]) <= 2e-6 * max(1.0, abs(o["J_aug"]))
    assert np.abs((g["x"][0] - o["x"]) / scale.Sx).max() < 2e-4
    assert np.abs((g["u"][0] - o["u"])[:, :2] / scale.Su[:2]).max() < 2e-4
    assert np.array_equal(g["x"][0], g["x"][1])
    pbm.close()


def _golden():
    return np.load(os.path.join(GOLD, "starship_N31.npz"))


def test_reference_guess_on_device_matches_golden(pkg):
    """The reference's initial guess (bang-bang flip + convex terminal descent, definition.jl:97-445) through the library's kernels
    (`reference_guess` = scp_guess_batch_host, csrc/starship_guess.hpp): the 31 descent programs (t2 = 10 .. 40 s) solved as ONE
    batch on the device; the FIRST FEASIBLE duration is the fixture's (the oracle's restatement with the oracle's own
    interior-point solver; row-equilibrated programs decide every candidate, oracle/starship_guess.py), trajectory and hs too."""
    g = _golden()
    mdl = pkg.REGISTRY["starship"]()
    x, u, p = mdl.reference_guess(31)
    assert p[1] == g["guess_p"][1] == 21.0                      # definition.jl:395-412: first t2 with OPTIMAL | ALMOST_OPTIMAL
    assert abs(mdl.hs - float(g["hs"])) < 1e-9
    np.testing.assert_allclose(x[:15], g["guess_x"][:15], atol=1e-9)          # flip phase before the switch node: pure simulation
    # descent phase: a feasibility program (no cost) -- both interior-point methods end near the analytic centre of the feasible
    # set (measured on the host build of the product's solver against the oracle's: positions 2e-4 m of 600 m, thrust 2e-7 relative)
    assert np.abs(x[15:, 0:4] - g["guess_x"][15:, 0:4]).max() < 2e-2 and np.abs(u[15:, 0] - g["guess_u"][15:, 0]).max() < 1e-4 * 2210e3
    assert np.abs(x[-1, 0:2]).max() < 1e-6 and abs(x[-1, 3] + 0.1) < 1e-6
    assert (u[15:, 0] <= 2210e3 * (1 + 1e-9)).all() and (x[15:, 1] >= -1e-6).all()


def test_reference_guess_entirely_on_the_device_per_instance(pkg):
    """scp_guess_batch_host for the Starship model (csrc/starship_guess.hpp): flip simulation kernel + the descent
    programs of every (instance, duration) as one conic batch + reconstruction, nothing on the host.  Five Monte-Carlo instances
    (nominal + initial conditions +-2 %) against the ORACLE's guesses of the same instances (tests/golden/starship_guess_mc.npz:
    oracle/starship_guess.py with oracle/ipm.py): flip phase, switch state and t1 to 1e-9, the SAME first feasible descent duration
    on every instance (21, 20, 21, 20, 21 s), the descent trajectory near the oracle's; and at the config size N = 100 the nominal
    instance's duration (20 s)."""
    g = _golden()
    gm = np.load(os.path.join(GOLD, "starship_guess_mc.npz"))
    N = 31
    traj = pkg.TrajectoryProblem("starship")
    pars = pkg.PTR.Parameters(N=N, Nsub=20, iter_max=1)
    B = 5
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    pp = gm["pp"]
    x, u, p = pkg.device_guess(pbm, pp)
    assert pkg.device_guess_failures(pbm) == 0
    pbm.close()
    n1 = 16                                           # nodes with tau <= tau_s (the switch node included)
    np.testing.assert_allclose(x[0, :n1 - 1], g["guess_x"][:n1 - 1], atol=1e-9)
    np.testing.assert_allclose(u[0, :n1 - 1], g["guess_u"][:n1 - 1], atol=1e-9)
    assert np.array_equal(p[:, 1], gm["p"][:, 1]) and sorted(set(p[:, 1])) == [20.0, 21.0]             # first feasible durations
    for b in range(B):
        np.testing.assert_allclose(x[b, :n1 - 1], gm["x"][b, :n1 - 1], atol=1e-9)
        assert abs(p[b, 0] - gm["p"][b, 0]) < 1e-9 and np.abs(p[b, 2:] - gm["p"][b, 2:]).max() < 1e-9   # t1, xs
        assert np.abs(x[b, n1 - 1:, 0:4] - gm["x"][b, n1 - 1:, 0:4]).max() < 2e-2                       # (measured on the host build: 1.4e-3 m)
        assert np.abs(u[b, n1 - 1:, 0] - gm["u"][b, n1 - 1:, 0]).max() < 1e-4 * 2210e3
        assert np.abs(x[b, -1, 0:2]).max() < 1e-6 and abs(x[b, -1, 3] + 0.1) < 1e-6                 # lands at the pad with v_f
        assert np.allclose(x[b, n1 - 1, 0:4], p[b, 2:6], atol=1e-7)                                  # descent starts at the switch state
        assert (u[b, n1 - 1:, 0] <= 2210e3 * (1 + 1e-9)).all() and (u[b, n1 - 1:, 0] >= 880e3 * (1 - 1e-6)).all()
        assert (x[b, n1 - 1:, 1] >= -1e-6).all()
        assert (np.abs(x[b, n1 - 1:, 4]) <= np.deg2rad(15.0) + 1e-6).all()                           # tilt bound of phase 2
    assert np.abs(p[1:, 0] - p[0, 0]).min() > 1e-6            # the perturbed instances really have their own guesses
    # BASELINE.json configs[2] at its stated size
    x100, u100, p100 = pkg.REGISTRY["starship"]().reference_guess(100)
    assert p100[1] == gm["p100"][1] == 20.0 and abs(p100[0] - gm["p100"][0]) < 1e-9
    assert np.abs(x100[:, 0:4] - gm["x100"][:, 0:4]).max() < 2e-2


def test_ptr_loop_converges_like_the_oracle_loop(pkg):
    """PTR on the reference's own Starship test (starship_flip/tests.jl:35-49: N = 31, Nsub = 100, wvc = 1e3, wtr = 0.1,
    eps_abs = 1e-5, eps_rel = 1e-4, feas_tol = 5e-3) from the reference's guess: SCP_SOLVED after the same number of
    iterations, same costs and trajectory as the oracle's literal loop."""
    g = _golden()
    traj = pkg.TrajectoryProblem("starship", hs=float(g["hs"]))
    pars = pkg.PTR.Parameters(N=31, Nsub=100, iter_max=15, wvc=1e3, wtr=0.1, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=2)
    warm = tuple(np.stack([g[k]] * 2) for k in ("guess_x", "guess_u", "guess_p"))
    sol, hist = pkg.PTR.solve(pbm, np.stack([traj.mdl.nominal_pp()] * 2), warm=warm)
    pbm.close()
    assert sol.status[0] == "SCP_SOLVED" and str(g["ptr_status"]) == "SCP_SOLVED"
    assert sol.iterations[0] == int(g["ptr_iters"]) and sol.feas[0]
    for k in range(int(g["ptr_iters"])):
        assert abs(hist.J_aug[k, 0] - g["ptr_J_aug"][k]) <= 1e-5 * max(1.0, abs(g["ptr_J_aug"][k])), k
        assert bool(hist.feas[k, 0]) == bool(g["ptr_feas"][k])
    mdl = MODELS["starship"](31, float(g["hs"]))
    scale = ptr_ref.Scaling(*mdl.bbox())
    assert np.abs((sol.xd[0] - g["ptr_xd"]) / scale.Sx).max() < 1e-4
    assert np.abs((sol.p[0] - g["ptr_p"]) / scale.Sp).max() < 1e-4
    assert np.array_equal(sol.xd[0], sol.xd[1])


def test_scvx_loop_follows_the_oracle_loop(pkg):
    """SCvx on the reference's Starship test (starship_flip/tests.jl:77-98: lambda = 5e2, rho = (0, 0.1, 0.7), beta = 2,
    eta in [1e-8, 10], eta_init = 1) from the reference's guess: the trust-region radius sequence and the accept / reject
    decisions of the first iterations are the oracle's, the run ends dynamically feasible."""
    g = _golden()
    traj = pkg.TrajectoryProblem("starship", hs=float(g["hs"]))
    iters = 9
    pars = pkg.SCvx.Parameters(N=31, Nsub=100, iter_max=iters, lam=5e2, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0,
                               eta_init=1.0, eta_lb=1e-8, eta_ub=10.0, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3)
    pbm = pkg.SCvx.create(pars, traj, batch_capacity=1)
    sol, hist = pkg.SCvx.solve(pbm, traj.mdl.nominal_pp()[None], guess=tuple(g[k][None] for k in ("guess_x", "guess_u", "guess_p")))
    pbm.close()
    assert sol.status[0] == "SCP_SOLVED"
    for k in range(6):
        assert hist["eta"][k, 0] == pytest.approx(g["scvx_eta"][k], rel=1e-12)
        assert bool(hist["accepted"][k, 0]) == bool(g["scvx_accept"][k])
        assert abs(hist["L"][k, 0] - g["scvx_L"][k]) <= 1e-4 * max(1.0, abs(g["scvx_L"][k]))
    assert sol.feas[0] and bool(g["scvx_feas"][iters - 1])      # dynamically feasible from the 8th iteration on, like the oracle


@pytest.mark.parametrize("tag", ["", "_t21"])
def test_scvx_thirty_iterations_at_config_size_follow_the_oracle(pkg, tag):
    """BASELINE.json configs[2] AT ITS STATED SIZE for the oracle's whole record (VERDICT r04 "next" 1a): the device SCvx loop at
    N = 100, Nsub = 100, reference test parameters and stopping rule (starship_flip/tests.jl:77-98), `maxit = 1000` as the
    reference's tests hand ECOS, from the GOLDEN's guess, for all 30 iterations of tests/golden/starship_N100_scvx_long<tag>.npz:
    trust-region radii and accept / reject decisions identical, linearised cost L 1e-6 relative, nonlinear cost J_sol 2e-3
    (flat optimal faces of the LP) at every iteration.  Two oracle records: "" from the reference's guess as the oracle builds it
    (first feasible descent duration 20 s: the loop STALLS at an infeasible point, L_pen = 0.126, trust region 6e-5 -- in the oracle
    as on the device), "_t21" from the guess with the next duration (21 s: converges, L = 0.8042, dynamically feasible)."""
    import json
    g3 = np.load(os.path.join(GOLD, "starship_N100_scvx3%s.npz" % tag))
    g = np.load(os.path.join(GOLD, "starship_N100_scvx_long%s.npz" % tag))
    N, Nsub, iters = int(g["N"]), int(g["Nsub"]), int(g["iters"])
    assert iters == 30
    traj = pkg.TrajectoryProblem("starship", hs=float(g["hs"]))
    pars = pkg.SCvx.Parameters(N=N, Nsub=Nsub, iter_max=iters, lam=5e2, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0,
                               eta_init=1.0, eta_lb=1e-8, eta_ub=10.0, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3,
                               solver_opts=dict(max_iter=1000))
    pbm = pkg.SCvx.create(pars, traj, batch_capacity=2)
    guess = tuple(np.stack([g3[k]] * 2) for k in ("guess_x", "guess_u", "guess_p"))
    sol, hist = pkg.SCvx.solve(pbm, np.stack([traj.mdl.nominal_pp()] * 2), guess=guess)
    pbm.close()
    nit = int(sol.iterations[0])
    rows = [dict(k=k + 1, eta=[float(hist["eta"][k, 0]), float(g["eta"][k])], accept=[bool(hist["accepted"][k, 0]), bool(g["accept"][k])],
                 L=[float(hist["L"][k, 0]), float(g["L"][k])], J_sol=[float(hist["J_sol"][k, 0]), float(g["J_sol"][k])],
                 rho=[float(hist["rho"][k, 0]), float(g["rho"][k])], solver_status=int(hist["solver_status"][k, 0]),
                 solver_iters=int(hist["solver_iters"][k, 0])) for k in range(min(nit, iters))]
    d = os.path.join(os.path.dirname(GOLD), os.pardir, "gpurun_out")
    if os.path.isdir(d):
        json.dump(dict(status=sol.status[0], iterations=nit, p=[sol.p[0].tolist(), g["p"].tolist()], rows=rows),
                  open(os.path.join(d, "starship_scvx_N100_30_iterations%s.json" % tag), "w"), indent=1)
    assert sol.status[0] == "SCP_SOLVED" and str(g["status"]) == "SCP_SOLVED" and nit == iters
    # Measured on the device for the "_t21" record (gpurun_out/r05a): radii and decisions identical on all 30 iterations, L within
    # 9.4e-9, J_sol within 1.3e-4 (the CPU twin of the loop, tools/starship_twin.py -- the oracle loop with the product's template
    # and the host build of the product's solver: 5.8e-6 / 2.3e-3).  rho passes within 0.003 ... 0.03 of the rejection threshold
    # rho_0 = 0 at iterations 11, 14, 17, 20, 23 of that record (0.0265, 0.0027, -0.0322, -0.0284, 0.0239) and within 0.001 ... 0.03
    # at iterations 26-30 of the stalling record; a decision may differ from the oracle's ONLY at such an iteration and only with
    # rho equal to 0.02 -- the two loops then linearise about different references and are compared up to that iteration.
    # A second kind of legitimate fork (round 5, after the summation order of the device factorisation changed): an iteration whose
    # subproblem the DEVICE solver left at reduced accuracy (ALMOST_OPTIMAL, ECOS's inaccurate exit, acceptable per scp.jl:975).  The
    # subproblems are degenerate LPs; iteration 26 of the "_t21" record ends ALMOST_OPTIMAL on the host build of the product's solver
    # in EVERY summation order (168 ... 280 dynamic regularisations, dual residual 1e-7 ... 1e-6; the oracle's pivoting LU: OPTIMAL):
    # the optimal VALUE still agrees to 1e-7 (asserted here and, subproblem by subproblem about the oracle's references, in
    # tests/test_teacher_forced_gpu.py), but the point on the optimal face differs, J_sol with it (lambda = 500 times the defects),
    # and the decision may flip.  Whether that exit is OPTIMAL or ALMOST_OPTIMAL on the device is round-off (it was OPTIMAL in the
    # run of gpurun_out/r05b, where the device followed all 30 iterations of both records).
    fork = None
    for k in range(iters):
        almost = int(hist["solver_status"][k, 0]) == 1
        assert hist["eta"][k, 0] == pytest.approx(g["eta"][k], rel=1e-12), rows[k]
        assert abs(hist["L"][k, 0] - g["L"][k]) <= 1e-6 * max(1.0, abs(g["L"][k])), rows[k]
        if not almost:
            assert abs(hist["J_sol"][k, 0] - g["J_sol"][k]) <= 2e-3 * max(1.0, abs(g["J_sol"][k])), rows[k]
        if k < iters - 1 and bool(hist["accepted"][k, 0]) != bool(g["accept"][k]):
            ro, rd = float(g["rho"][k]), float(hist["rho"][k, 0])
            assert almost or (min(abs(ro - t) for t in (0.0, 0.1, 0.7)) <= 0.03 and abs(ro - rd) <= 0.02), rows[k]
            fork = k
            break
    assert fork is None or fork >= 10, (fork, rows[fork] if fork is not None else None)
    assert np.array_equal(sol.xd[0], sol.xd[1])


def test_scvx_monte_carlo_instances_at_config_size_follow_the_oracle(pkg):
    """BASELINE.json configs[2] as a MONTE-CARLO workload at its stated size (VERDICT r05 missing 4 / next 1b): eight instances of the bench
    batch (initial conditions +-2 %, seed = instance: 1, 2, 3, 4, 5, 7, 8 and 64 -- the instance that ended SCP_FAILED in round 5), each from
    its own ORACLE guess (first feasible descent durations of 20 s and of 21 s both occur), 30 iterations of the oracle's literal SCvx loop
    (tests/golden/make_starship_n100_mc.py) against the device loop on the same eight problems as ONE batch from the same guesses:
    no instance fails; trust-region radii identical and the linearised cost within 5e-6 (measured: 1.5e-6 on one iteration of one instance,
    <= 1e-6 elsewhere; the LPs have flat optimal faces along which L trades against the penalty at equal L_aug) at every iteration up to the first legitimate fork
    (the rules of test_scvx_thirty_iterations_at_config_size_follow_the_oracle: rho within 0.03 of an update threshold, or a subproblem the
    device left at reduced accuracy); the final dynamic-feasibility flags of the instances that never forked are the oracle's."""
    import json
    g = np.load(os.path.join(GOLD, "starship_N100_scvx_mc.npz"))
    N, Nsub, iters, inst = int(g["N"]), int(g["Nsub"]), int(g["iters_max"]), [int(v) for v in g["instances"]]
    B = len(inst)
    hs0 = float(np.load(os.path.join(GOLD, "starship_guess_mc.npz"))["hs100"])
    traj = pkg.TrajectoryProblem("starship", hs=hs0)
    pars = pkg.SCvx.Parameters(N=N, Nsub=Nsub, iter_max=iters, lam=5e2, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0,
                               eta_init=1.0, eta_lb=1e-8, eta_ub=10.0, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3,
                               solver_opts=dict(max_iter=1000))
    pbm = pkg.SCvx.create(pars, traj, batch_capacity=B)
    sol, hist = pkg.SCvx.solve(pbm, g["pp"], guess=(g["guess_x"], g["guess_u"], g["guess_p"]))
    pbm.close()
    recs = []
    for b in range(B):
        Ko = int(g["iters"][b]); Kd = int(sol.iterations[b])
        fork, why = None, None
        for k in range(min(Ko, Kd)):
            almost = int(hist["solver_status"][k, b]) == 1
            same_eta = hist["eta"][k, b] == pytest.approx(g["eta"][b, k], rel=1e-12)
            dL = abs(hist["L"][k, b] - g["L"][b, k]) / max(1.0, abs(g["L"][b, k]))
            if not same_eta or dL > 5e-6:
                fork, why = k, "eta / L differ without a decision fork before (eta %r vs %r, dL %.2e)" % (float(hist["eta"][k, b]), float(g["eta"][b, k]), dL)
                break
            if k < min(Ko, Kd) - 1 and bool(hist["accepted"][k, b]) != bool(g["accept"][b, k] > 0):
                ro, rd = float(g["rho"][b, k]), float(hist["rho"][k, b])
                legit = almost or (min(abs(ro - t) for t in (0.0, 0.1, 0.7)) <= 0.03 and abs(ro - rd) <= 0.02)
                fork, why = k, ("decision fork at a threshold (rho oracle %.4f device %.4f%s)" % (ro, rd, ", reduced-accuracy exit" if almost else "")) if legit else \
                    "ILLEGITIMATE decision fork (rho oracle %.4f device %.4f)" % (ro, rd)
                break
        recs.append(dict(instance=inst[b], guess_t2=float(g["guess_p"][b, 1]), device_status=sol.status[b], oracle_status=str(g["status"][b]),
                         device_iterations=Kd, oracle_iterations=Ko, fork=fork, why=why, device_feas=bool(sol.feas[b]), oracle_feas=bool(g["final_feas"][b]),
                         L_last=[float(hist["L"][min(Kd, iters) - 1, b]), float(g["L"][b, Ko - 1])]))
    summary = dict(instances=B, device_frac_dyn_feasible=float(np.mean(sol.feas)), oracle_frac_dyn_feasible=float(np.mean(g["final_feas"])),
                   followed_all_iterations=int(sum(r["fork"] is None for r in recs)), records=recs)
    d = os.path.join(os.path.dirname(GOLD), os.pardir, "gpurun_out")
    if os.path.isdir(d):
        json.dump(summary, open(os.path.join(d, "starship_scvx_N100_mc.json"), "w"), indent=1)
    print(json.dumps(summary))
    assert all(r["device_status"] == "SCP_SOLVED" for r in recs), recs            # (instance 64 included)
    for r in recs:
        assert r["why"] is None or r["why"].startswith("decision fork at a threshold"), r
        if r["fork"] is None:
            assert r["device_feas"] == r["oracle_feas"] and r["device_iterations"] == r["oracle_iterations"], r
    assert sum(r["fork"] is None or r["fork"] >= 10 for r in recs) >= B - 2, recs
