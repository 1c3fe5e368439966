"""Whole Monte-Carlo batches of the bench records on the MI355X against the ORACLE's literal loops, instance by instance
("next" 1).  The goldens hold what oracle/{ptr,scvx,gusto}_ref.py + oracle/ipm.py produce on the SAME instances
(seed = instance index; tests/golden/make_{ptr,scvx,gusto,freeflyer_gusto}_outcomes.py):

* PTR, rocket landing N = 100 (the headline workload): first 256 instances -- status, dynamic feasibility flag, augmented cost;
* SCvx and GuSTO, quadrotor N = 30 at the reference's test parameters: 128-instance slices -- status, accept / reject decisions,
  radii, costs iteration by iteration;
* GuSTO, free-flyer N = 50 at the reference's test parameters: the 128 instances of the bench record;
* GuSTO, free-flyer at the config's N = 200: the four subproblems of the oracle's literal loop through the DEVICE path.

With eps_abs = eps_rel = 0 the reference's stopping rules (`<=`, ptr.jl:908-932, scvx.jl:711-734, gusto.jl:1203-1230) fire only
on EXACT equality of two costs (or a deviation of exactly 0): a loop that has reached its fixed point to the last bit stops there.
Whether the last bits coincide is round-off, so loops are compared over the iterations BOTH executed, and end values at each
loop's OWN last iteration."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def _dump(name, c):
    """the comparison record of a test, kept with the run's artefacts (gpurun_out/ travels back from the GPU box)"""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "outcomes_%s.json" % name), "w") as f:
            json.dump(c, f, indent=1, default=str)


def test_headline_batch_ends_like_the_literal_loop(pkg):
    """first 256 instances of `python bench.py` (rocket landing PTR N = 100, Nsub = 15, 15 iterations) vs the literal loop:
    every instance SCP_SOLVED in both, the same dynamic-feasibility flag on every instance, J_aug of the last subproblem 2e-6
    relative where both converged (measured for the CPU sibling: 9.4e-7 max, 5.7e-9 median), final time 2e-3 s."""
    import bench
    g = np.load(os.path.join(GOLD, "ptr_outcomes_rocket_landing_N100.npz"))
    nb = int(g["status"].size)
    traj = pkg.TrajectoryProblem("rocket_landing")
    pars = pkg.PTR.Parameters(N=int(g["N"]), Nsub=int(g["Nsub"]), iter_max=int(g["iter_max"]), wvc=1e3, wtr=0.1, eps_abs=0.0,
                              eps_rel=0.0, feas_tol=1e-3)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=nb)
    sol, h = pkg.PTR.solve(pbm, bench.mc_pp(traj.mdl, nb, 0), device_guess=True)
    pbm.close()
    o = bench.oracle_outcomes_ptr("rocket_landing", int(g["N"]), int(g["Nsub"]), int(g["iter_max"]), 0, sol)
    _dump("ptr_headline", o)
    assert o["instances"] == nb == 256
    assert o["same_status"] == 1.0 and o["oracle_frac_solved"] == 1.0
    assert o["same_feasibility_flag"] == 1.0, np.flatnonzero(sol.feas[:nb] != g["feas"])
    assert o["converged_in_both"] >= 230
    assert o["J_aug_rel_diff_max"] <= 2e-6 and o["tf_abs_diff_max_s"] <= 2e-3, o


def _scvx_pars(pkg, iters):
    return pkg.SCvx.Parameters(N=30, Nsub=15, iter_max=iters, lam=30.0, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0,
                               eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)


def test_scvx_quadrotor_slice_follows_the_literal_loop(pkg):
    """128 instances of bench.py's `scvx_quadrotor` record: same status; on every iteration both loops executed the same
    accept / reject decision and radius; the linearised cost L of every subproblem 2e-5 relative (tolerance of the loop test,
    tests/test_scvx_gpu.py) wherever the two loops are still on the same reference (all earlier decisions equal)."""
    import bench
    g = np.load(os.path.join(GOLD, "scvx_outcomes_quadrotor_N30.npz"))
    iters, nb = int(g["iter_max"]), 128
    traj = pkg.TrajectoryProblem("quadrotor")
    pbm = pkg.SCvx.create(_scvx_pars(pkg, iters), traj, batch_capacity=nb)
    sol, hist = pkg.SCvx.solve(pbm, bench.mc_pp(traj.mdl, nb, 0))
    pbm.close()
    c = bench.compare_scvx_outcomes(sol, hist, g, nb)
    _dump("scvx_quadrotor_first", c)
    assert c["same_status"] == 1.0
    # a decision may differ only where rho sits on a threshold of the update rule within the solvers' tolerance
    assert c["instances_with_a_different_decision"] <= c["instances_with_rho_on_a_threshold"], c
    _dump("scvx_quadrotor", c)
    # iteration 1 is the same program for both solvers (same projected guess): the optimum is pinned to the loop test's 2e-5; later
    # iterations linearise about solutions that may differ along flat directions of the earlier subproblems
    # (measured: first iteration 7.7e-7; later iterations median 4e-7, 90 % below 8e-5, 99 % below 1.4e-3, one instance 1.5e-2)
    assert c["L_rel_diff_first_iteration_max"] <= 2e-5, c
    q50, q90, q99 = c["L_rel_diff_quantiles_50_90_99"]
    assert q50 <= 1e-5 and q90 <= 1e-3 and q99 <= 1e-2 and c["L_rel_diff_max_on_common_path"] <= 5e-2, c
    assert c["last_L_rel_diff_max_same_decisions"] <= 5e-3, c
    assert c["eta_rel_diff_max_on_common_path"] <= 1e-12, c


def test_gusto_quadrotor_slice_follows_the_literal_loop(pkg):
    """128 instances of bench.py's `gusto_quadrotor` record (goal +-10 %, 6 iterations).  Wherever the ORACLE loop ends
    SCP_SOLVED the device ends SCP_SOLVED; the oracle's SCP_FAILED exits are exits of ITS conic solver on the escalated
    subproblems (lambda >= 1e6: ITERATION_LIMIT / NUMERICAL_ERROR, tests/test_outcomes_cpu.py) -- with the oracle's objective
    normalised (what the product's solver does) the oracle solves them too, and then the statuses agree on every instance.
    (eta, lambda) sequences and accept / reject decisions equal on the iterations both executed."""
    import bench
    g = np.load(os.path.join(GOLD, "gusto_outcomes_quadrotor_N30.npz"))
    iters, nb = int(g["iter_max"]), 128
    traj = pkg.TrajectoryProblem("quadrotor")
    gp = pkg.GuSTO.Parameters(N=30, Nsub=15, iter_max=iters, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.9, beta_sh=2.0, beta_gr=2.0,
                              gamma_fail=5.0, eta_init=10.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=6, eps_abs=0.0, eps_rel=0.0,
                              feas_tol=1e-3)
    pbm = pkg.GuSTO.create(gp, traj, batch_capacity=nb)
    sol, hist = pkg.GuSTO.solve(pbm, bench.mc_pp(traj.mdl, nb, 0))
    pbm.close()
    c = bench.compare_gusto_outcomes(sol, hist, g, nb)
    _dump("gusto_quadrotor", c)
    assert c["same_status_as_normalised_oracle"] == 1.0, c
    assert c["device_solved_where_oracle_solved"] == 1.0, c
    assert c["oracle_failures_are_solver_exits_at_large_lambda"] is True, c
    assert c["instances_with_a_different_decision"] <= c["instances_with_rho_on_a_threshold"], c
    assert c["lam_rel_diff_max_on_common_path"] <= 1e-12 and c["eta_rel_diff_max_on_common_path"] <= 1e-12, c
    assert c["L_aug_rel_diff_max_first_iteration"] <= 2e-5, c


def test_freeflyer_gusto_batch_follows_the_literal_loop(pkg):
    """the 128 instances of bench.py's `freeflyer_gusto.full_run_reference_grid` record (reference test parameters, N = 50,
    15 iterations): all SCP_SOLVED and dynamically feasible in both; accept / reject decisions equal on the iterations both
    loops executed; the cost at each loop's own last iteration 1e-5 relative (tolerance of the golden-run test)."""
    import bench
    g = np.load(os.path.join(GOLD, "gusto_outcomes_freeflyer_N50.npz"))
    nb = int(g["status"].size)
    sol, hist, _ = bench.freeflyer_gusto_full_run(pkg, int(g["N"]), int(g["Nsub"]), nb, int(g["iter_max"]))
    c = bench.compare_freeflyer_gusto_outcomes(sol, hist, g, nb)
    _dump("freeflyer_gusto", c)
    assert c["same_status"] == 1.0 and c["same_feasibility_flag"] == 1.0, c
    assert c["instances_with_a_different_decision"] == 0, c
    assert c["last_L_rel_diff_max"] <= 1e-5, c


def test_freeflyer_gusto_subproblems_at_config_size_on_the_device(pkg):
    """BASELINE.json configs[4] at its stated size (free-flyer, GuSTO, N = 200, np = 1 201): the four subproblems of the oracle's
    literal loop (tests/golden/freeflyer_gusto_N200.npz: reference trajectory, eta, lambda of every iteration) through the
    DEVICE path -- discretize!, linearise, gather into the n = 10 402 template, conic_ipm_kernel, read-out -- as one batch:
    safe status, the oracle's optimal value L_aug to 1e-6 relative."""
    g = np.load(os.path.join(GOLD, "freeflyer_gusto_N200.npz"))
    N, Nsub, nb = int(g["N"]), int(g["Nsub"]), int(g["eta"].size)
    traj = pkg.TrajectoryProblem("freeflyer")
    gp = pkg.GuSTO.Parameters(N=N, Nsub=Nsub, iter_max=1, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.5, beta_sh=2.0, beta_gr=2.0,
                              gamma_fail=5.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=16, eps_abs=0.0, eps_rel=0.0,
                              feas_tol=1e-3)
    pbm = pkg.GuSTO.create(gp, traj, batch_capacity=nb)
    assert (pbm.template.n, pbm.template.p, pbm.template.m) == (10402, 2613, 21602)
    r = pbm.sub.solve(g["ref_xd"], g["ref_ud"], g["ref_p"], pp=np.tile(g["pp"], (nb, 1)), scal=np.stack([g["eta"], g["lam"]], axis=1))
    pbm.close()
    assert (r["status"] <= 1).all(), r["status"]
    rel = np.abs(r["pcost"] - g["L_aug"]) / np.maximum(1.0, np.abs(g["L_aug"]))
    assert rel.max() <= 1e-6, rel
