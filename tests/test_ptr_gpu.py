"""End-to-end GPU parity of the batched PTR loop (discretize! + solve_subproblem! + stopping logic through the
C ABI) against the oracle's literal restatement of src/solvers/ptr.jl on identical inputs.

Stated tolerances (fp64): converged trajectory 1e-4 in scaled variables, converged cost 1e-6 relative;
intermediate iterates 5e-3 relative in the augmented cost (see the comment in the test)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle(model, N, Nsub, iters, pp=None):
    from oracle import ptr_ref
    from oracle.models import MODELS
    mdl = MODELS[model]()
    pars = ptr_ref.PTRParameters(N, Nsub, iters, 1e3, 0.1, 0, 0, 1e-3)
    st, hist = ptr_ref.ptr_solve(model, pars, pp=pp)
    return mdl, ptr_ref.Scaling(*mdl.bbox()), st, hist


@pytest.mark.parametrize("model,N,Nsub,iters", [("quadrotor", 30, 15, 15), ("double_integrator", 30, 10, 6),
                                                ("rocket_landing", 16, 10, 10)])
def test_ptr_loop_matches_oracle(pkg, orc, model, N, Nsub, iters):
    mdl, scale, st, hist = _oracle(model, N, Nsub, iters)
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=2)
    pp = np.repeat(traj.mdl.nominal_pp()[None], 2, 0)
    sol, h = pkg.PTR.solve(pbm, pp)
    assert sol.status == ["SCP_SOLVED", "SCP_SOLVED"] and st == "SCP_SOLVED"
    # eps_abs = eps_rel = 0: the reference's rule (`<=`, ptr.jl:924-927) fires on EXACT equality only.  A warm-started solve of a
    # converged problem can return the snapshot point bit for bit (0 IPM iterations; INTEGRATION.md 3.1 -- more often since the fine
    # snapshot level is 1e-9, round 6): the loop may then stop before iter_max, and only for that reason
    nit = int(sol.iterations[0])
    assert (sol.iterations == nit).all() and 2 <= nit <= iters
    if nit < iters:
        assert h.improv_rel[nit - 1, 0] == 0.0 or h.deviation[nit - 1, 0] == 0.0, (nit, h.improv_rel[nit - 1, 0], h.deviation[nit - 1, 0])
        assert nit >= iters // 2, nit
    # identical problems in the batch give identical results
    assert np.array_equal(sol.xd[0], sol.xd[1])
    for it in range(nit):
        o = hist[it]
        assert bool(h.feas[it, 0]) == o["sol"].feas, it
        # iterates are compared loosely: the first subproblems (virtual control active) have flat optimal
        # faces, the two interior-point solvers stop at slightly different points of them and the next
        # linearisation inherits the difference; the CONVERGED result below is compared tightly
        if o["sub"]["J_vc"] < 1e-6:
            assert abs(h.J_aug[it, 0] - o["sub"]["J_aug"]) <= 5e-3 * max(1.0, abs(o["sub"]["J_aug"])), (it, h.J_aug[it, 0])
    fin = hist[-1]["sol"]
    k = 1.0   # one stated tolerance for every model
    assert np.abs((sol.xd[0] - fin.xd) / scale.Sx).max() <= k * 1e-4
    assert np.abs((sol.ud[0] - fin.ud) / scale.Su).max() <= k * 1e-4
    if mdl.np:
        assert np.abs((sol.p[0] - fin.p) / scale.Sp).max() <= k * 1e-4
    assert abs(sol.J[0] - hist[-1]["sub"]["J"]) <= 1e-6 * max(1.0, abs(hist[-1]["sub"]["J"])) * k
    assert sol.feas.all() and fin.feas
    pbm.close()


def test_monte_carlo_batch_and_stopping(pkg, orc):
    """Perturbed initial conditions (x0*(1+0.1 xi), seed = problem index) with the reference's
    stopping rule enabled: every problem must stop on its own iteration, identically to the oracle."""
    model, N, Nsub, iters = "quadrotor", 12, 8, 12
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=1e-4, eps_rel=1e-5)
    B = 3
    pp = []
    for b in range(B):
        rng = np.random.default_rng(b)
        q = traj.mdl.nominal_pp().copy()
        q[6:9] = q[6:9] * (1 + 0.1 * rng.uniform(-1, 1, 3))  # r0 = 0, so the terminal position carries the spread
        pp.append(q)
    pp = np.stack(pp)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    sol, h = pkg.PTR.solve(pbm, pp)
    from oracle import ptr_ref
    for b in range(B):
        opars = ptr_ref.PTRParameters(N, Nsub, iters, 1e3, 0.1, 1e-4, 1e-5, 1e-3)
        st, hist = ptr_ref.ptr_solve(model, opars, pp=pp[b])
        scale = ptr_ref.Scaling(*traj.mdl.scale_advice())
        assert sol.status[b] == st
        assert abs(int(sol.iterations[b]) - len(hist)) <= 1, (b, sol.iterations[b], len(hist))
        fin = hist[-1]["sol"]
        # both loops stop as soon as the deviation drops below eps_abs = 1e-4 (possibly one iteration
        # apart), so the results agree to a small multiple of that stopping tolerance, not better
        assert np.abs((sol.xd[b] - fin.xd) / scale.Sx).max() <= 3e-3
        assert np.abs((sol.ud[b] - fin.ud) / scale.Su).max() <= 3e-3
    # stopped problems are frozen: history marks them inactive afterwards
    for b in range(B):
        n = int(sol.iterations[b])
        assert h.active[:n, b].all() and not h.active[n:, b].any()
    pbm.close()


def test_full_size_batch_properties(pkg):
    """BASELINE-size workload (rocket landing, N=100, Monte-Carlo batch): size-independent properties --
    all problems solved, dynamically feasible at the end, virtual control gone, the cost split adds up,
    restart on the device reproduces the run bit for bit."""
    model, N, Nsub, iters, B = "rocket_landing", 100, 15, 15, 128
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    pp = []
    for b in range(B):
        rng = np.random.default_rng(b)
        pp.append(traj.mdl.nominal_pp() * (1 + 0.1 * rng.uniform(-1, 1, 6)))
    pp = np.stack(pp)
    sol, h = pkg.PTR.solve(pbm, pp)
    assert all(s == "SCP_SOLVED" for s in sol.status)
    # +-10 % initial conditions: a few instances need more than 15 iterations (or are infeasible)
    assert sol.feas.mean() >= 0.9
    assert (h.J_vc[-1][sol.feas] < 1e-4).all()
    np.testing.assert_allclose(h.J[-1] + h.J_tr[-1] + h.J_vc[-1], h.J_aug[-1], rtol=1e-12)
    assert (h.solver_status <= 1).all()
    # final mass is physical and the time of flight within its bounds
    assert (np.exp(sol.xd[:, -1, 6]) > 1505.0 - 1e-6).all() and (np.exp(sol.xd[:, -1, 6]) < 1905.0).all()
    assert ((sol.p[:, 0] > 40.0) & (sol.p[:, 0] < 120.0)).all()
    pkg.PTR.restart(pbm)
    pkg.PTR.run_resident(pbm)
    sol2, _ = pkg.PTR.collect(pbm, B)
    assert np.array_equal(sol.xd, sol2.xd) and np.array_equal(sol.ud, sol2.ud)
    pbm.close()


def test_large_batch_kernel_variant_matches_small_batch(pkg):
    """Batches larger than 4 problems per CU run the two-waves-per-SIMD build of the IPM kernel; it must produce the
    same trajectories as the one-wave build used for small batches (same algorithm, different register budget)."""
    model, N, Nsub, iters, B, Bs = "quadrotor", 12, 8, 4, 1100, 48
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0)
    pp = []
    for b in range(B):
        rng = np.random.default_rng(b % Bs)       # the large batch repeats the small one
        q = traj.mdl.nominal_pp().copy()
        q[6:9] = q[6:9] * (1 + 0.1 * rng.uniform(-1, 1, 3))
        pp.append(q)
    pp = np.stack(pp)
    big = pkg.PTR.create(pars, traj, batch_capacity=B)
    sol_b, h_b = pkg.PTR.solve(big, pp)
    big.close()
    small = pkg.PTR.create(pars, traj, batch_capacity=Bs)
    sol_s, h_s = pkg.PTR.solve(small, pp[:Bs])
    small.close()
    assert all(s == "SCP_SOLVED" for s in sol_b.status)
    for b in (0, 1, Bs - 1, Bs, 2 * Bs + 5, B - 1):
        r = b % Bs
        np.testing.assert_allclose(sol_b.xd[b], sol_s.xd[r], rtol=0, atol=1e-7)
        np.testing.assert_allclose(sol_b.ud[b], sol_s.ud[r], rtol=0, atol=1e-7)
        np.testing.assert_allclose(h_b.J_aug[:, b], h_s.J_aug[:, r], rtol=1e-8)


@pytest.mark.parametrize("model,N", [("double_integrator", 9), ("quadrotor", 12), ("rocket_landing", 20)])
def test_device_side_guess_equals_host_guess(pkg, model, N):
    """scp_ptr_init_guess_host (traj.guess on the device, only pp crosses PCIe) starts the PTR solve from the
    trajectories the host-side guess rule produces (to the last bit of log()), and -- where the subproblems are solved
    to full accuracy -- ends on the same solution."""
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=6, iter_max=3, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0)
    B = 7
    rng = np.random.default_rng(2)
    nom = traj.mdl.nominal_pp()
    pp = np.stack([nom * (1 + 0.1 * rng.uniform(-1, 1, nom.size)) for _ in range(B)])
    b = pkg.PTR.create(pars, traj, batch_capacity=B)
    pkg.PTR.upload(b, pp, device_guess=True)
    g, _ = pkg.PTR.collect(b, B)          # before the first iteration the reference trajectory IS the guess
    for i in range(B):
        x, u, p = traj.guess(N, pp[i])
        np.testing.assert_allclose(g.xd[i], x, rtol=1e-15, atol=1e-15)
        np.testing.assert_allclose(g.ud[i], u, rtol=1e-15, atol=1e-15)
        np.testing.assert_allclose(g.p[i], p, rtol=1e-15, atol=0)
    sol_d, h_d = pkg.PTR.solve(b, pp, device_guess=True)
    b.close()
    if model != "rocket_landing":         # rocket subproblems exit at ECOS' reduced tolerances: iterates are not unique to 1e-9
        a = pkg.PTR.create(pars, traj, batch_capacity=B)
        sol_h, h_h = pkg.PTR.solve(a, pp)
        a.close()
        np.testing.assert_allclose(sol_d.xd, sol_h.xd, rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(sol_d.ud, sol_h.ud, rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(h_d.J_aug, h_h.J_aug, rtol=1e-8)


def test_stream_groups_match_single_handle(pkg):
    """A batch split over several handles / HIP streams with iterations enqueued ahead (scp_ptr_iterate_async +
    scp_ptr_poll) gives the same result as one handle iterated synchronously: problems are independent."""
    model, N, Nsub, iters, B = "rocket_landing", 30, 10, 8, 12
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0)
    pp = np.stack([traj.mdl.nominal_pp() * (1 + 0.1 * np.random.default_rng(b).uniform(-1, 1, 6)) for b in range(B)])
    one = pkg.PTR.create(pars, traj, batch_capacity=B)
    sol1, h1 = pkg.PTR.solve(one, pp, device_guess=True)
    grp = pkg.PTR.SCPProblemGroup(pars, traj, batch_capacity=B, streams=5)      # uneven split: 3,3,2,2,2
    assert [hi - lo for lo, hi in grp.ranges] == [3, 3, 2, 2, 2]
    pkg.PTR.group_upload(grp, pp, device_guess=True)
    n = pkg.PTR.group_run_resident(grp, lookahead=3)
    assert n == 9                                   # 3 chunks of 3: the 9th iteration is a no-op past iter_max
    sol2, h2 = pkg.PTR.group_collect(grp)
    assert sol2.status == sol1.status and (sol2.iterations == sol1.iterations).all()
    assert np.array_equal(sol1.xd, sol2.xd) and np.array_equal(sol1.ud, sol2.ud) and np.array_equal(sol1.p, sol2.p)
    assert np.array_equal(h1.solver_iters, h2.solver_iters)
    # restart + one-iteration look-ahead reproduces it again
    pkg.PTR.group_restart(grp)
    assert pkg.PTR.group_run_resident(grp, lookahead=1) == iters
    sol3, _ = pkg.PTR.group_collect(grp)
    assert np.array_equal(sol1.xd, sol3.xd)
    one.close(); grp.close()


def test_four_level_warm_start_changes_the_work_not_the_result(pkg):
    """Round 6: the structured solver restarts a subproblem from one of FOUR snapshots of the problem's previous solve, chosen by the scale
    of the reference move (include/scp_mi355x.h: ipm_warm_*).  On a Monte-Carlo batch at the headline size the run with the
    warm start and the run with cold solves (`warm = 0`) must agree -- same statuses and feasibility flags, converged costs to 1e-6,
    converged trajectories as two members of PTR's fixed-point set (5e-4 scaled) -- while the warm run needs well under two thirds of the interior-point iterations, its late
    launches a handful per problem, and no launch is held up by a warm attempt that ran to its limit and was repeated cold."""
    model, N, Nsub, iters, B = "rocket_landing", 100, 15, 15, 128
    traj = pkg.TrajectoryProblem(model)
    pp = np.stack([traj.mdl.nominal_pp() * (1 + 0.1 * np.random.default_rng(b).uniform(-1, 1, 6)) for b in range(B)])
    runs = {}
    for name, opts in (("warm", {}), ("cold", {"warm": 0})):
        pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0, solver_opts=opts)
        pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
        runs[name] = pkg.PTR.solve(pbm, pp)
        pbm.close()
    (sw, hw), (sc, hc) = runs["warm"], runs["cold"]
    assert sw.status == sc.status and np.array_equal(sw.feas, sc.feas)
    from oracle import ptr_ref
    scale = ptr_ref.Scaling(*traj.mdl.scale_advice())          # (diagonal scaling: vectors)
    Sx, Su = np.asarray(scale.Sx).reshape(-1), np.asarray(scale.Su).reshape(-1)
    if Sx.size != sw.xd.shape[-1]:
        Sx, Su = np.diag(np.asarray(scale.Sx)), np.diag(np.asarray(scale.Su))
    conv = sw.feas.astype(bool) & (hc.J_vc[-1] < 1e-6)          # converged in both: dynamically feasible, virtual control gone
    assert conv.sum() >= 0.85 * B
    assert np.abs(sw.J[conv] - sc.J[conv]).max() <= 1e-6 * np.abs(sc.J[conv]).max()
    # (trajectories: PTR's soft trust region w_tr |dx| is non-smooth -- every point whose reduced gradient is below w_tr is a fixed point, and
    # two correct loops stop at members of that set 3e-4 apart on the rocket with costs equal to 5e-8)
    dxm, dum = (np.abs(sw.xd[conv] - sc.xd[conv]) / Sx).max(), (np.abs(sw.ud[conv] - sc.ud[conv]) / Su).max()
    # (inputs: thrust can be redistributed between neighbouring nodes at equal cost to the solver's tolerance -- the flat direction
    # tests/test_config_size_gpu.py documents; measured 3e-2 of the input range on one node of one instance)
    assert dxm <= 5e-4 and dum <= 5e-2, (dxm, dum)
    # the work: executed interior-point iterations (a warm run may also stop a few problems early on exact equality)
    itw, itc = hw.solver_iters.sum(), hc.solver_iters.sum()
    assert itw <= 0.62 * itc, (itw, itc)
    late = hw.solver_iters[8:][hw.active[8:].astype(bool)]
    assert np.median(late) <= 5 and np.percentile(late, 99) <= 30, (np.median(late), np.percentile(late, 99))
    assert hw.solver_iters[8:].max() <= 60, hw.solver_iters[8:].max()        # (45 + cold repeat = 80 ... 105 before the level rules)


def test_cpu_twin_mirrors_the_devices_iteration_statistics(pkg):
    """oracle/cpu_ptr.cpp is where this round's warm-start rules were found before they went into the kernel: it must go on
    mirroring the device -- per PTR iteration the mean interior-point iteration count of a 64-instance Monte-Carlo batch within one iteration
    of the device's (cold launch, the launches from the coarse / mid levels, the late launches of a converged run), the same statuses of the
    final solves, and converged costs to 1e-6."""
    from oracle import cpu_ptr
    model, N, Nsub, iters, B = "rocket_landing", 100, 15, 15, 64
    traj = pkg.TrajectoryProblem(model)
    import bench
    pp = bench.mc_pp(traj.mdl, B, 0)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    sol, h = pkg.PTR.solve(pbm, pp)
    pbm.close()
    c = cpu_ptr.solve_batch(model, N, Nsub, iters, pp, want_hist=True)
    dev = np.where(h.active.astype(bool), h.solver_iters, np.nan)
    cpu = c["hist"][:, :, 4].T                                   # [iter, B]
    m_dev, m_cpu = np.nanmean(dev, axis=1), cpu.mean(axis=1)
    assert np.abs(m_dev - m_cpu).max() <= 1.0, (m_dev.round(1).tolist(), m_cpu.round(1).tolist())
    assert m_dev[0] > 30 and m_dev[-1] < 4 and m_cpu[-1] < 4     # cold start ... a converged run's re-solves
    last = np.array([int(sol.iterations[b]) - 1 for b in range(B)])
    st_dev = h.solver_status[last, np.arange(B)]
    assert (st_dev <= 1).all() and (c["hist"][:, -1, 5] <= 1).all()
    conv = sol.feas.astype(bool)
    Jd = h.J_aug[last, np.arange(B)]
    assert np.abs(Jd[conv] - c["hist"][conv, -1, 0]).max() <= 1e-6 * np.abs(Jd[conv]).max()
