"""GPU parity of K1 (`discretize!` in HIP, through the C ABI) against the CPU
oracle on identical seeded inputs.  Tolerance: 1e-10 relative (fp64; SURVEY.md
§8c) -- both sides integrate the same augmented ODE with the same RK4 grid.

Two device kernels are covered: the variational form K1v (default for models with constant Jacobians) and the
reference formulation K1 (int Phi^-1 [..] then Phi *, forced with SCP_DISC_REFERENCE_FORM=1)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-10
NAMES = ("A", "Bm", "Bp", "F", "r", "E", "defect")


def _run(pkg, orc, model, N, Nsub, B, seed, feas_tol=1e-3, noise=0.05):
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=1, feas_tol=feas_tol)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    rng = np.random.default_rng(seed)
    xs, us, ps = [], [], []
    pp0 = traj.mdl.nominal_pp()
    for b in range(B):
        pp = pp0 * (1 + 0.1 * rng.uniform(-1, 1, size=pp0.size))
        x, u, p = traj.guess(N, pp)
        xs.append(x + noise * rng.standard_normal(x.shape) * (1 + np.abs(x)))
        us.append(u + noise * rng.standard_normal(u.shape))
        ps.append(p * (1 + 0.1 * rng.uniform(-1, 1, size=p.shape)))
    ref = pkg.SubproblemSolutionBatch(np.stack(xs), np.stack(us), np.stack(ps).reshape(B, -1), pbm)
    pkg.discretize_(ref, pbm)
    o = orc.discretize(model, orc.default_params(model), N, Nsub, ref.xd, ref.ud, ref.p, pbm.scale.iSx, feas_tol)
    got = dict(A=ref.dyn.A, Bm=ref.dyn.B[0], Bp=ref.dyn.B[1], F=ref.dyn.F, r=ref.dyn.r, E=ref.dyn.E,
               defect=ref.defect)
    pbm.close()
    return ref, got, o


@pytest.mark.parametrize("model,N,Nsub,B", [
    ("double_integrator", 30, 10, 5),     # configs[0] sizes
    ("quadrotor", 50, 15, 7),             # configs[1]
    ("rocket_landing", 100, 15, 3),       # configs[3] sizes, small batch
    ("quadrotor", 2, 2, 1),               # minimum grid, single problem
    ("rocket_landing", 3, 2, 33),         # ragged: intervals not a multiple of the groups per block (coarse grid: K1)
    ("rocket_landing", 11, 11, 4),        # coarsest grid on which the variational kernel is dispatched for this model
])
@pytest.mark.parametrize("form", ["variational", "reference"])
def test_discretize_parity(pkg, orc, model, N, Nsub, B, form, monkeypatch):
    if form == "reference":
        monkeypatch.setenv("SCP_DISC_REFERENCE_FORM", "1")   # read at scp_problem_create
    else:
        monkeypatch.delenv("SCP_DISC_REFERENCE_FORM", raising=False)
    ref, got, o = _run(pkg, orc, model, N, Nsub, B, seed=10)
    for nm in NAMES:
        scale = max(1.0, float(np.max(np.abs(o[nm])))) if o[nm].size else 1.0
        err = float(np.max(np.abs(got[nm] - o[nm]))) / scale if o[nm].size else 0.0
        assert err < TOL, (nm, err)
    assert (ref.feas == o["feas"]).all()


def test_feasibility_flag_both_ways(pkg, orc):
    # a dynamically consistent trajectory (tiny noise) must be flagged feasible with a
    # loose tolerance and infeasible with a tight one, identically to the oracle
    ref, got, o = _run(pkg, orc, "quadrotor", 20, 10, 6, seed=3, feas_tol=10.0, noise=1e-3)
    assert ref.feas.all() and o["feas"].all()
    ref, got, o = _run(pkg, orc, "quadrotor", 20, 10, 6, seed=3, feas_tol=1e-9, noise=1e-3)
    assert (~ref.feas).all() and (~o["feas"]).all()


def test_full_size_batch_properties(pkg):
    """BASELINE-size batch (rocket, N=100, B=1024): size-independent properties --
    the linearisation identity holds at every (b,k) and identical problems give
    bit-identical outputs."""
    model, N, Nsub, B = "rocket_landing", 100, 15, 1024
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=1)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    rng = np.random.default_rng(0)
    x, u, p = traj.guess(N)
    xd = np.repeat(x[None], B, 0); ud = np.repeat(u[None], B, 0); pd = np.repeat(p[None], B, 0)
    xd[1:] += 0.05 * rng.standard_normal(xd[1:].shape) * (1 + np.abs(xd[1:]))
    xd[-1] = xd[0]; ud[-1] = ud[0]
    ref = pkg.SubproblemSolutionBatch(xd, ud, pd, pbm)
    pkg.discretize_(ref, pbm)
    d = ref.dyn
    lin = (np.einsum("bkji,bkj->bki", d.A, xd[:, :-1]) + np.einsum("bkji,bkj->bki", d.B[0], ud[:, :-1])
           + np.einsum("bkji,bkj->bki", d.B[1], ud[:, 1:]) + np.einsum("bkji,bj->bki", d.F, pd) + d.r)
    resid = xd[:, 1:] - ref.defect - lin
    assert np.max(np.abs(resid)) < 1e-8 * max(1.0, np.max(np.abs(xd)))
    for arr in (d.A, d.B[0], d.B[1], d.F, d.r, d.E, ref.defect):
        assert np.array_equal(arr[0], arr[-1])
    pbm.close()


@pytest.mark.parametrize("model,N,res", [("double_integrator", 10, 57), ("quadrotor", 20, 200), ("rocket_landing", 30, 1000),
                                         ("quadrotor", 2, 2)])
def test_propagate_parity(pkg, orc, model, N, res):
    """`propagate` (continuous-time propagation of a discrete solution through the nonlinear dynamics,
    discretization.jl:515-541) on the device vs the oracle restatement, 1e-10 relative."""
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=5, iter_max=1)
    B = 5
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    rng = np.random.default_rng(11)
    xs, us, ps = [], [], []
    for b in range(B):
        pp = traj.mdl.nominal_pp() * (1 + 0.1 * rng.uniform(-1, 1, size=traj.mdl.nominal_pp().size))
        x, u, p = traj.guess(N, pp)
        xs.append(x); us.append(u + 0.1 * rng.standard_normal(u.shape)); ps.append(p)
    ref = pkg.SubproblemSolutionBatch(np.stack(xs), np.stack(us), np.stack(ps).reshape(B, -1), pbm)
    tc, xc = pkg.propagate(ref, pbm, res=res)
    assert xc.shape == (B, res, pbm.nx) and tc[0] == 0.0 and abs(tc[-1] - 1.0) < 1e-15
    for b in range(B):
        to, xo = orc.propagate(model, orc.default_params(model), N, ref.xd[b], ref.ud[b], ref.p[b], res=res)
        np.testing.assert_allclose(tc, to, rtol=0, atol=1e-15)
        scale = max(1.0, float(np.abs(xo).max()))
        assert float(np.abs(xc[b] - xo).max()) / scale < TOL
    pbm.close()


def test_propagate_of_converged_solution_hits_the_nodes(pkg):
    """Size-independent property: propagating a converged (dynamically feasible) PTR solution with res = N samples
    reproduces the discrete states at the nodes to about the feasibility tolerance."""
    model, N = "quadrotor", 30
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=15, iter_max=15, wvc=1e3, wtr=0.1, eps_abs=1e-5, eps_rel=1e-4)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=2)
    sol, _ = pkg.PTR.solve(pbm, np.tile(traj.mdl.nominal_pp(), (2, 1)))
    assert sol.feas.all()
    # fine propagation, then sample at the nodes (res - 1 a multiple of N - 1)
    res = 20 * (N - 1) + 1
    tc, xc = pkg.propagate(sol, pbm, res=res)
    err = np.abs(xc[:, ::20, :] - sol.xd) / pbm.scale.Sx
    assert err.max() < 5e-2      # accumulated over N - 1 intervals of defects <= feas_tol = 1e-3 each
    # the continuous-time part of SCPSolution(history) (scp.jl:227-237): xc at 2 Nsub (N - 1) samples, uc first-order hold
    tc2, xc2, uc = pkg.continuous_time(sol, pbm)
    assert xc2.shape == (2, 2 * 15 * (N - 1), pbm.nx) and sol.xc is xc2
    assert np.abs(xc2[:, 0] - sol.xd[:, 0]).max() == 0.0
    assert np.abs((xc2[:, -1] - sol.xd[:, -1]) / pbm.scale.Sx).max() < 5e-2
    for k in (0, 7, N - 1):
        assert np.abs(uc.sample(pbm.t_grid[k]) - sol.ud[:, k]).max() < 1e-14
    mid = 0.5 * (pbm.t_grid[3] + pbm.t_grid[4])
    assert np.abs(uc.sample(mid) - 0.5 * (sol.ud[:, 3] + sol.ud[:, 4])).max() < 1e-12
    pbm.close()


def test_device_pointer_entry_point_matches_host_entry_point(pkg):
    """scp_discretize_batch_dev takes caller-owned DEVICE pointers (allocated here with hipMalloc through ctypes on the
    HIP runtime the library itself uses) and is asynchronous on the handle's stream; results must equal the
    host-pointer entry point bit for bit."""
    import ctypes
    # the HIP runtime instance the library is bound to in THIS process (whichever libamdhip64 got loaded first)
    loaded = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln]
    assert loaded, "libscp_mi355x.so should have pulled in libamdhip64"
    hip = ctypes.CDLL(loaded[0])
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    H2D, D2H = 1, 2
    bufs = []

    def dalloc(nbytes):
        ptr = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(ptr), max(int(nbytes), 8)) == 0
        bufs.append(ptr)
        return ptr

    def to_dev(a):
        a = np.ascontiguousarray(a)
        d = dalloc(a.nbytes)
        assert hip.hipMemcpy(d, a.ctypes.data_as(ctypes.c_void_p), a.nbytes, H2D) == 0
        return d

    def to_host(d, shape, dtype=np.float64):
        out = np.zeros(shape, dtype=dtype)
        assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), d, out.nbytes, D2H) == 0
        return out

    model, N, Nsub, B = "rocket_landing", 24, 8, 6
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=1)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    rng = np.random.default_rng(4)
    xs, us, ps = [], [], []
    for b in range(B):
        pp = traj.mdl.nominal_pp() * (1 + 0.1 * rng.uniform(-1, 1, size=traj.mdl.nominal_pp().size))
        x, u, p = traj.guess(N, pp)
        xs.append(x + 0.02 * rng.standard_normal(x.shape) * (1 + np.abs(x))); us.append(u); ps.append(p)
    ref = pkg.SubproblemSolutionBatch(np.stack(xs), np.stack(us), np.stack(ps).reshape(B, -1), pbm)
    pkg.discretize_(ref, pbm)                                   # host-pointer path
    nx, nu, npF = pbm.nx, pbm.nu, pbm.npF
    xd, ud, p = to_dev(ref.xd), to_dev(ref.ud), to_dev(ref.p)
    shapes = dict(A=(B, N - 1, nx, nx), Bm=(B, N - 1, nu, nx), Bp=(B, N - 1, nu, nx), F=(B, N - 1, max(npF, 1), nx),
                  r=(B, N - 1, nx), E=(B, N - 1, nx, nx), defect=(B, N - 1, nx))
    d = {k: dalloc(8 * int(np.prod(v))) for k, v in shapes.items()}
    feas = dalloc(4 * B)
    L = pkg._lib.lib()
    rc = L.scp_discretize_batch_dev(pbm.handle, B, xd, ud, p, d["A"], d["Bm"], d["Bp"], d["F"], d["r"], d["E"], d["defect"], feas)
    assert rc == 0
    assert L.scp_sync(pbm.handle) == 0                          # asynchronous on the handle's stream
    got = {k: to_host(d[k], shapes[k]) for k in shapes}
    np.testing.assert_array_equal(got["A"], ref.dyn.A)
    np.testing.assert_array_equal(got["Bm"], ref.dyn.B[0])
    np.testing.assert_array_equal(got["Bp"], ref.dyn.B[1])
    np.testing.assert_array_equal(got["F"][:, :, :npF], ref.dyn.F)
    np.testing.assert_array_equal(got["r"], ref.dyn.r)
    np.testing.assert_array_equal(got["E"], ref.dyn.E)
    np.testing.assert_array_equal(got["defect"], ref.defect)
    assert ((to_host(feas, (B,), np.int32) != 0) == ref.feas).all()
    for b_ in bufs:
        hip.hipFree(b_)
    pbm.close()
