"""HIP path against the committed golden fixtures (tests/golden/, generated from the oracle by make_golden.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODELS = ["double_integrator", "quadrotor", "rocket_landing"]


@pytest.mark.parametrize("model", MODELS)
def test_discretize_against_golden(pkg, model):
    g = np.load(os.path.join(GOLD, "discretize_%s.npz" % model))
    N, Nsub = int(g["N"]), int(g["Nsub"])
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=1, feas_tol=1e-3)
    B = g["xd"].shape[0]
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    np.testing.assert_allclose(pbm.scale.iSx, g["iSx"])
    ref = pkg.SubproblemSolutionBatch(g["xd"], g["ud"], g["p"], pbm)
    pkg.discretize_(ref, pbm)
    got = dict(A=ref.dyn.A, Bm=ref.dyn.B[0], Bp=ref.dyn.B[1], F=ref.dyn.F, r=ref.dyn.r, E=ref.dyn.E, defect=ref.defect)
    for nm, v in got.items():
        if g[nm].size == 0:
            continue
        sc = max(1.0, float(np.abs(g[nm]).max()))
        assert np.abs(v - g[nm]).max() <= 1e-10 * sc, nm
    assert (ref.feas == g["feas"]).all()
    pbm.close()


@pytest.mark.parametrize("model", MODELS)
def test_ptr_against_golden(pkg, model):
    g = np.load(os.path.join(GOLD, "ptr_%s.npz" % model))
    N, Nsub, iters = int(g["N"]), int(g["Nsub"]), int(g["iters"])
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=Nsub, iter_max=iters, wvc=1e3, wtr=0.1, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=1)
    sol, h = pkg.PTR.solve(pbm)
    assert sol.status[0] == str(g["status"])
    k = 1.0   # one stated tolerance for every model
    s = pbm.scale
    assert np.abs((sol.xd[0] - g["xd"]) / s.Sx).max() <= k * 1e-4
    assert np.abs((sol.ud[0] - g["ud"]) / s.Su).max() <= k * 1e-4
    assert abs(sol.J[0] - g["J"][-1]) <= k * 1e-6 * max(1.0, abs(g["J"][-1]))
    assert bool(sol.feas[0]) == bool(g["feas"][-1])
    pbm.close()


@pytest.mark.parametrize("model", ["double_integrator", "quadrotor", "rocket_landing"])
def test_propagate_against_golden(pkg, model):
    g = np.load(os.path.join(GOLD, "propagate_%s.npz" % model))
    N, res = int(g["N"]), int(g["res"])
    traj = pkg.TrajectoryProblem(model)
    pars = pkg.PTR.Parameters(N=N, Nsub=5, iter_max=1)
    B = g["xd"].shape[0]
    pbm = pkg.PTR.create(pars, traj, batch_capacity=B)
    ref = pkg.SubproblemSolutionBatch(g["xd"], g["ud"], g["p"], pbm)
    _, xc = pkg.propagate(ref, pbm, res=res)
    scale = max(1.0, float(np.abs(g["xc"]).max()))
    assert float(np.abs(xc - g["xc"]).max()) / scale < 1e-10
    pbm.close()
