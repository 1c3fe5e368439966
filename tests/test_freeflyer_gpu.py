"""6-DoF free-flyer on the MI355X (-m gpu): discretize! with a 13-dimensional state-dependent Jacobian and the first real
integration action (quaternion renormalisation, freeflyer/definition.jl:69-82), propagate; the subproblem side is refused
(np = 1 + 6N in the reference, csrc/models/freeflyer.hpp)."""
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
    assert pbm.info.has_subproblem == 0 and pbm.info.structured == 0 and (pbm.nx, pbm.nu, pbm.np) == (13, 6, 1)
    ref = pkg.SubproblemSolutionBatch(xs, us, ps, pbm)
    pkg.discretize_(ref, pbm)
    o = orc.discretize("freeflyer", orc.default_params("freeflyer"), N, Nsub, xs, us, ps, pbm.scale.iSx, pars.feas_tol)
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


def test_propagate_matches_oracle(pkg, orc):
    N, B, res = 20, 2, 101
    traj, xs, us, ps = _batch(pkg, N, B, seed=3)
    pbm = pkg.PTR.create(pkg.PTR.Parameters(N=N, Nsub=5, iter_max=1), traj, batch_capacity=B)
    ref = pkg.SubproblemSolutionBatch(xs, us, ps, pbm)
    tc, xc = pkg.propagate(ref, pbm, res=res)
    for b in range(B):
        to, xo = orc.propagate("freeflyer", orc.default_params("freeflyer"), N, xs[b], us[b], ps[b], res=res)
        assert np.abs(xc[b] - xo).max() / max(1.0, np.abs(xo).max()) < 1e-10
    assert np.abs(np.linalg.norm(xc[:, :, 6:10], axis=2) - 1.0).max() < 1e-12
    pbm.close()


def test_device_guess_matches_the_oracle_restatement(pkg):
    """traj.guess on the device (scp_guess_batch_host): axis-by-axis path at constant speed, SLERP attitude, constant body
    rate (freeflyer/definition.jl:84-186) for a Monte-Carlo batch of boundary conditions, against oracle/models.py; and the
    straight-line guesses of a structured and an unstructured model against the host mirrors."""
    om = MODELS["freeflyer"]()
    rng = np.random.default_rng(2)
    N, B = 50, 4
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
    for model in ("quadrotor", "starship"):
        traj = pkg.TrajectoryProblem(model)
        pbm = pkg.PTR.create(pkg.PTR.Parameters(N=20, Nsub=5, iter_max=1), traj, batch_capacity=2)
        pp = np.stack([traj.mdl.nominal_pp(), traj.mdl.nominal_pp() * 1.05])
        xd, ud, p = pkg.device_guess(pbm, pp)
        for b in range(2):
            xh, uh, ph = traj.guess(20, pp[b])
            assert np.abs(xd[b] - xh).max() < 1e-12 * max(1.0, np.abs(xh).max()) and np.abs(ud[b] - uh).max() < 1e-9 * max(1.0, np.abs(uh).max())
            assert np.abs(p[b] - ph).max() < 1e-12 * max(1.0, np.abs(ph).max())
        pbm.close()


def test_the_subproblem_side_is_refused(pkg):
    traj = pkg.TrajectoryProblem("freeflyer")
    pbm = pkg.PTR.create(pkg.PTR.Parameters(N=10, Nsub=5, iter_max=2), traj, batch_capacity=1)
    with pytest.raises((pkg._lib.ScpError, NotImplementedError)):
        pkg.PTR.solve(pbm, traj.mdl.nominal_pp()[None])
    pbm.close()
    with pytest.raises((pkg._lib.ScpError, NotImplementedError)):
        pkg.SCvx.create(pkg.SCvx.Parameters(N=10, Nsub=5, iter_max=2, lam=30.0, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0,
                                            beta_gr=2.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0), traj, batch_capacity=1)
