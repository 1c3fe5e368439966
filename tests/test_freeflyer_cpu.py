"""Oracle pins for the 6-DoF free-flyer (test/examples/freeflyer): the reference ships no golden data, so the C
restatement of its dynamics / Jacobians / integration action (oracle/scp_oracle.c) and the restatements of its initial
guess are pinned on mathematics (finite differences, quaternion identities, closed forms) and against each other."""
import numpy as np

from oracle.models import MODELS


def _state(rng):
    x = rng.standard_normal(13)
    x[6:10] /= np.linalg.norm(x[6:10])
    return x, 1e-2 * rng.standard_normal(6), np.array([130.0])


def test_jacobians_match_finite_differences(orc):
    par = orc.default_params("freeflyer")
    rng = np.random.default_rng(0)
    for _ in range(3):
        x, u, p = _state(rng)
        f, A, B, F = orc.model_eval("freeflyer", par, 0.3, 2, x, u, p)
        eps = 1e-6
        fd = lambda g, v, j: (g(v + eps * np.eye(v.size)[j]) - g(v - eps * np.eye(v.size)[j])) / (2 * eps)
        fx = lambda xx: orc.model_eval("freeflyer", par, 0.3, 2, xx, u, p)[0]
        fu = lambda uu: orc.model_eval("freeflyer", par, 0.3, 2, x, uu, p)[0]
        fp = lambda pq: orc.model_eval("freeflyer", par, 0.3, 2, x, u, pq)[0]
        Afd = np.stack([fd(fx, x, j) for j in range(13)], axis=1)
        Bfd = np.stack([fd(fu, u, j) for j in range(6)], axis=1)
        assert np.abs(A - Afd).max() < 1e-6 * max(1.0, np.abs(A).max())
        assert np.abs(B - Bfd).max() < 1e-6 * max(1.0, np.abs(B).max())
        assert np.abs(F[:, 0] - fd(fp, p, 0)).max() < 1e-7
        # quaternion kinematics keep the norm: q . q' = 0 ; torque-free rigid body keeps the kinetic energy
        assert abs(x[6:10] @ f[6:10]) < 1e-12 * p[0]


def test_discretize_applies_the_quaternion_action_and_the_translation_is_a_double_integrator(orc):
    """the propagated state V[x] = x_{k+1} - defect_k has a unit quaternion (action after every RK4 step,
    freeflyer/definition.jl:69-82); the (r, v) block of A_k is [[I, dt I], [0, I]] and r, v see only T / m."""
    mdl = MODELS["freeflyer"]()
    N, Nsub = 12, 8
    x, u, p = mdl.guess(N, mdl.nominal_pp())
    rng = np.random.default_rng(1)
    u = u + 1e-3 * rng.standard_normal(u.shape)
    x[:, 10:13] += 1e-3 * rng.standard_normal((N, 3))
    iSx = 1.0 / (mdl.bbox()[0][:, 1] - mdl.bbox()[0][:, 0])
    o = orc.discretize("freeflyer", mdl.par(), N, Nsub, x[None], u[None], p[None], iSx, 1e-3)
    prop = x[1:] - o["defect"][0]
    assert np.abs(np.linalg.norm(prop[:, 6:10], axis=1) - 1.0).max() < 1e-13
    dt = p[0] / (N - 1)
    A0 = o["A"][0, 0].T
    np.testing.assert_allclose(A0[0:3, 3:6], dt * np.eye(3), atol=1e-12)
    np.testing.assert_allclose(A0[0:6, 6:13], 0.0, atol=1e-14)
    Bm0 = o["Bm"][0, 0].T
    np.testing.assert_allclose(Bm0[3:6, 0:3], dt / 2 / mdl.par()[0] * np.eye(3), rtol=1e-12)     # int sigma- dt = dt / 2


def test_guess_restatements_agree_and_interpolate_the_boundary_attitudes(pkg):
    om = MODELS["freeflyer"]()
    pm = pkg.REGISTRY["freeflyer"]()
    pp = om.nominal_pp()
    np.testing.assert_allclose(pm.nominal_pp(), pp, atol=1e-15)
    for N in (10, 50):
        xo, uo, po = om.guess(N, pp)
        xp, up, pq = pm.guess(N, pp)
        np.testing.assert_allclose(xp, xo, atol=1e-12)
        assert po[0] == pq[0] == 130.0 and not uo.any() and not up.any()
        np.testing.assert_allclose(np.linalg.norm(xo[:, 6:10], axis=1), 1.0, atol=1e-13)
        np.testing.assert_allclose(xo[0, 6:10], pp[6:10], atol=1e-13)
        np.testing.assert_allclose(xo[-1, 6:10], pp[19:23], atol=1e-13)      # SLERP ends at q_f
        np.testing.assert_allclose(xo[0, 0:3], pp[0:3], atol=1e-13)
        np.testing.assert_allclose(xo[-1, 0:3], pp[13:16], atol=1e-12)
        # constant-speed L1 path: every node moves along exactly one axis
        assert ((np.abs(xo[:, 3:6]) > 0).sum(axis=1) <= 1).all()
