"""The oracle reproduces the committed golden fixtures (tests/golden/make_golden.py); host-side model data of the
product (guesses, scaling boxes, parameter blobs) equals the oracle's independent definitions."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODELS = ["double_integrator", "quadrotor", "rocket_landing"]


@pytest.mark.parametrize("model", MODELS)
def test_oracle_reproduces_discretize_golden(orc, model):
    g = np.load(os.path.join(GOLD, "discretize_%s.npz" % model))
    out = orc.discretize(model, orc.default_params(model), int(g["N"]), int(g["Nsub"]), g["xd"], g["ud"], g["p"], g["iSx"], 1e-3)
    for nm in ("A", "Bm", "Bp", "F", "r", "E", "defect"):
        np.testing.assert_allclose(out[nm], g[nm], rtol=1e-12, atol=1e-13)
    assert (out["feas"] == g["feas"]).all()


def test_oracle_reproduces_ptr_golden(orc):
    from oracle import ptr_ref
    g = np.load(os.path.join(GOLD, "ptr_double_integrator.npz"))
    pars = ptr_ref.PTRParameters(int(g["N"]), int(g["Nsub"]), int(g["iters"]), 1e3, 0.1, 0, 0, 1e-3)
    st, hist = ptr_ref.ptr_solve("double_integrator", pars)
    assert st == str(g["status"])
    np.testing.assert_allclose([h["sub"]["J_aug"] for h in hist], g["J_aug"], rtol=1e-7)
    np.testing.assert_allclose(hist[-1]["sol"].xd, g["xd"], atol=1e-6)


@pytest.mark.parametrize("model", MODELS)
def test_host_model_data_matches_oracle(pkg, orc, model):
    from oracle.models import MODELS as OM
    o = OM[model]()
    m = pkg.REGISTRY[model]()
    np.testing.assert_allclose(m.par(), o.par())
    np.testing.assert_allclose(m.nominal_pp(), o.nominal_pp())
    for a, b in zip(m.scale_advice(), o.bbox()):
        np.testing.assert_allclose(np.asarray(a, float).reshape(-1, 2), np.asarray(b, float).reshape(-1, 2))
    for N in (5, 12):
        for a, b in zip(m.guess(N, m.nominal_pp()), o.guess(N, o.nominal_pp())):
            np.testing.assert_allclose(a, b, atol=1e-14)
    info = pkg._lib.ScpModelInfo()
    assert pkg._lib.lib().scp_model_query(pkg.models.MODEL_IDS[model], info) == 0
    assert (info.nx, info.nu, info.np, info.ns, info.nic, info.ntc) == (o.nx, o.nu, o.np, o.ns, o.nic, o.ntc)


@pytest.mark.parametrize("model", ["double_integrator", "quadrotor", "rocket_landing"])
def test_oracle_propagate_reproduces_golden(orc, model):
    g = np.load(os.path.join(GOLD, "propagate_%s.npz" % model))
    N, res = int(g["N"]), int(g["res"])
    for b in range(g["xd"].shape[0]):
        _, xc = orc.propagate(model, orc.default_params(model), N, g["xd"][b], g["ud"][b], g["p"][b], res=res)
        np.testing.assert_allclose(xc, g["xc"][b], rtol=1e-13, atol=1e-13 * max(1.0, float(np.abs(g["xc"][b]).max())))


def test_config_size_goldens_are_optimal_and_consistent(orc):
    """tests/golden/cfg_*.npz: every subproblem of the config-size oracle runs was solved to OPTIMAL (ECOS default
    tolerances), the cost split adds up, the stored literal solution satisfies the reference's penalty definitions, and
    one stored subproblem is reproduced by re-running the oracle (rocket landing N=100 converged regime, ~6 s)."""
    import glob
    from oracle import ptr_ref
    from oracle.models import MODELS as OM
    files = sorted(glob.glob(os.path.join(GOLD, "cfg_*.npz")))
    assert len(files) >= 6
    for f in files:
        g = np.load(f)
        assert str(g["status"]) == "SCP_SOLVED" and all(str(s) == "OPTIMAL" for s in g["ipm_status"])
        np.testing.assert_allclose(g["J"] + g["J_tr"] + g["J_vc"], g["J_aug"], rtol=1e-12)
        assert bool(g["feas"][-1])
        N = int(g["N"])
        w = np.full(N, 1.0 / (N - 1)); w[0] = w[-1] = 0.5 / (N - 1)
        for it in (1, 4, 12):
            pre = "s%d_" % it
            Jvc = g[pre + "cost"][2]
            assert abs(1e3 * (w @ g[pre + "P"] + g[pre + "Pf"].sum()) - Jvc) <= 1e-6 * max(1.0, Jvc)
    g = np.load(os.path.join(GOLD, "cfg_rocket_landing_N100_nom.npz"))
    mdl = OM["rocket_landing"]()
    pars = ptr_ref.PTRParameters(100, 15, 15, 1e3, 0.1, 0, 0, 1e-3)
    scale = ptr_ref.Scaling(*mdl.bbox())
    ref = ptr_ref.discretize(mdl, pars, scale, g["s12_ref_x"], g["s12_ref_u"], g["s12_ref_p"])
    sub = ptr_ref.solve_subproblem(mdl, pars, scale, ref, g["pp"])
    assert sub["status"] == "OPTIMAL"
    assert abs(sub["J_aug"] - g["s12_cost"][3]) <= 1e-8 * abs(g["s12_cost"][3])
    assert np.abs((sub["u"] - g["s12_u"]) / scale.Su).max() <= 1e-6
