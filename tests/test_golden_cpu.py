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
