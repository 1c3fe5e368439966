"""Golden record of the reference's lossless-convexification double integrator (the only known answers the reference's tests hold on
the conic seam): the maximum-principle shooting solution (oracle/lcvx_ref.py::solve_mp, a restatement of
test/examples/double_integrator/definition.jl:137-294 -- needs no solver) and the oracle interior-point solution of the LCvx conic
program (definition.jl:38-118) for both parameter choices.  python tests/golden/make_lcvx_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ipm, lcvx_ref as L  # noqa: E402

out = {}
for ch in (1, 2):
    mdl = L.DoubleIntegratorParameters(ch)
    mp = L.solve_mp(mdl)
    P = L.lcvx_program(mdl)
    r = ipm.solve(P["c"], P["G"], P["h"], P["l"], P["q"], A=P["A"], b=P["b"])
    assert r["status"] == ipm.OPTIMAL, r["status"]
    cmp_ = L.compare_with_mp(mdl, r["x"], mp)
    print(ch, mp["c"], mp["ts"], mp["err"], r["pcost"], cmp_)
    p = "c%d_" % ch
    out.update({p + "mp_t": mp["t"], p + "mp_x": mp["x"], p + "mp_u": mp["u"], p + "mp_c": mp["c"], p + "mp_ts": mp["ts"], p + "mp_err": mp["err"],
                p + "mp_iterations": mp["iterations"], p + "lcvx_x": r["x"], p + "lcvx_pcost": r["pcost"], p + "lcvx_iters": r["iters"],
                p + "A": mdl.A, p + "Bm": mdl.Bm, p + "Bp": mdl.Bp, p + "w": mdl.w})
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lcvx_double_integrator.npz"), **out)

# LCvx 3-DoF rocket landing (test/examples/rocket_landing/definition.jl:33-150) at fixed times of flight inside the range the
# reference's golden-section search brackets (tests.jl:28-32): oracle optimal costs
R = L.Rocket()
pdg = {}
for tf in (75.0, 80.0, 90.0):
    P = L.pdg_program(R, tf)
    r = ipm.solve(P["c"], P["G"], P["h"], P["l"], P["q"], A=P["A"], b=P["b"])
    assert r["status"] == ipm.OPTIMAL
    pdg["tf%d_cost" % tf] = r["pcost"] + P["cost_const"]; pdg["tf%d_x" % tf] = r["x"]; pdg["tf%d_N" % tf] = P["N"]
    print("pdg", tf, P["N"], pdg["tf%d_cost" % tf], r["iters"])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lcvx_rocket_landing.npz"), **pdg)
