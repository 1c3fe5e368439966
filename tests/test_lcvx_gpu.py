"""The reference's lossless-convexification double integrator through the C-ABI conic seam on the MI355X (-m gpu): both parameter
choices of test/examples/double_integrator/tests.jl:25-45 as ONE batch of `socp_solve_batch` (same pattern, different right-hand
sides) -- the program of `solve_lcvx` (definition.jl:38-118) an unmodified `ConicProgram` would hand to `pars.solver`.  The device
solution is held against the reference's own known answer, the maximum-principle trajectory (committed record of the restated
shooting search, tests/golden/make_lcvx_golden.py; tolerances of tests/test_lcvx_cpu.py), and against the oracle's solution of the
same program (optimal value 1e-8, trajectory 1e-6)."""
import numpy as np
import pytest

from oracle import lcvx_ref as L
from test_lcvx_cpu import GOLD, check_against_mp, golden_mp

pytestmark = pytest.mark.gpu


def test_lcvx_double_integrator_batch_matches_the_maximum_principle(pkg):
    mdls = [L.DoubleIntegratorParameters(ch) for ch in (1, 2)]
    Ps = [L.lcvx_program(m) for m in mdls]
    # one pattern: G, A, c, h are the same for both choices (T, N equal); b carries the friction (w) and the travel distance
    assert (Ps[0]["G"] != Ps[1]["G"]).nnz == 0 and (Ps[0]["A"] != Ps[1]["A"]).nnz == 0
    c = np.stack([P["c"] for P in Ps]); h = np.stack([P["h"] for P in Ps]); b = np.stack([P["b"] for P in Ps])
    x, y, s, z, st = pkg.conic.socp_solve_batch(c, Ps[0]["G"], h, Ps[0]["l"], Ps[0]["q"], A=Ps[0]["A"], b=b)
    assert (st == 0).all(), st                      # OPTIMAL: what the reference's test asserts (definition.jl:101-104)
    for i, ch in enumerate((1, 2)):
        P, mdl = Ps[i], mdls[i]
        cmp_ = check_against_mp(mdl, x[i], golden_mp(ch))
        pc = float(P["c"] @ x[i])
        assert pc == pytest.approx(float(GOLD["c%d_lcvx_pcost" % ch]), rel=1e-8)
        np.testing.assert_allclose(x[i], GOLD["c%d_lcvx_x" % ch], atol=1e-6)
        # solver-independent certificate of the device solution
        assert np.linalg.norm(P["A"] @ x[i] - P["b"]) < 1e-7 and np.linalg.norm(P["G"] @ x[i] + s[i] - P["h"]) < 1e-7
        assert np.linalg.norm(P["A"].T @ y[i] + P["G"].T @ z[i] + P["c"]) < 1e-7 and abs(s[i] @ z[i]) < 1e-6
        print("LCvx double integrator choice %d on the device vs maximum principle: %s" % (ch, cmp_))
