"""Generic conic solver on the MI355X (-m gpu), through the C ABI of include/scp_conic.h: parity with the independent
restatement oracle/ipm.py on random programs, the committed golden PTR conic programs at the config sizes (as one
batch), HiGHS on LPs, certificates, the one-shot `socp_solve_batch`, error behaviour, and a full-size property test."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import linprog

from oracle import ipm
from test_conic_cpu import random_socp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_random_socps_match_oracle_ipm(pkg):
    rng = np.random.default_rng(1)
    for trial in range(6):
        q = [(4, 3, 5), (3,), (), (6, 6)][trial % 4]
        c, G, h, l, q, A, b = random_socp(rng, n=10 + trial, pe=trial % 4, l=5 + trial, q=q)
        P = sp.diags(rng.uniform(0.1, 1.0, c.size)) if trial % 2 else None
        r0 = ipm.solve(c, G, h, l, q, A, b, P=P)
        prog = pkg.conic.ConicProgramBatch(c.size, G, l, q, A=A if A.shape[0] else None, P=P, batch_capacity=1)
        r1 = prog.solve(c[None], h[None], b=b[None] if A.shape[0] else None)
        prog.close()
        assert r0["status"] == "OPTIMAL" and r1["status"][0] == 0
        assert abs(r0["pcost"] - r1["pcost"][0]) <= 1e-8 * max(1.0, abs(r0["pcost"]))
        if P is not None:
            np.testing.assert_allclose(r1["x"][0], r0["x"], atol=1e-7)
        assert r1["pres"][0] < 1e-8 and r1["dres"][0] < 1e-8
        assert abs(int(r1["iters"][0]) - r0["iters"]) <= 1


@pytest.mark.parametrize("name", ["conic_quadrotor_N50", "conic_rocket_landing_N100"])
def test_golden_ptr_conic_programs_as_a_batch(pkg, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n, l, q = int(g["n"]), int(g["l"]), list(g["q"])
    m = l + sum(q)
    p = g["b"].shape[1]
    G = sp.csc_matrix((np.ones(len(g["Gi"])), g["Gi"], g["Gp"]), shape=(m, n))
    A = sp.csc_matrix((np.ones(len(g["Ai"])), g["Ai"], g["Ap"]), shape=(p, n))
    P = sp.csc_matrix((np.ones(len(g["Pi"])), g["Pi"], g["Pp"]), shape=(n, n))
    reps = 22            # 66 problems: more than one wavefront, ragged last wave
    prog = pkg.conic.ConicProgramBatch(n, G, l, q, A=A, P=P, batch_capacity=3 * reps)
    tile = lambda a: np.tile(a, (reps, 1))
    r = prog.solve(tile(g["c"]), tile(g["h"]), b=tile(g["b"]), Gx=tile(g["Gx"]), Ax=tile(g["Ax"]), Px=tile(g["Px"]))
    st = prog.stats()
    prog.close()
    assert (r["status"] == 0).all(), r["status"]
    pc, x, it = tile(g["pcost"][:, None])[:, 0], tile(g["x"]), tile(g["iters"][:, None])[:, 0]
    assert np.all(np.abs(r["pcost"] - pc) <= 1e-8 * np.maximum(1.0, np.abs(pc)))
    assert np.abs(r["x"] - x).max() < 5e-5      # flat optimal faces of the L1/Linf epigraphs: x is gap-limited
    assert np.all(np.abs(r["iters"] - it) <= 1)
    # identical inputs -> bitwise identical outputs across lanes / waves
    assert np.array_equal(r["x"][:3], r["x"][3:6]) and np.array_equal(r["x"][:3], r["x"][-3:])
    assert st["kkt_dim"] == n + p + m and st["nnzL"] > 0


def test_lp_matches_highs_and_shared_arrays(pkg):
    rng = np.random.default_rng(3)
    n, m, B = 15, 30, 70
    G = rng.standard_normal((m, n))
    x0 = rng.standard_normal(n)
    h = G @ x0 + rng.uniform(0.1, 1.0, m)
    cs = np.stack([-(G.T @ rng.uniform(0.1, 1.0, m)) for _ in range(B)])
    prog = pkg.conic.ConicProgramBatch(n, sp.csc_matrix(G), m, [], batch_capacity=B)
    r = prog.solve(cs, h, shared=("h",))       # G values default to the pattern matrix (shared), h shared, c per problem
    prog.close()
    assert (r["status"] == 0).all()
    for t in range(0, B, 7):
        ref = linprog(cs[t], A_ub=G, b_ub=h, bounds=[(None, None)] * n, method="highs")
        assert abs(ref.fun - r["pcost"][t]) <= 1e-7 * max(1.0, abs(ref.fun))


def test_certificates_and_mixed_status_batch(pkg):
    """one batch, one pattern: feasible, infeasible and unbounded members keep their own status"""
    G = sp.csc_matrix(np.array([[1.0], [-1.0]]))
    prog = pkg.conic.ConicProgramBatch(1, G, 2, [], batch_capacity=3)
    c = np.array([[1.0], [1.0], [-1.0]])
    h = np.array([[2.0, 1.0],      # -1 <= x <= 2, min x  -> -1
                  [-1.0, -1.0],    # x <= -1, x >= 1      -> infeasible
                  [2.0, 1.0]])
    Gx = np.array([[1.0, -1.0], [1.0, -1.0], [0.0, -1.0]])   # third: only x >= -1, min -x -> unbounded
    r = prog.solve(c, h, Gx=Gx)
    prog.close()
    assert list(r["status"]) == [0, 4, 5]
    assert abs(r["x"][0, 0] + 1.0) < 1e-7


def test_one_shot_socp_solve_batch(pkg):
    rng = np.random.default_rng(5)
    c, G, h, l, q, A, b = random_socp(rng)
    B = 4
    cs = np.stack([c * (1 + 0.05 * rng.standard_normal(c.size)) for _ in range(B)])
    x, y, s, z, st = pkg.conic.socp_solve_batch(cs, G, np.tile(h, (B, 1)), l, q, A=A, b=np.tile(b, (B, 1)))
    assert (st == 0).all()
    for t in range(B):
        r0 = ipm.solve(cs[t], G, h, l, q, A, b)
        assert abs(cs[t] @ x[t] - r0["pcost"]) <= 1e-8 * max(1.0, abs(r0["pcost"]))
        # KKT certificate, solver independent
        assert np.linalg.norm(A @ x[t] - b) < 1e-7 and np.linalg.norm(G @ x[t] + s[t] - h) < 1e-7
        assert np.linalg.norm(A.T @ y[t] + G.T @ z[t] + cs[t]) < 1e-7 and abs(s[t] @ z[t]) < 1e-6


def test_error_behaviour(pkg):
    G = sp.csc_matrix(np.ones((3, 2)))
    with pytest.raises(pkg._lib.ScpError) as e:
        pkg.conic.ConicProgramBatch(2, G, 3, [0], batch_capacity=1)        # wrong cone size: l + sum(q) != m is caught host side
    prog = pkg.conic.ConicProgramBatch(2, G, 3, [], batch_capacity=2)
    with pytest.raises(pkg._lib.ScpError) as e:
        prog.solve(np.zeros((3, 2)), np.ones((3, 3)))                        # B > batch_capacity
    assert e.value.code == 6
    with pytest.raises(ValueError):
        prog.solve(np.zeros((2, 5)), np.ones((2, 3)))                        # wrong length
    prog.close()


def test_full_size_batch_properties(pkg):
    """4096 rocket-landing N=100 conic programs (the golden program scaled per problem): every member reaches OPTIMAL,
    cost scales linearly with the cost scaling (x is invariant), KKT residuals certify each solution."""
    g = np.load(os.path.join(GOLD, "conic_rocket_landing_N100.npz"))
    n, l, q = int(g["n"]), int(g["l"]), list(g["q"])
    m = l + sum(q)
    p = g["b"].shape[1]
    G = sp.csc_matrix((g["Gx"][1], g["Gi"], g["Gp"]), shape=(m, n))
    A = sp.csc_matrix((g["Ax"][1], g["Ai"], g["Ap"]), shape=(p, n))
    P = sp.csc_matrix((g["Px"][1], g["Pi"], g["Pp"]), shape=(n, n))
    B = 4096
    prog = pkg.conic.ConicProgramBatch(n, G, l, q, A=A, P=P, batch_capacity=B)
    scale = 1.0 + (np.arange(B) % 8) / 8.0
    r = prog.solve(g["c"][1][None] * scale[:, None], g["h"][1], b=g["b"][1], Px=g["Px"][1][None] * scale[:, None],
                   shared=("h", "b"))
    prog.close()
    assert (r["status"] == 0).all()
    assert np.abs(r["pcost"] / scale - g["pcost"][1]).max() <= 1e-7 * max(1.0, abs(g["pcost"][1]))
    assert r["pres"].max() < 1e-8 and r["dres"].max() < 1e-8
    assert np.array_equal(r["x"][0], r["x"][8])       # same data, different wave -> identical bits
