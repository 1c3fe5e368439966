"""Generic conic solver, CPU side (-m "not gpu"): the HOST build of the product's solver sources
(oracle/_build/libconic_host.so <- scptoolbox.jl_amd/csrc/conic_{symbolic,ipm}.hpp) against the independent
restatement oracle/ipm.py, scipy's HiGHS, closed forms and the committed golden conic programs.  The device kernel
runs the same solver body (tests/test_conic_gpu.py checks it through the C ABI)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import linprog

from oracle import conic_host, ipm

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def random_socp(rng, n=12, pe=3, l=8, q=(4, 3, 5), dens=0.4):
    m = l + sum(q)
    G = sp.random(m, n, dens, random_state=int(rng.integers(1 << 30)), data_rvs=rng.standard_normal).tocsc()
    A = sp.random(pe, n, 0.6, random_state=int(rng.integers(1 << 30)), data_rvs=rng.standard_normal).tocsc()
    x0 = rng.standard_normal(n)
    K = ipm.Cone(l, list(q))
    s0 = K.e()
    s0[:l] = rng.uniform(0.5, 2, l)
    for o, d in zip(K.offs, K.q):
        s0[o + 1:o + d] = 0.3 * rng.standard_normal(d - 1)
        s0[o] = np.linalg.norm(s0[o + 1:o + d]) + 1
    h = G @ x0 + s0
    b = A @ x0
    c = -(A.T @ rng.standard_normal(pe) + G.T @ s0)   # dual feasible by construction
    return c, G, h, l, list(q), A, b


@pytest.fixture(scope="module", autouse=True)
def _build(orc):
    return orc


def test_random_socps_match_oracle_ipm():
    rng = np.random.default_rng(1)
    for trial in range(6):
        q = [(4, 3, 5), (3,), (), (6, 6)][trial % 4]
        c, G, h, l, q, A, b = random_socp(rng, n=10 + trial, pe=trial % 4, l=5 + trial, q=q)
        P = sp.diags(rng.uniform(0.1, 1.0, c.size)) if trial % 2 else None
        r0 = ipm.solve(c, G, h, l, q, A, b, P=P)
        r1 = conic_host.solve(c, G, h, l, q, A if A.shape[0] else None, b, P=P)
        assert r0["status"] == "OPTIMAL" and r1["status"] == 0
        assert abs(r0["pcost"] - r1["pcost"]) <= 1e-8 * max(1.0, abs(r0["pcost"]))
        if P is not None:   # strictly convex: unique minimiser (the LP/SOCP instances may have optimal faces)
            np.testing.assert_allclose(r1["x"], r0["x"], atol=1e-7)
        assert r1["pres"] < 1e-8 and r1["dres"] < 1e-8
        assert abs(int(r1["iters"]) - r0["iters"]) <= 1   # same algorithm, same path


def test_offdiagonal_quadratic_cost():
    rng = np.random.default_rng(2)
    c, G, h, l, q, A, b = random_socp(rng, n=9, pe=2, l=6, q=(3,))
    M = rng.standard_normal((9, 9))
    P = sp.csc_matrix(M @ M.T * 0.1)
    r0 = ipm.solve(c, G, h, l, q, A, b, P=P)
    r1 = conic_host.solve(c, G, h, l, q, A, b, P=P)     # upper triangle is extracted by the binding
    assert r1["status"] == 0
    np.testing.assert_allclose(r1["x"], r0["x"], atol=1e-7)


def test_lp_matches_highs():
    rng = np.random.default_rng(3)
    n, m = 15, 30
    G = rng.standard_normal((m, n))
    x0 = rng.standard_normal(n)
    h = G @ x0 + rng.uniform(0.1, 1.0, m)
    c = G.T @ rng.uniform(0.1, 1.0, m) * -1.0
    ref = linprog(c, A_ub=G, b_ub=h, bounds=[(None, None)] * n, method="highs")
    r = conic_host.solve(c, sp.csc_matrix(G), h, m, [])
    assert ref.status == 0 and r["status"] == 0
    assert abs(ref.fun - r["pcost"]) <= 1e-7 * max(1.0, abs(ref.fun))


def test_closed_form_soc_projection():
    """min t s.t. ||x - a|| <= t, x in a box face: distance from a to the hyperplane x_0 = 0 is |a_0|."""
    a = np.array([2.0, -1.0, 0.5])
    # variables (t, x0, x1, x2); equality x0 = 0; SOC (t, x - a)
    c = np.array([1.0, 0, 0, 0])
    A = sp.csc_matrix(np.array([[0.0, 1, 0, 0]])); b = np.zeros(1)
    G = -sp.eye(4, format="csc"); h = np.concatenate([[0.0], -a])
    r = conic_host.solve(c, G, h, 0, [4], A, b)
    assert r["status"] == 0
    assert abs(r["pcost"] - 2.0) < 1e-7
    np.testing.assert_allclose(r["x"], [2.0, 0.0, -1.0, 0.5], atol=1e-6)


def test_infeasible_and_unbounded_certificates():
    # x <= -1 and x >= 1
    G = sp.csc_matrix(np.array([[1.0], [-1.0]])); h = np.array([-1.0, -1.0])
    r = conic_host.solve(np.array([1.0]), G, h, 2, [])
    assert r["status"] == 4
    # min -x s.t. x >= 0: unbounded (what compute_scaling tolerates as DUAL_INFEASIBLE, scp.jl:467-473)
    r = conic_host.solve(np.array([-1.0]), sp.csc_matrix(np.array([[-1.0]])), np.array([0.0]), 1, [])
    assert r["status"] == 5


def test_batch_with_shared_arrays_and_perm():
    rng = np.random.default_rng(4)
    c, G, h, l, q, A, b = random_socp(rng)
    B = 5
    cs = np.stack([c * (1 + 0.05 * rng.standard_normal(c.size)) for _ in range(B)])
    hs = np.stack([h + 0.05 * np.abs(rng.standard_normal(h.size)) for _ in range(B)])
    Gc = sp.csc_matrix(G); Gc.sort_indices()
    r = conic_host.solve(c, G, h, l, q, A, b, B=B, values=dict(c=cs, h=hs), shared_mask=8 | 16 | 2 | 32)
    for t in range(B):
        r0 = ipm.solve(cs[t], G, hs[t], l, q, A, b)
        assert r["status"][t] == 0
        np.testing.assert_allclose(r["x"][t], r0["x"], atol=1e-7)
    # a user-supplied ordering (cone rows first, then equalities, then the variables) gives the same optimum
    nk = c.size + A.shape[0] + G.shape[0]
    r2 = conic_host.solve(c, G, h, l, q, A, b, perm=np.arange(nk)[::-1].copy())
    r3 = conic_host.solve(c, G, h, l, q, A, b)
    np.testing.assert_allclose(r2["x"], r3["x"], atol=1e-8)
    assert r2["stats"][0] >= r3["stats"][0] * 0.5


@pytest.mark.parametrize("name", ["conic_quadrotor_N50", "conic_rocket_landing_N100"])
def test_golden_ptr_conic_programs(name):
    """The literal PTR conic programs at the config sizes (tests/golden/make_conic_golden.py): same optimum and the
    same iteration count as the oracle's IPM, as ONE batch (shared pattern, per-problem values)."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n, l, q = int(g["n"]), int(g["l"]), list(g["q"])
    m = l + sum(q)
    p = len(g["Ap"]) and g["b"].shape[1]
    G = sp.csc_matrix((g["Gx"][0], g["Gi"], g["Gp"]), shape=(m, n))
    A = sp.csc_matrix((g["Ax"][0], g["Ai"], g["Ap"]), shape=(p, n))
    P = sp.csc_matrix((g["Px"][0], g["Pi"], g["Pp"]), shape=(n, n))
    r = conic_host.solve(g["c"][0], G, g["h"][0], l, q, A, g["b"][0], P=P, B=3,
                         values=dict(c=g["c"], h=g["h"], b=g["b"], Gx=g["Gx"], Ax=g["Ax"], Px=g["Px"]))
    assert (r["status"] == 0).all(), r["status"]
    assert np.all(np.abs(r["pcost"] - g["pcost"]) <= 1e-8 * np.maximum(1.0, np.abs(g["pcost"])))
    assert np.abs(r["x"] - g["x"]).max() < 5e-5   # optimal faces of the L1/Linf epigraphs are flat: x is gap-limited
    assert np.all(np.abs(r["iters"] - g["iters"]) <= 1)


@pytest.mark.parametrize("name,min_gain", [("conic_quadrotor_N50", 5.0), ("conic_rocket_landing_N100", 8.0)])
def test_nested_dissection_schedule(name, min_gain, monkeypatch):
    """conic_symbolic.hpp nd_ranks: the chain of node blocks is dissected at its equality rows -- same optimum, same
    iteration count, several times fewer elimination levels (= workgroup barriers of the device kernel), modest fill."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n, l, q = int(g["n"]), int(g["l"]), list(g["q"])
    m = l + sum(q)
    p = g["b"].shape[1]
    G = sp.csc_matrix((g["Gx"][0], g["Gi"], g["Gp"]), shape=(m, n))
    A = sp.csc_matrix((g["Ax"][0], g["Ai"], g["Ap"]), shape=(p, n))
    P = sp.csc_matrix((g["Px"][0], g["Pi"], g["Pp"]), shape=(n, n))
    vals = dict(c=g["c"], h=g["h"], b=g["b"], Gx=g["Gx"], Ax=g["Ax"], Px=g["Px"])
    out = {}
    for mode in ("seq", "nd", "auto"):
        monkeypatch.setenv("CONIC_HOST_ORDER", mode)
        out[mode] = conic_host.solve(g["c"][0], G, g["h"][0], l, q, A, g["b"][0], P=P, B=3, values=vals)
    s_, d_ = out["seq"], out["nd"]
    assert s_["stats"][4] == 0 and d_["stats"][4] >= 5                      # dissection depth
    assert d_["stats"][5] * min_gain <= s_["stats"][5]                      # elimination levels
    assert d_["stats"][1] <= 1.5 * s_["stats"][1] and d_["stats"][0] <= 1.25 * s_["stats"][0]   # multiply-adds, nnz(L)
    for r in (d_, out["auto"]):
        assert (r["status"] == 0).all() and np.array_equal(r["iters"], s_["iters"])
        assert np.all(np.abs(r["pcost"] - g["pcost"]) <= 1e-8 * np.maximum(1.0, np.abs(g["pcost"])))
        assert np.abs(r["x"] - g["x"]).max() < 5e-5
    assert out["auto"]["fallback"] == 0


def _random_chain(rng, N, nx, nu, soc):
    """a random time-staged program with the SCP structure: min sum |u_k|^2 + c'x  s.t.  x_{k+1} = A_k x_k + B_k u_k + r_k,
    x_0 given, box / second-order-cone rows per node, one global variable (a 'time dilation') in every dynamics row."""
    nz = nx + nu
    n = N * nz + 1
    rows, cols, vals, b = [], [], [], []
    r = 0
    for i in range(nx):
        rows.append(r); cols.append(i); vals.append(1.0); b.append(rng.standard_normal()); r += 1
    for k in range(N - 1):
        A = np.eye(nx) + 0.2 * rng.standard_normal((nx, nx)); Bm = rng.standard_normal((nx, nu)); f = 0.1 * rng.standard_normal(nx)
        for i in range(nx):
            rows.append(r); cols.append((k + 1) * nz + i); vals.append(1.0)
            for j in range(nx):
                rows.append(r); cols.append(k * nz + j); vals.append(-A[i, j])
            for j in range(nu):
                rows.append(r); cols.append(k * nz + nx + j); vals.append(-Bm[i, j])
            rows.append(r); cols.append(n - 1); vals.append(-f[i])
            b.append(0.1 * rng.standard_normal()); r += 1
    A_ = sp.csc_matrix((vals, (rows, cols)), shape=(r, n))
    g_rows, g_cols, g_vals, h = [], [], [], []
    m = 0
    for k in range(N):
        for j in range(nz):                                   # |z| <= 5
            for sgn in (1.0, -1.0):
                g_rows.append(m); g_cols.append(k * nz + j); g_vals.append(sgn); h.append(5.0); m += 1
    g_rows += [m, m + 1]; g_cols += [n - 1, n - 1]; g_vals += [1.0, -1.0]; h += [2.0, 0.5]; m += 2     # 'time' in [-0.5, 2]
    l = m
    q = []
    if soc:
        for k in range(N):                                    # |u_k| <= 3
            h.append(3.0); m += 1
            for j in range(nu):
                g_rows.append(m); g_cols.append(k * nz + nx + j); g_vals.append(-1.0); h.append(0.0); m += 1
            q.append(nu + 1)
    G = sp.csc_matrix((g_vals, (g_rows, g_cols)), shape=(m, n))
    Pd = np.zeros(n)
    for k in range(N):
        Pd[k * nz + nx:(k + 1) * nz] = 2.0
    P = sp.diags(Pd).tocsc()
    return 0.1 * rng.standard_normal(n), G, np.array(h), l, q, A_, np.array(b), P


@pytest.mark.parametrize("N,nx,nu,soc", [(40, 3, 2, True), (64, 5, 2, False), (25, 2, 1, True)])
def test_nested_dissection_on_random_chains(N, nx, nu, soc, monkeypatch):
    """random time-staged programs: the dissection finds the chain (depth ~ log2 N), the levels drop several times and
    the three ordering modes return the same optimum"""
    rng = np.random.default_rng(N)
    c, G, h, l, q, A, b, P = _random_chain(rng, N, nx, nu, soc)
    out = {}
    for mode in ("seq", "nd", "auto"):
        monkeypatch.setenv("CONIC_HOST_ORDER", mode)
        out[mode] = conic_host.solve(c, G, h, l, q, A, b, P=P)
    assert out["nd"]["stats"][4] >= int(np.log2(N)) - 1
    assert 2 * out["nd"]["stats"][5] <= out["seq"]["stats"][5]
    for mode in ("nd", "auto"):
        assert out[mode]["status"] == out["seq"]["status"] == 0
        assert abs(out[mode]["pcost"] - out["seq"]["pcost"]) <= 1e-8 * max(1.0, abs(out["seq"]["pcost"]))
        assert np.abs(out[mode]["x"] - out["seq"]["x"]).max() < 1e-6


def test_bad_patterns_are_rejected():
    G = sp.csc_matrix(np.ones((3, 2)))
    with pytest.raises(ValueError):
        conic_host.solve(np.zeros(2), G, np.ones(3), 3, [0])    # cone of dimension 0
    with pytest.raises(ValueError):
        conic_host.solve(np.zeros(2), G, np.ones(3), 3, [], perm=np.zeros(5, np.int32))   # not a permutation
