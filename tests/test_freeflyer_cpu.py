"""Oracle pins for the 6-DoF free-flyer (test/examples/freeflyer): the reference ships no golden data, so the C
restatement of its dynamics / Jacobians / integration action (oracle/scp_oracle.c) and the restatements of its initial
guess are pinned on mathematics (finite differences, quaternion identities, closed forms) and against each other."""
import os

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


def test_full_problem_definition_is_consistent(orc):
    """Freeflyer(N): constraint Jacobians vs finite differences, the room SDF, the delta bookkeeping of the guess."""
    N = 9
    mdl = MODELS["freeflyer"](N)
    assert mdl.np == 1 + 6 * N and mdl.np_dyn == 1
    x, u, p = mdl.guess(N, mdl.nominal_pp())
    rng = np.random.default_rng(4)
    k = 4
    xk = x[k - 1] + 0.05 * rng.standard_normal(13)
    pk = p + 0.05 * rng.standard_normal(p.size)
    s0, C, G = mdl.s(0.0, k, xk, u[0], pk), mdl.C(0.0, k, xk, u[0], pk), mdl.G(0.0, k, xk, u[0], pk)
    eps = 1e-6
    for j in range(13):
        d = np.zeros(13); d[j] = eps
        fd = (mdl.s(0.0, k, xk + d, u[0], pk) - mdl.s(0.0, k, xk - d, u[0], pk)) / (2 * eps)
        assert np.abs(fd - C[:, j]).max() < 1e-7
    for j in mdl.id_delta(k):
        d = np.zeros(p.size); d[j] = eps
        fd = (mdl.s(0.0, k, xk, u[0], pk + d) - mdl.s(0.0, k, xk, u[0], pk - d)) / (2 * eps)
        assert np.abs(fd - G[:, j]).max() < 1e-7
    assert not G[:, np.setdiff1d(np.arange(p.size), mdl.id_delta(k))].any()       # only the node's own slacks
    # delta of the guess = the signed distance 1 - |(r - c) / s|_inf of every room; the start position is inside room 1 or 2
    d0 = p[mdl.id_delta(1)]
    assert d0.max() > 0 and np.allclose(d0, 1 - np.abs((x[0, 0:3] - mdl.room_c) / mdl.room_s).max(axis=1))
    # X rows: the LINF room cone holds with equality in its tightest coordinate at the guess
    for kind, M, Mp, m0 in mdl.X(0.0, 1):
        z = M @ x[0] + Mp @ p + m0
        if kind == "LINF":
            assert abs(z[0] - np.abs(z[1:]).max()) < 1e-12


def test_scvx_loop_on_the_full_problem_and_the_golden_run(orc):
    """The oracle's literal SCvx loop on the reference's free-flyer problem: a short run on a coarse grid (regression) and
    the committed run at the reference's own test parameters (freeflyer/tests.jl:25-80), whose only pinned outcome in the
    reference is `status == SCP_SOLVED`."""
    import os
    from oracle import scvx_ref
    N = 20
    mdl = MODELS["freeflyer"](N)
    pars = scvx_ref.SCvxParameters(N, 15, 4, lam=1e3, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                   eta_lb=1e-6, eta_ub=10.0, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
    st, hist = scvx_ref.scvx_solve(mdl, pars)
    assert st == "SCP_SOLVED" and all(h["sub"]["status"] in ("OPTIMAL", "ALMOST_OPTIMAL") for h in hist)
    assert hist[-1]["sol"].feas and hist[-1]["sub"]["L"] < hist[0]["sub"]["L"]
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "freeflyer_scvx_N50.npz"))
    assert str(g["status"]) == "SCP_SOLVED" and int(g["N"]) == 50 and bool(g["feas"][-1])
    big = MODELS["freeflyer"](50)
    xd, ud, p = g["xd"], g["ud"], g["p"]
    assert np.abs(np.linalg.norm(xd[:, 6:10], axis=1) - 1.0).max() < 1e-3          # unit attitude up to feas_tol
    assert max(big.s(0.0, k + 1, xd[k], ud[k], p).max() for k in range(50)) < 1e-6    # obstacles cleared, inside the station
    assert np.linalg.norm(xd[:, 3:6], axis=1).max() <= big.v_max + 1e-6 and np.linalg.norm(ud[:, 0:3], axis=1).max() <= big.T_max * (1 + 1e-6)
    assert big.tf_min - 1e-6 <= p[0] <= big.tf_max + 1e-6
    assert np.abs(big.gic(xd[0], p, g["pp"])).max() < 1e-6 and np.abs(big.gtc(xd[-1], p, g["pp"])).max() < 1e-6
    L = g["L"]
    assert L[-1] < 0.5 * L[0] and abs(L[-1] - L[-2]) < 1e-6                           # converged


def test_gusto_loop_with_cone_indicators_and_the_golden_run(orc):
    """GuSTO on the free-flyer (freeflyer/tests.jl:84-140): the convex state constraints enter as cone indicators with a soft
    quadratic penalty (SOC speed limits, LINF rooms; define_conic_constraint!, gusto.jl:883-995).  Two iterations of the
    oracle loop at the reference's grid (first step accepted inside the initial trust region) + the committed full run."""
    import os
    from oracle import gusto_ref
    N = 50
    mdl = MODELS["freeflyer"](N)
    gp = gusto_ref.GuSTOParameters(N, 15, 2, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.5, beta_sh=2.0, beta_gr=2.0,
                                   gamma_fail=5.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=16, eps_abs=0.0,
                                   eps_rel=0.0, feas_tol=1e-3)
    st, hist = gusto_ref.gusto_solve(mdl, gp)
    assert st == "SCP_SOLVED" and hist[0]["accept"] and hist[0]["deviation"] < 1.0 and not hist[0]["trust_viol"]
    assert hist[1]["sub"]["L"] < hist[0]["sub"]["L"] and hist[1]["sol"].feas
    # indicators at the iterate: inside every cone up to the soft-penalty slack
    x1, p1 = hist[1]["sol"].xd, hist[1]["sol"].p
    assert max(max(gusto_ref._indicators(mdl, 0.0, k + 1, x1[k], p1)) for k in range(N)) < 1e-2
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "freeflyer_gusto_N50.npz"))
    s = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "freeflyer_scvx_N50.npz"))
    assert str(g["status"]) == "SCP_SOLVED" and bool(g["feas"][-1]) and bool(g["accept"][:8].all())
    assert abs(g["L"][0] - hist[0]["sub"]["L"]) < 1e-6 and abs(g["L"][1] - hist[1]["sub"]["L"]) < 1e-6     # reproducible
    assert np.all(g["lam"][2:] == 1e4) and g["eta"][-1] == 10.0
    # the soft state constraints let GuSTO cut the corner slightly: a few % below the hard-constrained SCvx optimum
    assert 0.9 * s["L"][-1] < g["L"][-1] < s["L"][-1]
    assert max(mdl.s(0.0, k + 1, g["xd"][k], g["ud"][k], g["p"]).max() for k in range(N)) < 1e-3          # within c_buffer


def _eval_host(pkg, mdl_py, N, k, x, u, p):
    """scp_model_eval_host of the compiled free-flyer"""
    import ctypes
    L = pkg._lib.lib()
    par = np.ascontiguousarray(mdl_py.par(), float)
    f = np.zeros(13); A = np.zeros((13, 13)); B = np.zeros((6, 13)); F = np.zeros((1, 13))
    s = np.zeros(4); C = np.zeros((4, 13)); D = np.zeros((4, 6)); G = np.zeros((4, 7)); q = np.zeros(32); nq = ctypes.c_int(0)
    vp = lambda a: np.ascontiguousarray(a, float).ctypes.data_as(ctypes.c_void_p)
    xx, uu, pp_ = np.ascontiguousarray(x, float), np.ascontiguousarray(u, float), np.ascontiguousarray(p, float)
    rc = L.scp_model_eval_host(4, vp(par), N, k, vp(xx), vp(uu), vp(pp_), vp(f), vp(A), vp(B), vp(F), vp(s), vp(C), vp(D), vp(G),
                               vp(q), ctypes.byref(nq))
    assert rc == 0
    return dict(f=f, A=A.T, B=B.T, F=F.T, s=s, C=C, D=D, G=G, q=q[:nq.value])


def test_compiled_model_equals_the_oracle_model(pkg, orc):
    """The device model csrc/models/freeflyer.hpp evaluated on the host (scp_model_eval_host / scp_model_rows) against the
    oracle's restatement of freeflyer/definition.jl: dynamics, s / C / G with the compact parameter columns (the node's own six
    slacks), the cone indicators of X (2 SOC, t_f bounds, 6 LINF rooms), the lowered X / U rows, the cost, scaling and guess
    -- with every constant coming through the parameter blob."""
    from oracle import gusto_ref
    N = 12
    mdl = MODELS["freeflyer"](N)
    pm = pkg.REGISTRY["freeflyer"](N=N)
    assert pm.par().size == 61 and pm.np == 1 + 6 * N
    np.testing.assert_allclose(pm.room_c, mdl.room_c, atol=1e-14); np.testing.assert_allclose(pm.room_s, mdl.room_s, atol=1e-14)
    for a, b in zip(pm.scale_advice(), mdl.bbox()):
        np.testing.assert_allclose(np.asarray(a), np.asarray(b), atol=1e-14)
    pp = mdl.nominal_pp()
    x, u, p = mdl.guess(N, pp)
    xg, ug, pg = pm.guess(N, pp)
    np.testing.assert_allclose(xg, x, atol=1e-13); np.testing.assert_allclose(pg, p, atol=1e-13)
    rng = np.random.default_rng(1)
    t = np.linspace(0.0, 1.0, N)
    for k in (1, 5, N):
        xk = x[k - 1] + np.concatenate([0.4 * rng.standard_normal(3), 0.3 * rng.standard_normal(3), 0.1 * rng.standard_normal(4),
                                        0.02 * rng.standard_normal(3)])
        uk = 1e-2 * rng.standard_normal(6)
        pk = p + 0.05 * rng.standard_normal(p.size)
        e = _eval_host(pkg, pm, N, k, xk, uk, pk)
        f, A, B, F = orc.model_eval("freeflyer", mdl.par(), t[k - 1], k, xk, uk, pk[:1])
        for nm, want in (("f", f), ("A", A), ("B", B), ("F", F)):
            assert np.abs(e[nm] - want).max() <= 1e-13 * max(1.0, np.abs(want).max()), nm
        assert np.abs(e["s"] - mdl.s(t[k - 1], k, xk, uk, pk)).max() < 1e-13
        assert np.abs(e["C"] - mdl.C(t[k - 1], k, xk, uk, pk)).max() < 1e-13 and not e["D"].any()
        Gfull = mdl.G(t[k - 1], k, xk, uk, pk)
        cols = np.concatenate([[0], mdl.id_delta(k)])
        assert np.abs(e["G"] - Gfull[:, cols]).max() < 1e-13 and not np.delete(Gfull, cols, axis=1).any()
        want_q = gusto_ref._indicators(mdl, t[k - 1], k, xk, pk)
        assert e["q"].size == 10 and np.abs(np.sort(e["q"]) - np.sort(want_q)).max() < 1e-13
    # X / U rows and cost through ModelRows vs the oracle rows (template_util.OracleRows lowers them the same way)
    from template_util import OracleRows
    mr = pkg.subproblem.ModelRows(pm, N); orr = OracleRows(mdl, N)
    assert (mr.np, mr.npF, mr.nl, mr.nsoc, mr.ng, mr.ns, mr.state_indicators(N)) == (1 + 6 * N, 1, 36, 4, 2, 4, 10)
    for k in (1, 7, N):
        for a, b in zip(mr.rows(N, k), orr.rows(N, k)):
            np.testing.assert_allclose(a, b, atol=1e-14)
        assert mr.linf_groups(N, k) == orr.linf_groups(N, k)
    for a, b in zip(mr.global_rows(N), orr.global_rows(N)):
        np.testing.assert_allclose(a, b, atol=1e-14)
    ca, cb = mr.cost_terms(N), orr.cost_terms(N)
    for key in ca:
        np.testing.assert_allclose(ca[key], cb[key], atol=1e-14)
    # an override reaches the compiled model
    pm2 = pkg.REGISTRY["freeflyer"](N=N, hom=5.0, eps_sdf=3e-4, v_max=0.2)
    e1, e2 = _eval_host(pkg, pm, N, 3, x[2], u[2], p), _eval_host(pkg, pm2, N, 3, x[2], u[2], p)
    assert abs(e1["s"][3] - e2["s"][3]) > 1e-5 and np.array_equal(e1["s"][:3], e2["s"][:3])
    assert pkg.subproblem.ModelRows(pm2, N).cost_terms(N)["tp"][1] == -3e-4
    assert pkg.subproblem.ModelRows(pm2, N).rows(N, 1)[4][0] == 0.2


def test_templates_of_the_compiled_model_equal_the_oracle_programs(pkg, orc):
    """SCvx and GuSTO templates formulated from the COMPILED model's rows (scp_model_rows, compact parameter columns) solve to
    the optimum of the oracle's literal programs -- the host half of what the device runs (tests/test_freeflyer_gpu.py)."""
    from oracle import conic_host, gusto_ref, ptr_ref, scvx_ref
    from template_util import make_src, template_matrices
    N, Nsub = 10, 8
    mdl = MODELS["freeflyer"](N)
    scale = ptr_ref.Scaling(*mdl.bbox())
    mr = pkg.subproblem.ModelRows(pkg.REGISTRY["freeflyer"](N=N), N)
    pp = mdl.nominal_pp()
    x, u, p = mdl.guess(N, pp)
    rng = np.random.default_rng(0)
    x = x + 0.02 * scale.Sx * rng.standard_normal(x.shape); x[:, 6:10] /= np.linalg.norm(x[:, 6:10], axis=1, keepdims=True)
    sp = scvx_ref.SCvxParameters(N, Nsub, 3, lam=1e3, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                 eta_lb=1e-6, eta_ub=10.0, eps_abs=0.0, eps_rel=0.0, feas_tol=1e-3)
    ref = ptr_ref.discretize(mdl, sp, scale, x, u, p)
    T = pkg.subproblem.build_scvx(mr, N, scale, sp.lam)
    assert T.sources.segs["Gs"][1] == (4, 7, N) and T.sources.segs["K0"][1] == (13, 1) and T.sources.segs["pref"][1] == (1 + 6 * N,)
    o = ptr_ref.solve_subproblem(mdl, sp, scale, ref, pp, algo="scvx", eta=0.5)
    v, G, A, P = template_matrices(T, make_src(T, mdl, ref, pp, 0.5, Fcols=[0]))
    r = conic_host.solve(v["c"], G, v["h"], T.l, T.q, A, v["b"], P=P)
    assert r["status"] in (0, 1) and abs(r["pcost"] + T.cost_const - o["L_aug"]) <= 2e-7 * max(1.0, abs(o["L_aug"]))
    gp = gusto_ref.GuSTOParameters(N, Nsub, 3, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.5, beta_sh=2.0, beta_gr=2.0,
                                   gamma_fail=5.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=16, eps_abs=0.0,
                                   eps_rel=0.0, feas_tol=1e-3)
    T = pkg.subproblem.build_gusto(mr, N, scale, literal_slack=True)
    assert T.nst == mr.state_indicators(N) + mdl.ns == 14
    o = gusto_ref.solve_subproblem(mdl, gp, scale, ref, pp, 5e4, 0.2)
    assert T.n == o["sizes"]["n"] and T.p == o["sizes"]["p"]
    v, G, A, P = template_matrices(T, make_src(T, mdl, ref, pp, [0.2, 5e4], Fcols=[0]))
    r = conic_host.solve(v["c"], G, v["h"], T.l, T.q, A, v["b"], P=P)
    assert r["status"] in (0, 1) and abs(r["pcost"] + T.cost_const - o["L_aug"]) <= 2e-6 * max(1.0, abs(o["L_aug"]))


def test_product_solver_on_the_oracle_loops_own_subproblems_at_config_size(pkg, orc, monkeypatch):
    """BASELINE.json configs[4] at its stated size (free-flyer 6-DoF, GuSTO, N = 200, np = 1 201): the first and the third
    subproblem of the ORACLE's literal loop (tests/golden/freeflyer_gusto_N200.npz, reference test parameters) formulated by the
    product's template (n = 10 402, p = 2 613, m = 21 602) and solved by the product's solver in the nested order: OPTIMAL, the
    oracle's optimum (measured on all four: 4e-10 ... 3.3e-7 relative)."""
    from oracle import conic_host, ptr_ref
    from oracle.models import MODELS
    from template_util import make_src, template_matrices
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "freeflyer_gusto_N200.npz"))
    N, Nsub = int(g["N"]), int(g["Nsub"])
    mdl = MODELS["freeflyer"](N)
    pm = pkg.REGISTRY["freeflyer"](N=N)
    mr = pkg.subproblem.ModelRows(pm, N)
    scale = ptr_ref.Scaling(*mdl.bbox())
    pars = ptr_ref.PTRParameters(N, Nsub, 3, 1e3, 0.1, 0, 0, 1e-3)
    T = pkg.subproblem.build_gusto(mr, N, scale)
    assert (T.n, T.p, T.m) == (10402, 2613, 21602)
    monkeypatch.setenv("CONIC_HOST_ORDER", "nd")
    for k in (0, 2):
        ref = ptr_ref.discretize(mdl, pars, scale, g["ref_xd"][k], g["ref_ud"][k], g["ref_p"][k])
        v, G, A, P = template_matrices(T, make_src(T, mdl, ref, g["pp"], [float(g["eta"][k]), float(g["lam"][k])], Fcols=[0]))
        r = conic_host._solve(v["c"], G, v["h"], T.l, T.q, A, v["b"], P=P)
        assert r["status"] == 0, (k, r["status"])
        assert abs(r["pcost"] + T.cost_const - g["L_aug"][k]) <= 5e-7 * max(1.0, abs(g["L_aug"][k])), (k, r["pcost"] + T.cost_const, g["L_aug"][k])
