"""The model is DATA (src/parser/problem.jl:64-121: `traj.mdl` is arbitrary user data the closures read): every vehicle /
environment constant of the compiled models crosses the C ABI in the parameter blob, and the host-side scaling, guess and
formulation read the same values -- an override changes what the LIBRARY computes (here through its host entry points
scp_model_eval_host / scp_model_rows; on the device in tests/test_freeflyer_gpu.py::test_model_constants_cross_the_abi)."""
import ctypes

import numpy as np


def _eval(pkg, mdl, N, k, x, u, p):
    L = pkg._lib.lib()
    info = pkg._lib.ScpModelInfo()
    mid = pkg.models.MODEL_IDS[mdl.name]
    assert L.scp_model_query(mid, ctypes.byref(info)) == 0
    par = np.ascontiguousarray(mdl.par(), float)
    assert par.size == info.npar
    npc = info.np + info.np_node
    s = np.zeros(max(info.ns, 1)); C = np.zeros((max(info.ns, 1), info.nx)); G = np.zeros((max(info.ns, 1), max(npc, 1)))
    f = np.zeros(info.nx)
    vp = lambda a: np.ascontiguousarray(a, float).ctypes.data_as(ctypes.c_void_p)
    xx, uu, pp = (np.ascontiguousarray(a, float) for a in (x, u, p))
    assert L.scp_model_eval_host(mid, vp(par), N, k, vp(xx), vp(uu), vp(pp), vp(f), None, None, None, vp(s), vp(C), None, vp(G), None, None) == 0
    return f, s, C


def test_quadrotor_constants_are_data(pkg):
    N = 10
    base = pkg.REGISTRY["quadrotor"]()
    mod = pkg.REGISTRY["quadrotor"](g=3.71, u_max=12.0, tilt_max=np.deg2rad(30), tf_max=4.0,
                                    obstacles=[([2.0, 2.0, 0.0], [0.5, 0.5, 0.0]), ([1.5, 1.5, 0.0], [2.0, 5.0, 0.0])])
    assert base.par().size == 19 and mod.par()[0] == 3.71
    x = np.array([0.5, 0.5, 0.0, 0.1, 0.2, 0.0]); u = np.array([0.0, 0.0, 5.0, 5.0]); p = np.array([1.5])
    f0, s0, C0 = _eval(pkg, base, N, 3, x, u, p)
    f1, s1, C1 = _eval(pkg, mod, N, 3, x, u, p)
    assert f0[5] == (5.0 - 9.81) * 1.5 and f1[5] == (5.0 - 3.71) * 1.5           # gravity
    assert s1[0] == 1.0 and s0[0] < 0.0 and s0[1] == s1[1]                          # first obstacle moved onto the point
    assert not np.isnan(C1).any() and not C1[0].any()                                # guarded gradient at the obstacle's centre
    mr0, mr1 = pkg.subproblem.ModelRows(base), pkg.subproblem.ModelRows(mod)
    L0, _, l0, _, _ = mr0.rows(N, 1); L1, _, l1, _, _ = mr1.rows(N, 1)
    assert l0[1] == -23.2 and l1[1] == -12.0 and L1[2, 6 + 3] == np.cos(np.deg2rad(30)) and L0[2, 6 + 3] == np.cos(np.deg2rad(60))
    assert mr1.global_rows(N)[1][0] == -4.0
    # scaling and guess on the host follow the same values
    assert mod.scale_advice()[1][3, 1] == 12.0 and mod.scale_advice()[2][0, 1] == 4.0
    assert mod.guess(N, mod.nominal_pp())[2][0] == 2.0 and mod.guess(N, mod.nominal_pp())[1][0, 2] == 3.71


def test_rocket_constants_are_data(pkg):
    N = 10
    base = pkg.REGISTRY["rocket_landing"]()
    mod = pkg.REGISTRY["rocket_landing"](m_dry=1400.0, rho_max=20000.0, v_max=100.0, tf_max=150.0, g=[0.0, 0.0, -9.81])
    assert base.par().size == 17
    x = np.array([100.0, 0.0, 500.0, 1.0, 2.0, -10.0, np.log(1800.0)]); u = np.array([0.0, 0.0, 4.0, 4.5]); p = np.array([60.0])
    f0, s0, _ = _eval(pkg, base, N, 2, x, u, p)
    f1, s1, _ = _eval(pkg, mod, N, 2, x, u, p)
    assert abs((f1[5] - f0[5]) / 60.0 - (-9.81 + 3.7114)) < 1e-12                   # gravity enters v_z'
    assert abs(s1[1] - (4.5 - 20000.0 / 1800.0)) < 1e-12 and abs(s0[1] - s1[1]) > 1e-3   # upper thrust bound xi - rho_max e^-z
    mr1 = pkg.subproblem.ModelRows(mod)
    L, Lp, l, Mm, m = mr1.rows(N, 1)
    assert np.log(1400.0) in l and 100.0 in m                                        # z >= ln m_dry, ||v|| <= v_max
    assert mr1.global_rows(N)[1][0] == -150.0
    assert mod.scale_advice()[0][6, 0] == np.log(1400.0) and mod.guess(N, mod.nominal_pp())[0][-1, 6] == np.log(1400.0)


def test_starship_constants_are_data(pkg, orc):
    """starship_flip/parameters.jl:99-212 through the blob: vehicle, engine, aerodynamic and trajectory constants; at the
    defaults the compiled model's dynamics and Jacobians equal the C oracle's (which has its own constants); overrides are
    followed."""
    N = 31
    base = pkg.REGISTRY["starship"](); base.N = N
    mod = pkg.REGISTRY["starship"](m=100e3, T_max1=2500e3, gamma_gs=np.deg2rad(20.0), tf_max=50.0, rate_delay=0.1, vf_y=-0.5, cost_alt=0.6)
    mod.N = N
    assert base.par().size == 25 and mod.par()[3] == 100e3 and mod.T_max3 == 7500e3            # derived quantities follow
    assert abs(mod.J - 1.0 / 12.0 * 100e3 * (6 * 4.5 ** 2 + 50.0 ** 2)) < 1e-6
    x = np.array([50.0, 300.0, 5.0, -60.0, 1.2, 0.05, -500.0, 0.02]); u = np.array([1.5e6, 0.05, 0.01])
    p = np.concatenate([[9.0, 11.0], x])
    f0, s0, C0 = _eval(pkg, base, N, 5, x, u, p)
    f1, s1, C1 = _eval(pkg, mod, N, 5, x, u, p)
    # the compiled model (blob) against the C oracle (own constants) at the defaults: f, A, B, F
    L_ = pkg._lib.lib()
    info = pkg._lib.ScpModelInfo(); L_.scp_model_query(pkg.models.MODEL_IDS["starship"], ctypes.byref(info))
    vp = lambda a: np.ascontiguousarray(a, float).ctypes.data_as(ctypes.c_void_p)
    for k in (5, 20):                                                         # flip and landing phase
        fa = np.zeros(8); Af = np.zeros(64); Bf = np.zeros(24); Ff = np.zeros(8 * info.npF)        # column-major outputs
        par = np.ascontiguousarray(base.par(), float); xx, uu, pq = (np.ascontiguousarray(a, float) for a in (x, u, p))
        assert L_.scp_model_eval_host(pkg.models.MODEL_IDS["starship"], vp(par), N, k, vp(xx), vp(uu), vp(pq), vp(fa), vp(Af), vp(Bf),
                                      vp(Ff), None, None, None, None, None, None) == 0
        Aa, Ba, Fa = Af.reshape(8, 8, order="F"), Bf.reshape(8, 3, order="F"), Ff.reshape(8, info.npF, order="F")
        fo, Ao, Bo, Fo = orc.model_eval("starship", orc.default_params("starship"), (k - 1) / (N - 1), k, x, u, p)
        assert np.abs(fa - fo).max() <= 1e-13 * np.abs(fo).max() and np.abs(Aa - Ao).max() <= 1e-13 * np.abs(Ao).max()
        assert np.abs(Ba - Bo).max() <= 1e-13 * np.abs(Bo).max() and np.abs(Fa - Fo[:, :info.npF]).max() <= 1e-13 * np.abs(Fo).max()
    td = 9.0 / 0.5                                                            # node 5 of 31: flip phase, t1 / tau_s
    assert abs(f0[6] - (-1.0 / (330.0 * 9.81)) * 1.5e6 * td) < 1e-9 and f1[6] == f0[6]      # mass flow alpha_e T
    assert abs(f0[7] - (0.05 - 0.02) / 0.05 * td) < 1e-12
    assert abs(f1[7] / f0[7] - 0.5) < 1e-12                                   # gimbal delay 0.05 -> 0.1
    # thrust acceleration scales with 1 / m (the drag term does not: CD is defined through m, parameters.jl:134)
    th, de = 1.2, 0.05
    Tx = 1.5e6 * (-np.sin(de) * np.cos(th) + np.cos(de) * (-np.sin(th)))
    assert abs((f1[2] - f0[2]) / td - Tx * (1 / 100e3 - 1 / 120e3)) < 1e-9
    assert abs(s0[4] - (np.hypot(50.0, 300.0) * np.cos(np.deg2rad(27.0)) - 300.0)) < 1e-9
    assert abs(s1[4] - (np.hypot(50.0, 300.0) * np.cos(np.deg2rad(20.0)) - 300.0)) < 1e-9
    mr1 = pkg.subproblem.ModelRows(mod)
    L, Lp, l, Mm, m = mr1.rows(N, 20)                                          # node 20: landing phase, one engine
    assert -2500e3 in l and mr1.global_rows(N)[1][0] == -50.0
    assert mod.guess(N, mod.nominal_pp())[0][-1, 3] == -0.5 and mod.scale_advice()[0][6, 1] == 100e3
    ct = mr1.cost_terms(N)
    assert abs(ct["tp"][3] + 0.6 / 100.0) < 1e-15 and abs(ct["tx"][6] + 1.0 / 10e3) < 1e-18      # -cost_alt / hs on xs[alt], -1 / cost_mass on m_N


def test_unknown_model_constants_are_refused(pkg):
    import pytest
    with pytest.raises(ValueError, match="unknown model constant"):
        pkg.REGISTRY["starship"](T_max=1.0)              # the constant is called T_max1
    with pytest.raises(ValueError, match="unknown model constant"):
        pkg.TrajectoryProblem("quadrotor", gravity=3.7)
    pkg.TrajectoryProblem("rocket_landing", m_dry=1400.0)
