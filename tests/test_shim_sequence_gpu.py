"""Drives the Julia shim of INTEGRATION.md section 2 call by call through ctypes, with the C structs declared HERE
exactly as the Julia `struct`s of the shim declare them (field order, C types) -- independent of the package's own
`_lib.py` binding -- so that a header change that would break the documented shim breaks this test."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = C.POINTER(C.c_double)


class ModelInfo(C.Structure):   # struct ModelInfo
    _fields_ = [("nx", C.c_int), ("nu", C.c_int), ("np", C.c_int), ("npF", C.c_int), ("Fcols", C.c_int * 8), ("ns", C.c_int),
                ("nic", C.c_int), ("ntc", C.c_int), ("npar", C.c_int), ("npp", C.c_int), ("nl", C.c_int), ("nsoc", C.c_int),
                ("ng", C.c_int), ("structured", C.c_int), ("has_subproblem", C.c_int), ("np_node", C.c_int),
                ("global_rows_in_X", C.c_int), ("linf_groups", C.c_int), ("linf_rows", C.c_int), ("s_input_free", C.c_int)]


class Scaling(C.Structure):     # struct Scaling
    _fields_ = [(n, dp) for n in ("Sx", "cx", "Su", "cu", "Sp", "cp")]


class Desc(C.Structure):        # struct Desc
    _fields_ = [("model_id", C.c_int), ("model_par", dp), ("N", C.c_int), ("Nsub", C.c_int), ("disc_method", C.c_int),
                ("feas_tol", C.c_double), ("scale", Scaling), ("batch_capacity", C.c_int), ("device", C.c_int)]


class PTRParams(C.Structure):   # struct PTRParams
    _fields_ = [("iter_max", C.c_int), ("wvc", C.c_double), ("wtr", C.c_double), ("eps_abs", C.c_double), ("eps_rel", C.c_double),
                ("q_tr", C.c_double), ("q_exit", C.c_double), ("ipm_max_iter", C.c_int), ("ipm_feastol", C.c_double),
                ("ipm_abstol", C.c_double), ("ipm_reltol", C.c_double), ("ipm_reg", C.c_double), ("ipm_nref", C.c_int),
                ("ipm_ref_gap", C.c_double), ("ipm_ref_tol", C.c_double), ("ipm_stall", C.c_int), ("ipm_split_step", C.c_int),
                ("ipm_warm", C.c_int), ("ipm_warm_mu", C.c_double), ("ipm_warm_dev", C.c_double), ("ipm_warm_min_cold", C.c_int),
                ("ipm_wpe", C.c_int), ("ipm_warm_mu_coarse", C.c_double), ("ipm_warm_mu_mid", C.c_double), ("ipm_warm_dev_mid", C.c_double),
                ("ipm_warm_mu_vfine", C.c_double), ("ipm_warm_dev_vfine", C.c_double)]


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def test_shim_call_sequence(pkg, orc):
    from oracle import ptr_ref
    from oracle.models import MODELS
    L = C.CDLL(os.path.join(ROOT, "scptoolbox.jl_amd", "csrc", "libscp_mi355x.so"))
    model_id, N, Nsub, B = 1, 12, 8, 2                     # quadrotor
    mdl = MODELS["quadrotor"]()
    sc = ptr_ref.Scaling(*mdl.bbox())
    # model_info(model_id)
    info = ModelInfo()
    assert L.scp_model_query(C.c_int(model_id), C.byref(info)) == 0
    assert (info.nx, info.nu, info.np, info.npF) == (6, 4, 1, 1) and info.Fcols[0] == 0
    # create(model_id, model_par, pars, scale; batch)
    par = np.ascontiguousarray(mdl.par(), dtype=np.float64)
    arrs = [np.ascontiguousarray(v, dtype=np.float64) for v in (sc.Sx, sc.cx, sc.Su, sc.cu, sc.Sp, sc.cp)]
    d = Desc(model_id, par.ctypes.data_as(dp), N, Nsub, 0, 1e-3, Scaling(*[a.ctypes.data_as(dp) for a in arrs]), B, 0)
    h = C.c_void_p()
    assert L.scp_problem_create(C.byref(d), C.byref(h)) == 0
    del par, arrs   # "the library copies model_par and the scaling: nothing has to stay rooted"
    # discretize!(ref, pbm, nat): B = 1, Julia column-major arrays == numpy arrays with reversed dimensions
    pp = mdl.nominal_pp()
    x, u, p = mdl.guess(N, pp)
    x = x + 0.01 * np.random.default_rng(0).standard_normal(x.shape)
    nx, nu, npar_ = info.nx, info.nu, info.np
    A = np.zeros((N - 1, nx, nx)); Bm = np.zeros((N - 1, nu, nx)); Bp = np.zeros((N - 1, nu, nx)); Fc = np.zeros((N - 1, max(info.npF, 1), nx))
    r = np.zeros((N - 1, nx)); E = np.zeros((N - 1, nx, nx)); defect = np.zeros((N - 1, nx))
    feas = C.c_uint8(0); secs = C.c_double(0)
    L.scp_discretize_batch_host.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 11 + [C.POINTER(C.c_double)]
    xs, us, ps = np.ascontiguousarray(x), np.ascontiguousarray(u), np.ascontiguousarray(p)
    assert L.scp_discretize_batch_host(h, 1, P(xs), P(us), P(ps), P(A), P(Bm), P(Bp), P(Fc), P(r), P(E), P(defect), C.byref(feas), C.byref(secs)) == 0
    F = np.zeros((N - 1, npar_, nx))                       # scatter_F!
    for jj in range(info.npF):
        F[:, info.Fcols[jj], :] = Fc[:, jj, :]
    o = orc.discretize("quadrotor", orc.default_params("quadrotor"), N, Nsub, x[None], u[None], p[None], 1.0 / sc.Sx, 1e-3)
    for nm, got in (("A", A), ("Bm", Bm), ("Bp", Bp), ("F", F), ("r", r), ("E", E), ("defect", defect)):
        assert np.abs(got - o[nm][0]).max() <= 1e-10 * max(1.0, np.abs(o[nm][0]).max()), nm
    assert bool(feas.value) == bool(o["feas"][0]) and secs.value > 0
    # solve_batch(nat, pars, xd, ud, p, pp)
    pars = PTRParams(8, 1e3, 0.1, 0.0, 0.0, float("inf"), float("inf"), 100, 1e-8, 1e-8, 1e-8, 1e-12, 1, 1e-2, 0.0, 3, 0, 1, 1e-7, 1e-3, 25, 0, 1e-1)
    g = [mdl.guess(N, pp) for _ in range(B)]
    xd = np.ascontiguousarray(np.stack([a[0] for a in g])); ud = np.ascontiguousarray(np.stack([a[1] for a in g]))
    pv = np.ascontiguousarray(np.stack([a[2] for a in g]).reshape(B, -1)); ppb = np.ascontiguousarray(np.repeat(pp[None], B, 0))
    xo, uo, po = np.zeros_like(xd), np.zeros_like(ud), np.zeros_like(pv)
    status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); cost = np.zeros((B, 4)); fz = np.zeros(B, np.uint8); secs = C.c_double(0)
    L.scp_ptr_solve_batch_host.argtypes = [C.c_void_p, C.c_int, C.POINTER(PTRParams)] + [C.c_void_p] * 11 + [C.POINTER(C.c_double)]
    assert L.scp_ptr_solve_batch_host(h, B, C.byref(pars), P(xd), P(ud), P(pv), P(ppb), P(xo), P(uo), P(po), P(status), P(iters), P(cost),
                                      P(fz), C.byref(secs)) == 0
    assert (status == 0).all() and (iters == 8).all() and fz.all()
    opars = ptr_ref.PTRParameters(N, Nsub, 8, 1e3, 0.1, 0, 0, 1e-3)
    st, hist = ptr_ref.ptr_solve("quadrotor", opars)
    assert st == "SCP_SOLVED"
    assert abs(cost[0, 3] - hist[-1]["sub"]["J_aug"]) <= 1e-4 * abs(hist[-1]["sub"]["J_aug"])   # SCPSolution.cost = J_aug
    assert np.array_equal(xo[0], xo[1])
    # virtual_controls(nat, N, B)
    vd = np.zeros((B, N - 1, nx)); vs = np.zeros((B, N, max(info.ns, 1))); vic = np.zeros((B, max(info.nic, 1))); vtc = np.zeros((B, max(info.ntc, 1)))
    Pk = np.zeros((B, N)); Pf = np.zeros((B, 2))
    L.scp_ptr_get_virtual_controls_host.argtypes = [C.c_void_p] * 7
    assert L.scp_ptr_get_virtual_controls_host(h, P(vd), P(vs), P(vic), P(vtc), P(Pk), P(Pf)) == 0
    w = np.full(N, 1.0 / (N - 1)); w[0] = w[-1] = 0.5 / (N - 1)
    assert abs(1e3 * (w @ Pk[0] + Pf[0].sum()) - cost[0, 2]) <= 1e-9 * max(1.0, cost[0, 2])
    assert L.scp_problem_destroy(h) == 0
