"""CPU-side checks of the C-ABI library: it loads, exports every symbol the
header declares, answers model queries, and reports errors by status code
(never by exception/abort) -- no compute calls without a GPU."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    syms = set()
    for hdr in ("scp_mi355x.h", "scp_conic.h"):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        syms |= set(re.findall(r"\b((?:scp|socp)_[a-z_0-9]+)\s*\(", src))
    return sorted(syms)


def test_library_exports_every_declared_symbol(pkg):
    L = pkg._lib.lib()
    syms = _header_symbols()
    assert "scp_discretize_batch_host" in syms and "scp_problem_create" in syms
    for s in syms:
        assert hasattr(L, s), "missing export " + s
    assert sorted(pkg._lib.EXPORTS) == syms, "python binding list out of date vs the header"


def test_model_query(pkg):
    L = pkg._lib.lib()
    info = pkg._lib.ScpModelInfo()
    assert L.scp_model_query(1, ctypes.byref(info)) == 0
    assert (info.nx, info.nu, info.np, info.npF) == (6, 4, 1, 1)  # quadrotor/definition.jl:44
    assert L.scp_model_query(2, ctypes.byref(info)) == 0
    assert (info.nx, info.nu, info.np) == (7, 4, 1)
    assert L.scp_model_query(99, ctypes.byref(info)) == 2  # SCP_ERR_UNKNOWN_MODEL


def test_create_reports_errors_by_status(pkg):
    import torch
    L = pkg._lib.lib()
    d = pkg._lib.ScpProblemDesc()
    h = ctypes.c_void_p()
    assert L.scp_problem_create(None, ctypes.byref(h)) == 1
    d.model_id = 42
    assert L.scp_problem_create(ctypes.byref(d), ctypes.byref(h)) == 2
    d.model_id = 1
    d.N, d.Nsub, d.batch_capacity = 1, 5, 1
    assert L.scp_problem_create(ctypes.byref(d), ctypes.byref(h)) == 1  # N < 2
    if not torch.cuda.is_available():
        # valid description but no GPU: loud status, no fallback
        traj = pkg.TrajectoryProblem("quadrotor")
        pars = pkg.PTR.Parameters(N=5, Nsub=5, iter_max=1)
        try:
            pkg.PTR.create(pars, traj)
            raise AssertionError("create must fail without a GPU")
        except pkg._lib.ScpError as e:
            assert e.code == 3  # SCP_ERR_NO_DEVICE


def test_scaling_matches_reference_rule(pkg):
    """scp.jl:479-511: S = (max-min)/1, widths below sqrt(eps) -> 1, c = min."""
    s = pkg.SCPScaling([[0, 1], [2, 2], [-3, 5]], [[0.6, 23.2]], [[0.0, 2.5]])
    np.testing.assert_allclose(s.Sx, [1.0, 1.0, 8.0])
    np.testing.assert_allclose(s.cx, [0.0, 2.0, -3.0])
    np.testing.assert_allclose(s.Su, [22.6]); np.testing.assert_allclose(s.cu, [0.6])
    np.testing.assert_allclose(s.Sp, [2.5]); np.testing.assert_allclose(s.iSp, [0.4])


def test_straightline_and_linrange(pkg):
    from scptoolbox_jl_amd.models import linrange, straightline_interpolate
    t = linrange(0.0, 1.0, 5)
    np.testing.assert_allclose(t, [0, .25, .5, .75, 1.0])
    x = straightline_interpolate([0.0, 2.0], [4.0, 2.0], 5)
    np.testing.assert_allclose(x[:, 0], [0, 1, 2, 3, 4]); np.testing.assert_allclose(x[:, 1], 2.0)


def test_headers_are_plain_c99(tmp_path):
    """the drop-in boundary is a C ABI: both headers must compile as C99 without torch / C++ types (gcc -pedantic)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "scp_mi355x.h"\n#include "scp_conic.h"\n'
                   "int main(void) { scp_problem_desc d; scp_gusto_params g; scp_scvx_params s; scp_sub_template t; scp_model_info i;\n"
                   "  (void)d; (void)g; (void)s; (void)t; (void)i; return 0; }\n")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), "-c", str(src),
                        "-o", str(tmp_path / "hdr.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_a_plain_c_caller_links_and_calls_the_host_side_entry_points(tmp_path):
    """what a Julia `ccall` / cgo / JNI stub does, from C: link libscp_mi355x.so and call entry points that need no GPU
    (model registry, default solver options, error paths)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "scptoolbox.jl_amd", "csrc")
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <stdio.h>
#include "scp_mi355x.h"
#include "scp_conic.h"
int main(void) {
    scp_model_info i;
    if (scp_model_query(SCP_MODEL_FREEFLYER, &i) != SCP_OK) return 1;
    printf("%d %d %d %d %d %d\n", i.nx, i.nu, i.np, i.np_node, i.has_subproblem, i.structured);
    { double par[64]; int nq = -1, k; for (k = 0; k < 64; k++) par[k] = 1.0;
      if (i.npar > 64 || scp_model_state_indicators(SCP_MODEL_FREEFLYER, par, 50, &nq) != SCP_OK || nq != 10) return 5; }
    if (scp_model_query(SCP_MODEL_ROCKET_LANDING, &i) != SCP_OK) return 2;
    printf("%d %d %d %d %d\n", i.nx, i.nu, i.np, i.has_subproblem, i.structured);
    if (scp_model_query(42, &i) != SCP_ERR_UNKNOWN_MODEL) return 3;
    scp_conic_opts o;
    scp_conic_default_opts(&o);
    printf("%d %g\n", o.max_iter, o.feastol);
    scp_handle h = 0;
    if (scp_problem_create(0, &h) != SCP_ERR_BAD_ARGUMENT) return 4;
    return 0;
}
''')
    exe = tmp_path / "caller"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), "-L", libdir,
                        "-lscp_mi355x", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr)
    lines = r.stdout.split("\n")
    assert lines[0] == "13 6 1 6 1 0" and lines[1] == "7 4 1 1 1" and lines[2] == "100 1e-08"


def test_julia_shim_parameter_structs_have_the_headers_layout(tmp_path):
    """INTEGRATION.md section 2.2 declares ConicOpts / SCvxParams / GuSTOParams / PTRParams as Julia structs; the same field lists
    as ctypes Structures (Julia's isbits struct layout is C's) must have the size and the field offsets gcc computes from the headers."""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    I, D = C.c_int, C.c_double

    class ConicOpts(C.Structure):
        _fields_ = [("max_iter", I), ("feastol", D), ("abstol", D), ("reltol", D), ("reg", D), ("dyn_eps", D), ("dyn_delta", D),
                    ("nref", I), ("ref_tol", D), ("step", D)]

    class SCvxParams(C.Structure):
        _fields_ = [("iter_max", I), ("lam", D), ("rho_0", D), ("rho_1", D), ("rho_2", D), ("beta_sh", D), ("beta_gr", D),
                    ("eta_init", D), ("eta_lb", D), ("eta_ub", D), ("eps_abs", D), ("eps_rel", D), ("q_exit", D), ("solver", ConicOpts)]

    class GuSTOParams(C.Structure):
        _fields_ = [("iter_max", I), ("lam_init", D), ("lam_max", D), ("rho_0", D), ("rho_1", D), ("beta_sh", D), ("beta_gr", D),
                    ("gamma_fail", D), ("eta_init", D), ("eta_lb", D), ("eta_ub", D), ("mu", D), ("iter_mu", I), ("eps_abs", D),
                    ("eps_rel", D), ("q_tr", D), ("q_exit", D), ("nst", I), ("solver", ConicOpts), ("pen", I), ("hom", D)]

    class PTRParams(C.Structure):
        _fields_ = [("iter_max", I), ("wvc", D), ("wtr", D), ("eps_abs", D), ("eps_rel", D), ("q_tr", D), ("q_exit", D),
                    ("ipm_max_iter", I), ("ipm_feastol", D), ("ipm_abstol", D), ("ipm_reltol", D), ("ipm_reg", D), ("ipm_nref", I),
                    ("ipm_ref_gap", D), ("ipm_ref_tol", D), ("ipm_stall", I), ("ipm_split_step", I), ("ipm_warm", I), ("ipm_warm_mu", D),
                    ("ipm_warm_dev", D), ("ipm_warm_min_cold", I), ("ipm_wpe", I), ("ipm_warm_mu_coarse", D),
                    ("ipm_warm_mu_mid", D), ("ipm_warm_dev_mid", D), ("ipm_warm_mu_vfine", D), ("ipm_warm_dev_vfine", D)]
    probes = (("scp_conic_opts", ConicOpts, ("max_iter", "reg", "nref", "step")),
              ("scp_scvx_params", SCvxParams, ("lam", "eta_ub", "q_exit", "solver")),
              ("scp_gusto_params", GuSTOParams, ("lam_init", "iter_mu", "eps_abs", "nst", "solver", "pen", "hom")),
              ("scp_ptr_params", PTRParams, ("wvc", "q_exit", "ipm_max_iter", "ipm_nref", "ipm_ref_gap", "ipm_warm", "ipm_wpe", "ipm_warm_mu_coarse", "ipm_warm_dev_vfine")))
    body = "".join('  printf("%%zu", sizeof(%s));%s  printf("\\n");\n' % (
        ct, "".join(' printf(" %%zu", offsetof(%s, %s));' % (ct, f) for f in fields)) for ct, _, fields in probes)
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "scp_mi355x.h"\n#include "scp_conic.h"\nint main(void) {\n' + body + "  return 0; }\n")
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip().split("\n")
    for line, (ct, cls, fields) in zip(out, probes):
        want = [C.sizeof(cls)] + [getattr(cls, f).offset for f in fields]
        assert [int(v) for v in line.split()] == want, (ct, line, want)
