"""ctypes binding of the C-ABI library (include/scp_mi355x.h).

The product path is HIP-only: if libscp_mi355x.so is missing or does not export
every symbol of the header, importing this module's `lib()` fails loudly --
there is no CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCP_MI355X_LIB selects another build of the SAME library (e.g. the diagnostic `make prof` build)
LIB_PATH = os.environ.get("SCP_MI355X_LIB") or os.path.join(_HERE, "csrc", "libscp_mi355x.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_u8_p = ctypes.POINTER(ctypes.c_uint8)


class ScpModelInfo(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int), ("nu", ctypes.c_int), ("np", ctypes.c_int), ("npF", ctypes.c_int),
                ("Fcols", ctypes.c_int * 8), ("ns", ctypes.c_int), ("nic", ctypes.c_int), ("ntc", ctypes.c_int),
                ("npar", ctypes.c_int), ("npp", ctypes.c_int), ("nl", ctypes.c_int), ("nsoc", ctypes.c_int),
                ("ng", ctypes.c_int), ("structured", ctypes.c_int), ("has_subproblem", ctypes.c_int),
                ("np_node", ctypes.c_int), ("global_rows_in_X", ctypes.c_int), ("linf_groups", ctypes.c_int),
                ("linf_rows", ctypes.c_int), ("s_input_free", ctypes.c_int)]


class ScpScaling(ctypes.Structure):
    _fields_ = [("Sx", c_double_p), ("cx", c_double_p), ("Su", c_double_p), ("cu", c_double_p),
                ("Sp", c_double_p), ("cp", c_double_p)]


class ScpProblemDesc(ctypes.Structure):
    _fields_ = [("model_id", ctypes.c_int), ("model_par", c_double_p), ("N", ctypes.c_int), ("Nsub", ctypes.c_int),
                ("disc_method", ctypes.c_int), ("feas_tol", ctypes.c_double), ("scale", ScpScaling),
                ("batch_capacity", ctypes.c_int), ("device", ctypes.c_int)]


class ScpPtrParams(ctypes.Structure):
    """scp_ptr_params (include/scp_mi355x.h)."""
    _fields_ = [("iter_max", ctypes.c_int), ("wvc", ctypes.c_double), ("wtr", ctypes.c_double),
                ("eps_abs", ctypes.c_double), ("eps_rel", ctypes.c_double), ("q_tr", ctypes.c_double),
                ("q_exit", ctypes.c_double), ("ipm_max_iter", ctypes.c_int), ("ipm_feastol", ctypes.c_double),
                ("ipm_abstol", ctypes.c_double), ("ipm_reltol", ctypes.c_double), ("ipm_reg", ctypes.c_double),
                ("ipm_nref", ctypes.c_int), ("ipm_ref_gap", ctypes.c_double), ("ipm_ref_tol", ctypes.c_double),
                ("ipm_stall", ctypes.c_int), ("ipm_split_step", ctypes.c_int), ("ipm_warm", ctypes.c_int),
                ("ipm_warm_mu", ctypes.c_double), ("ipm_warm_dev", ctypes.c_double), ("ipm_warm_min_cold", ctypes.c_int),
                ("ipm_wpe", ctypes.c_int), ("ipm_warm_mu_coarse", ctypes.c_double),
                ("ipm_warm_mu_mid", ctypes.c_double), ("ipm_warm_dev_mid", ctypes.c_double), ("ipm_warm_mu_vfine", ctypes.c_double),
                ("ipm_warm_dev_vfine", ctypes.c_double)]


class ScpConicOpts(ctypes.Structure):
    """scp_conic_opts (include/scp_conic.h)."""
    _fields_ = [("max_iter", ctypes.c_int), ("feastol", ctypes.c_double), ("abstol", ctypes.c_double),
                ("reltol", ctypes.c_double), ("reg", ctypes.c_double), ("dyn_eps", ctypes.c_double),
                ("dyn_delta", ctypes.c_double), ("nref", ctypes.c_int), ("ref_tol", ctypes.c_double),
                ("step", ctypes.c_double)]


class ScpAffineMap(ctypes.Structure):
    """scp_affine_map (include/scp_mi355x.h)."""
    _fields_ = [("len", ctypes.c_int), ("val0", ctypes.c_void_p), ("ptr", ctypes.c_void_p), ("sidx", ctypes.c_void_p),
                ("coef", ctypes.c_void_p)]


class ScpSubTemplate(ctypes.Structure):
    """scp_sub_template (include/scp_mi355x.h)."""
    _fields_ = [("n", ctypes.c_int), ("p", ctypes.c_int), ("m", ctypes.c_int), ("l", ctypes.c_int), ("ncones", ctypes.c_int),
                ("q", ctypes.c_void_p), ("Pp", ctypes.c_void_p), ("Pi", ctypes.c_void_p), ("Ap", ctypes.c_void_p),
                ("Ai", ctypes.c_void_p), ("Gp", ctypes.c_void_p), ("Gi", ctypes.c_void_p),
                ("c", ScpAffineMap), ("b", ScpAffineMap), ("h", ScpAffineMap), ("Gx", ScpAffineMap), ("Ax", ScpAffineMap),
                ("Px", ScpAffineMap), ("nsrc", ctypes.c_int), ("nscal", ctypes.c_int), ("ix", ctypes.c_void_p),
                ("iu", ctypes.c_void_p), ("ip", ctypes.c_void_p), ("nfun", ctypes.c_int), ("fun", ScpAffineMap)]


class ScpScvxParams(ctypes.Structure):
    """scp_scvx_params (include/scp_mi355x.h)."""
    _fields_ = [("iter_max", ctypes.c_int), ("lam", ctypes.c_double), ("rho_0", ctypes.c_double), ("rho_1", ctypes.c_double),
                ("rho_2", ctypes.c_double), ("beta_sh", ctypes.c_double), ("beta_gr", ctypes.c_double),
                ("eta_init", ctypes.c_double), ("eta_lb", ctypes.c_double), ("eta_ub", ctypes.c_double),
                ("eps_abs", ctypes.c_double), ("eps_rel", ctypes.c_double), ("q_exit", ctypes.c_double), ("solver", ScpConicOpts)]


class ScpPtrGenericParams(ctypes.Structure):
    """scp_ptr_generic_params (include/scp_mi355x.h)."""
    _fields_ = [("iter_max", ctypes.c_int), ("wvc", ctypes.c_double), ("wtr", ctypes.c_double), ("eps_abs", ctypes.c_double),
                ("eps_rel", ctypes.c_double), ("q_exit", ctypes.c_double), ("cost_const", ctypes.c_double), ("solver", ScpConicOpts)]


class ScpGustoParams(ctypes.Structure):
    """scp_gusto_params (include/scp_mi355x.h)."""
    _fields_ = [("iter_max", ctypes.c_int), ("lam_init", ctypes.c_double), ("lam_max", ctypes.c_double),
                ("rho_0", ctypes.c_double), ("rho_1", ctypes.c_double), ("beta_sh", ctypes.c_double),
                ("beta_gr", ctypes.c_double), ("gamma_fail", ctypes.c_double), ("eta_init", ctypes.c_double),
                ("eta_lb", ctypes.c_double), ("eta_ub", ctypes.c_double), ("mu", ctypes.c_double), ("iter_mu", ctypes.c_int),
                ("eps_abs", ctypes.c_double), ("eps_rel", ctypes.c_double), ("q_tr", ctypes.c_double), ("q_exit", ctypes.c_double),
                ("nst", ctypes.c_int), ("solver", ScpConicOpts), ("pen", ctypes.c_int), ("hom", ctypes.c_double)]


HIST_WIDTH = 16
SCVX_HIST_WIDTH = 16

# every symbol include/scp_mi355x.h declares
EXPORTS = [
    "scp_model_query", "scp_model_rows", "scp_model_state_indicators", "scp_model_eval_host", "scp_problem_create", "scp_problem_destroy", "scp_sync", "scp_last_error", "scp_set_stream_priority",
    "scp_discretize_batch_host", "scp_discretize_batch_dev", "scp_set_discretize_precision",
    "scp_ptr_init_host", "scp_ptr_iterate", "scp_ptr_get_host", "scp_ptr_solve_batch_host",
    "scp_ptr_solve_subproblem_batch_host", "scp_debug_get_stage_problem", "scp_ptr_restart", "scp_get_kernel_timing", "scp_debug_get_ipm_profile", "scp_propagate_batch_host", "scp_ptr_init_guess_host",
    "scp_ptr_get_virtual_controls_host", "scp_ptr_iterate_async", "scp_ptr_poll", "scp_ptr_poll_iteration", "scp_guess_batch_host", "scp_guess_failures",
    "scp_sub_source_layout", "scp_sub_create", "scp_sub_destroy", "scp_sub_stats", "scp_sub_last_error", "scp_sub_solve_batch_host",
    "scp_scvx_init_host", "scp_scvx_iterate", "scp_scvx_get_host",
    "scp_gusto_init_host", "scp_gusto_iterate", "scp_gusto_get_host",
    "scp_ptr_generic_init_host", "scp_ptr_generic_iterate", "scp_ptr_generic_get_host",
    "scp_comm_unique_id", "scp_comm_preflight", "scp_comm_create", "scp_comm_destroy", "scp_comm_last_error", "scp_comm_all_reduce_sum_i64", "scp_shard_range",
    "scp_ptr_run_sharded",
    # include/scp_conic.h
    "scp_conic_default_opts", "scp_conic_create", "scp_conic_destroy", "scp_conic_last_error", "scp_conic_stats",
    "scp_conic_solve_batch_host", "socp_solve_batch",
]

STATUS = {0: "SCP_OK", 1: "SCP_ERR_BAD_ARGUMENT", 2: "SCP_ERR_UNKNOWN_MODEL", 3: "SCP_ERR_NO_DEVICE",
          4: "SCP_ERR_HIP", 5: "SCP_ERR_ALLOC", 6: "SCP_ERR_BATCH_TOO_LARGE", 7: "SCP_ERR_UNSUPPORTED", 8: "SCP_ERR_PEER"}


class ScpError(RuntimeError):
    """Mirror of SCPError (src/utils/globals.jl:52-56) for C-ABI failures."""

    def __init__(self, code, msg=""):
        self.code = code
        super().__init__("%s (%d) %s" % (STATUS.get(code, "?"), code, msg))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "HIP extension %s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the product path)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name in EXPORTS:
            if not hasattr(L, name):
                raise ImportError("libscp_mi355x.so does not export %s" % name)
        L.scp_last_error.restype = ctypes.c_char_p
        L.scp_last_error.argtypes = [ctypes.c_void_p]
        L.scp_model_query.argtypes = [ctypes.c_int, ctypes.POINTER(ScpModelInfo)]
        L.scp_model_rows.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 8
        L.scp_model_state_indicators.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, c_int_p]
        L.scp_model_eval_host.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 12 + [c_int_p]
        L.scp_problem_create.argtypes = [ctypes.POINTER(ScpProblemDesc), ctypes.POINTER(ctypes.c_void_p)]
        L.scp_problem_destroy.argtypes = [ctypes.c_void_p]
        L.scp_sync.argtypes = [ctypes.c_void_p]
        L.scp_set_stream_priority.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.scp_discretize_batch_host.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 11 + [c_double_p]
        L.scp_discretize_batch_dev.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 11
        PP = ctypes.POINTER(ScpPtrParams)
        L.scp_ptr_init_host.argtypes = [ctypes.c_void_p, ctypes.c_int, PP] + [ctypes.c_void_p] * 4
        L.scp_ptr_iterate.argtypes = [ctypes.c_void_p, c_int_p]
        L.scp_ptr_iterate_async.argtypes = [ctypes.c_void_p]
        L.scp_ptr_poll.argtypes = [ctypes.c_void_p, c_int_p]
        L.scp_ptr_poll_iteration.argtypes = [ctypes.c_void_p, ctypes.c_int, c_int_p]
        L.scp_ptr_get_host.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 9
        L.scp_ptr_solve_batch_host.argtypes = [ctypes.c_void_p, ctypes.c_int, PP] + [ctypes.c_void_p] * 11 + [c_double_p]
        L.scp_ptr_solve_subproblem_batch_host.argtypes = ([ctypes.c_void_p, ctypes.c_int, PP] + [ctypes.c_void_p] * 14
                                                          + [c_double_p])
        L.scp_ptr_restart.argtypes = [ctypes.c_void_p]
        L.scp_ptr_init_guess_host.argtypes = [ctypes.c_void_p, ctypes.c_int, PP, ctypes.c_void_p]
        L.scp_propagate_batch_host.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
        L.scp_ptr_get_virtual_controls_host.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 6
        L.scp_get_kernel_timing.argtypes = [ctypes.c_void_p, c_double_p, ctypes.POINTER(ctypes.c_long), ctypes.c_int]
        L.scp_debug_get_stage_problem.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.POINTER(ctypes.c_long)]
        L.scp_sub_source_layout.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, c_int_p]
        L.scp_sub_create.argtypes = [ctypes.c_void_p, ctypes.POINTER(ScpSubTemplate), ctypes.POINTER(ctypes.c_void_p)]
        L.scp_sub_destroy.argtypes = [ctypes.c_void_p]
        L.scp_sub_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.scp_sub_last_error.argtypes = [ctypes.c_void_p]
        L.scp_sub_last_error.restype = ctypes.c_char_p
        L.scp_sub_solve_batch_host.argtypes = ([ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5
                                               + [ctypes.POINTER(ScpConicOpts)] + [ctypes.c_void_p] * 10 + [c_double_p])
        L.scp_scvx_init_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ScpScvxParams)] + [ctypes.c_void_p] * 4
        L.scp_scvx_iterate.argtypes = [ctypes.c_void_p, c_int_p]
        L.scp_scvx_get_host.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 9
        L.scp_set_discretize_precision.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.scp_guess_batch_host.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
        L.scp_guess_failures.argtypes = [ctypes.c_void_p]
        L.scp_gusto_init_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ScpGustoParams)] + [ctypes.c_void_p] * 4
        L.scp_gusto_iterate.argtypes = [ctypes.c_void_p, c_int_p]
        L.scp_gusto_get_host.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 9
        L.scp_ptr_generic_init_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ScpPtrGenericParams)] + [ctypes.c_void_p] * 4
        L.scp_ptr_generic_iterate.argtypes = [ctypes.c_void_p, c_int_p]
        L.scp_ptr_generic_get_host.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 9
        L.scp_comm_unique_id.argtypes = [ctypes.c_void_p]
        L.scp_comm_preflight.argtypes = [ctypes.c_int]
        L.scp_comm_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.scp_comm_destroy.argtypes = [ctypes.c_void_p]
        L.scp_comm_destroy.restype = None
        L.scp_comm_last_error.argtypes = [ctypes.c_void_p]
        L.scp_comm_last_error.restype = ctypes.c_char_p
        L.scp_comm_all_reduce_sum_i64.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
        L.scp_shard_range.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]
        L.scp_shard_range.restype = None
        L.scp_ptr_run_sharded.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, c_int_p, c_int_p]
        L.scp_conic_default_opts.argtypes = [ctypes.POINTER(ScpConicOpts)]
        L.scp_conic_default_opts.restype = None
        L.scp_conic_create.argtypes = ([ctypes.c_int] * 5 + [ctypes.c_void_p] * 8 + [ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_void_p)])
        L.scp_conic_destroy.argtypes = [ctypes.c_void_p]
        L.scp_conic_last_error.argtypes = [ctypes.c_void_p]
        L.scp_conic_last_error.restype = ctypes.c_char_p
        L.scp_conic_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.scp_conic_solve_batch_host.argtypes = ([ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_uint,
                                                 ctypes.POINTER(ScpConicOpts)] + [ctypes.c_void_p] * 7 + [c_double_p])
        L.socp_solve_batch.argtypes = ([ctypes.c_int] * 5 + [ctypes.c_void_p] * 10 + [ctypes.c_int] + [ctypes.c_void_p] * 5)
        _lib = L
    return _lib


def check(rc, handle=None):
    if rc != 0:
        msg = ""
        if handle:
            msg = lib().scp_last_error(handle).decode(errors="replace")
        raise ScpError(rc, msg)
