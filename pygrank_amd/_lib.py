"""ctypes binding of the C-ABI declared in include/pgh.h.

The product binds exactly one library: ``pygrank_amd/csrc/libpgh_hip.so`` (hand-written HIP for gfx950).
There is NO CPU fallback: if the library is missing, if it reports a runtime other than ``hip:*``, or if no MI355X is
visible when the engine is first used, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpgh_hip.so")

c_vec = C.c_void_p
c_mat = C.c_void_p
c_graph = C.c_void_p
c_timer = C.c_void_p
c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)


NORM_COL, NORM_SYMMETRIC, NORM_NONE, NORM_BOTH, NORM_LAPLACIAN = 0, 1, 2, 3, 4      # include/pgh.h PGH_NORM_*


class LoopCfg(C.Structure):
    _fields_ = [("alpha", C.c_double), ("use_quotient", C.c_int32), ("err_kind", C.c_int32), ("tol", C.c_double),
                ("max_iters", C.c_int32), ("end_modulo", C.c_int32), ("out_scale", C.c_double),
                ("in_norm", C.c_double), ("start_from_p", C.c_int32), ("reserved", C.c_int32)]


class LoopResult(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("spmv_count", C.c_int32),
                ("flags", C.c_int32), ("last_error", C.c_double), ("loop_ms", C.c_double), ("in_norm", C.c_double)]


class DistCfg(C.Structure):
    _fields_ = [("alpha", C.c_double), ("tol", C.c_double), ("n_global", C.c_int64), ("err_kind", C.c_int32), ("max_iters", C.c_int32),
                ("end_modulo", C.c_int32), ("use_quotient", C.c_int32), ("preserve_norm", C.c_int32), ("every_row", C.c_int32),
                ("deg_local", C.c_void_p), ("lam_local", C.c_void_p)]


class DistResult(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("spmv_count", C.c_int32), ("converged", C.c_int32), ("column_blocks", C.c_int32),
                ("split_regions", C.c_int32), ("flags", C.c_int32), ("last_error", C.c_double), ("loop_ms", C.c_double),
                ("exchange_bytes", C.c_int64), ("gather_slots", C.c_int64)]


COMM_ID_BYTES = 128
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                           C.POINTER(C.c_int64), C.c_void_p)

# name -> (restype, argtypes); every symbol include/pgh.h declares
SIGNATURES = {
    "pgh_init": (C.c_int, [C.c_int]),
    "pgh_shutdown": (C.c_int, []),
    "pgh_last_error": (C.c_char_p, []),
    "pgh_runtime_name": (C.c_char_p, []),
    "pgh_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pgh_device_name": (C.c_int, [C.c_char_p, C.c_int]),
    "pgh_mem_info": (C.c_int, [c_i64p, c_i64p]),
    "pgh_set_stream": (C.c_int, [C.c_void_p]),
    "pgh_sync": (C.c_int, []),
    "pgh_timer_create": (C.c_int, [C.POINTER(c_timer)]),
    "pgh_timer_destroy": (C.c_int, [c_timer]),
    "pgh_timer_start": (C.c_int, [c_timer]),
    "pgh_timer_stop": (C.c_int, [c_timer]),
    "pgh_timer_elapsed_ms": (C.c_int, [c_timer, c_f64p]),
    "pgh_profile_enable": (C.c_int, [C.c_int]),
    "pgh_profile_reset": (C.c_int, []),
    "pgh_profile_read": (C.c_int, [C.c_int, c_i64p, c_f64p]),
    "pgh_vec_alloc": (C.c_int, [C.c_int64, C.POINTER(c_vec)]),
    "pgh_vec_wrap": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(c_vec)]),
    "pgh_vec_free": (C.c_int, [c_vec]),
    "pgh_vec_len": (C.c_int64, [c_vec]),
    "pgh_vec_ptr": (C.c_void_p, [c_vec]),
    "pgh_vec_h2d_f32": (C.c_int, [c_vec, C.c_void_p, C.c_int64]),
    "pgh_vec_h2d_f64": (C.c_int, [c_vec, C.c_void_p, C.c_int64]),
    "pgh_vec_d2h_f32": (C.c_int, [c_vec, C.c_void_p, C.c_int64]),
    "pgh_vec_d2h_f64": (C.c_int, [c_vec, C.c_void_p, C.c_int64]),
    "pgh_vec_fill": (C.c_int, [c_vec, C.c_double]),
    "pgh_vec_copy": (C.c_int, [c_vec, c_vec]),
    "pgh_vec_get": (C.c_int, [c_vec, C.c_int64, c_f64p]),
    "pgh_vec_set": (C.c_int, [c_vec, C.c_int64, C.c_double]),
    "pgh_vec_scatter_set": (C.c_int, [c_vec, C.c_void_p, C.c_void_p, C.c_int64]),
    "pgh_ewise_vv": (C.c_int, [C.c_int, c_vec, c_vec, c_vec]),
    "pgh_ewise_vs": (C.c_int, [C.c_int, c_vec, C.c_double, C.c_int, c_vec]),
    "pgh_ewise_unary": (C.c_int, [C.c_int, c_vec, c_vec]),
    "pgh_axpby": (C.c_int, [C.c_double, c_vec, C.c_double, c_vec, c_vec]),
    "pgh_filter_out": (C.c_int, [c_vec, c_vec, c_vec, c_i64p]),
    "pgh_vec_ordinals": (C.c_int, [c_vec, c_vec]),
    "pgh_vec_kth_largest": (C.c_int, [c_vec, C.c_int64, c_f64p]),
    "pgh_auc": (C.c_int, [c_vec, c_vec, c_f64p, c_i64p]),
    "pgh_vec_gap_threshold": (C.c_int, [c_vec, c_f64p]),
    "pgh_reduce": (C.c_int, [C.c_int, c_vec, c_f64p]),
    "pgh_dot": (C.c_int, [c_vec, c_vec, c_f64p]),
    "pgh_residual": (C.c_int, [C.c_int, c_vec, c_vec, c_f64p]),
    "pgh_mat_alloc": (C.c_int, [C.c_int64, C.c_int32, C.POINTER(c_mat)]),
    "pgh_mat_free": (C.c_int, [c_mat]),
    "pgh_mat_shape": (C.c_int, [c_mat, c_i64p, c_i32p]),
    "pgh_mat_ptr": (C.c_void_p, [c_mat]),
    "pgh_mat_h2d_f64": (C.c_int, [c_mat, C.c_void_p]),
    "pgh_mat_d2h_f64": (C.c_int, [c_mat, C.c_void_p]),
    "pgh_mat_set_col": (C.c_int, [c_mat, C.c_int32, c_vec]),
    "pgh_mat_col_abssum": (C.c_int, [c_mat, C.c_void_p]),
    "pgh_mat_div_cols": (C.c_int, [c_mat, C.c_void_p, c_mat]),
    "pgh_mat_gemv": (C.c_int, [c_mat, C.c_void_p, C.c_int32, c_vec]),
    "pgh_mat_gemm": (C.c_int, [c_mat, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_mat]),
    "pgh_mat_get_cols": (C.c_int, [c_mat, C.c_int32, c_mat]),
    "pgh_mat_set_cols": (C.c_int, [c_mat, C.c_int32, c_mat]),
    "pgh_mat_get_col": (C.c_int, [c_mat, C.c_int32, c_vec]),
    "pgh_graph_from_csr": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.POINTER(c_graph)]),
    "pgh_graph_from_csr_part": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                          C.c_void_p, C.POINTER(c_graph)]),
    "pgh_graph_from_factored_csr": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int, C.POINTER(c_graph)]),
    "pgh_graph_from_adjacency": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int,
                                           C.POINTER(c_graph)]),
    "pgh_graph_from_adjacency_ex": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double,
                                              C.c_int, C.POINTER(c_graph)]),
    "pgh_graph_destroy": (C.c_int, [c_graph]),
    "pgh_graph_info": (C.c_int, [c_graph, c_i64p, c_i64p, c_i64p, c_i64p]),
    "pgh_graph_format": (C.c_int, [c_graph, C.c_char_p, C.c_int]),
    "pgh_last_build_profile": (C.c_int, [C.c_char_p, C.c_int]),
    "pgh_graph_degrees": (C.c_int, [c_graph, c_vec]),
    "pgh_graph_degrees_dropout": (C.c_int, [c_graph, C.c_double, C.c_uint64, c_vec]),
    "pgh_graph_download": (C.c_int, [c_graph, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgh_spmv": (C.c_int, [c_graph, c_vec, c_vec]),
    "pgh_spmv_dropout": (C.c_int, [c_graph, c_vec, c_vec, C.c_double, C.c_uint64]),
    "pgh_ppr_step": (C.c_int, [c_graph, c_vec, C.c_double, c_vec, C.c_double, c_vec, c_f64p]),
    "pgh_absorb_step": (C.c_int, [c_graph, c_vec, C.c_double, c_vec, c_vec, c_vec, c_vec, c_f64p]),
    "pgh_poly_step": (C.c_int, [c_graph, c_vec, c_vec, C.c_double, C.c_double, c_vec, C.c_double, C.c_int, c_f64p]),
    "pgh_scaled_residual": (C.c_int, [C.c_int, c_vec, C.c_double, c_vec, C.c_double, c_f64p]),
    "pgh_graph_resident_len": (C.c_int, [c_graph, c_i64p, c_i64p]),
    "pgh_resident_in": (C.c_int, [c_graph, c_vec, C.c_double, c_vec, c_vec]),
    "pgh_resident_gather": (C.c_int, [c_graph, c_vec, c_vec]),
    "pgh_resident_out": (C.c_int, [c_graph, c_vec, C.c_double, c_vec]),
    "pgh_resident_step": (C.c_int, [c_graph, C.c_int32, c_vec, c_vec, C.c_double, c_vec, C.c_double, c_vec, c_vec, c_vec, c_vec, c_f64p]),
    "pgh_ppr_run": (C.c_int, [c_graph, c_vec, c_vec, C.POINTER(LoopCfg), C.POINTER(LoopResult)]),
    "pgh_ppr_run_f64": (C.c_int, [c_graph, c_vec, c_vec, C.POINTER(LoopCfg), C.POINTER(LoopResult)]),
    "pgh_absorb_run_f64": (C.c_int, [c_graph, c_vec, c_vec, c_vec, C.POINTER(LoopCfg), C.POINTER(LoopResult)]),
    "pgh_sarw_run_f64": (C.c_int, [c_graph, c_vec, c_vec, C.POINTER(LoopCfg), C.POINTER(LoopResult)]),
    "pgh_ppr_run_dropout": (C.c_int, [c_graph, c_vec, c_vec, C.POINTER(LoopCfg), C.c_double, C.c_uint64, C.POINTER(LoopResult)]),
    "pgh_absorb_run": (C.c_int, [c_graph, c_vec, c_vec, c_vec, C.POINTER(LoopCfg), C.POINTER(LoopResult)]),
    "pgh_sarw_run": (C.c_int, [c_graph, c_vec, c_vec, C.POINTER(LoopCfg), C.POINTER(LoopResult)]),
    "pgh_poly_run": (C.c_int, [c_graph, c_vec, C.c_void_p, C.c_int32, C.c_int32, c_vec, C.POINTER(LoopCfg),
                               C.POINTER(LoopResult)]),
    "pgh_poly_terms": (C.c_int, [c_graph, c_vec, C.c_int32, C.c_int32, C.c_int32, c_mat, C.c_int32]),
    "pgh_spmm": (C.c_int, [c_graph, c_mat, c_mat]),
    "pgh_ppr_run_batch": (C.c_int, [c_graph, c_mat, c_mat, C.POINTER(LoopCfg), C.c_void_p, C.POINTER(LoopResult)]),
    "pgh_spmm_dropout": (C.c_int, [c_graph, c_mat, c_mat, C.c_double, C.c_uint64]),
    "pgh_ppr_run_batch_dropout": (C.c_int, [c_graph, c_mat, c_mat, C.POINTER(LoopCfg), C.c_void_p, C.c_double, C.c_uint64,
                                            C.POINTER(LoopResult)]),
    "pgh_ppr_step_dist": (C.c_int, [c_graph, c_vec, C.c_double, c_vec, C.c_double, c_vec, c_vec, c_f64p]),
    "pgh_graph_gather_layout": (C.c_int, [c_graph, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_void_p]),
    "pgh_graph_set_gather_bases": (C.c_int, [c_graph, C.c_void_p]),
    "pgh_dist_state_init": (C.c_int, [C.c_void_p]),
    "pgh_dist_partial": (C.c_int, [c_graph, c_vec, C.c_void_p]),
    "pgh_dist_partial_stage": (C.c_int, [c_graph, c_vec, C.c_void_p, C.c_int32]),
    "pgh_graph_hot_prefix": (C.c_int, [c_graph, C.POINTER(C.c_int32)]),
    "pgh_dist_combine": (C.c_int, [c_graph, c_vec, C.c_double, c_vec, c_vec, C.c_void_p]),
    "pgh_dist_combine_absorb": (C.c_int, [c_graph, c_vec, c_vec, c_vec, c_vec, c_vec, C.c_void_p]),
    "pgh_dist_combine_poly": (C.c_int, [c_graph, c_vec, c_vec, C.c_double, C.c_double, c_vec, C.c_double, C.c_int32, c_vec, C.c_void_p]),
    "pgh_dist_close_sum": (C.c_int, [C.c_void_p, C.c_int32]),
    "pgh_dist_watch_isolated": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgh_dist_release_isolated": (C.c_int, [C.c_void_p]),
    "pgh_dist_residual": (C.c_int, [C.c_int32, c_vec, c_vec, C.c_void_p]),
    "pgh_dist_need_counts": (C.c_int, [c_graph, C.c_void_p]),
    "pgh_dist_need_list": (C.c_int, [c_graph, C.c_int32, C.c_void_p]),
    "pgh_dist_set_send_lists": (C.c_int, [c_graph, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "pgh_dist_pack": (C.c_int, [c_graph, c_vec, c_vec]),
    "pgh_dist_compact_from_dense": (C.c_int, [c_graph, C.c_int32, c_vec, C.c_int64, c_vec, C.c_int64]),
    "pgh_dist_close_err": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int64]),
    "pgh_dist_prescale": (C.c_int, [c_graph, c_vec, c_vec]),
    "pgh_graph_perm": (C.c_int, [c_graph, C.c_void_p, c_i64p]),
    "pgh_graph_set_gather_bases_split": (C.c_int, [c_graph, C.c_void_p, C.c_void_p]),
    "pgh_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pgh_comm_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "pgh_comm_create_external": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "pgh_comm_set_alltoallv": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pgh_comm_destroy": (C.c_int, [C.c_void_p]),
    "pgh_dist_ppr_run": (C.c_int, [c_graph, C.c_void_p, c_vec, c_vec, C.POINTER(DistCfg), C.POINTER(DistResult)]),
    "pgh_dist_poly_run": (C.c_int, [c_graph, C.c_void_p, c_vec, C.c_void_p, C.c_int32, c_vec, C.POINTER(DistCfg), C.POINTER(DistResult)]),
    "pgh_dist_set_timeout": (C.c_int, [C.c_double]),
    "pgh_graph_rmat_part": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, C.POINTER(c_graph)]),
    "pgh_graph_rmat": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_int32,
                                 C.c_int32, C.c_int64, C.c_int64, C.POINTER(c_graph)]),
}

# enum values of include/pgh.h
ADD, SUB, MUL, DIV, POW, MAXOP, MINOP, GT, GE, LT, LE, EQ, NE = range(13)
ABS, EXP, LOG, NEG, SQRT, SAFE_INV = range(6)
SUM, ABSSUM, MAX, MIN = range(4)
ERR_MABS, ERR_L1, ERR_LINF, ERR_ITERS = range(4)
K_SPMV, K_FIXUP, K_RESIDUAL, K_FINAL, K_SPMM, K_COMBINE, K_PB_GATHER, K_PB_ACCUM, K_PACK = range(9)

_lib = None
_initialised = False
ACCEPTED_RUNTIMES = ("hip:",)      # pgh_runtime_name() prefixes ensure_init() agrees to drive


class EngineError(Exception):
    """Raised for every non-zero status of the C-ABI (the reference raises plain Exception)."""


def _bind(cdll):
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(cdll, name)      # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    return cdll


def load_library(path=None):
    """dlopen + bind without touching the GPU (used by the build check and the symbol test)."""
    path = LIB_PATH if path is None else path
    if not os.path.exists(path):
        raise EngineError(
            f"MI355X engine library not found at {path}: build it with `make -C pygrank_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    return _bind(C.CDLL(path))


def lib():
    """The bound engine library; loads the HIP build on first use."""
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib


def check(status):
    if status != 0:
        msg = lib().pgh_last_error()
        raise EngineError(msg.decode("utf-8", "replace") if msg else f"engine call failed with status {status}")


def ensure_init(device=None):
    """backend_init(): selects the GPU (LOCAL_RANK when launched by torch.distributed.run) and fails loudly
    when none is visible."""
    global _initialised
    if _initialised:
        return
    L = lib()
    name = L.pgh_runtime_name().decode()
    if not name.startswith(ACCEPTED_RUNTIMES):
        raise EngineError(f"refusing to run the product path on runtime '{name}'")
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    check(L.pgh_init(int(device)))
    _initialised = True


def runtime_name():
    return lib().pgh_runtime_name().decode()
