"""ctypes binding of libegc_hip.so (the C ABI declared in include/egc_hip.h).

There is NO CPU fallback: if the shared library is missing or cannot be loaded, every entry point
raises RuntimeError.  Build it with ``egc_amd/csrc/build.sh`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_LIB_PATH = os.environ.get("EGC_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                         "libegc_hip.so")
_lib = None

EGC_MAX_AGGRS = 8
LONG_ROW_THRESHOLD = 64
LONG_ROW_CHUNK = 256

# enum egc_aggr
AGGR_SUM, AGGR_MEAN, AGGR_MAX, AGGR_MIN, AGGR_VAR, AGGR_STD, AGGR_SYMNORM = range(7)
# enum egc_edge_set
SET_RAW, SET_LOOPED = 0, 1
# enum egc_weight_layout
LAYOUT_HBA, LAYOUT_HAB = 0, 1
# enum egc_weight_act
ACT_NONE, ACT_SOFTMAX, ACT_SIGMOID, ACT_HARDTANH = range(4)

_STATUS = {1: "EGC_ERR_INVALID", 2: "EGC_ERR_WORKSPACE", 3: "EGC_ERR_HIP", 4: "EGC_ERR_UNSUPPORTED"}


class EgcGraph(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int64), ("n_edges", C.c_int64),
        ("rowptr", C.c_void_p), ("col", C.c_void_p), ("edge_id", C.c_void_p),
        ("dis_raw", C.c_void_p), ("dis_looped", C.c_void_p),
        ("max_index", C.c_void_p), ("plan", C.c_void_p), ("n_chunks", C.c_int64),
        ("n_src_rows", C.c_int64), ("edge_dis_raw", C.c_void_p), ("edge_dis_looped", C.c_void_p),
    ]


class EgcLayer(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("num_heads", C.c_int32),
        ("num_bases", C.c_int32), ("num_aggrs", C.c_int32), ("aggrs", C.c_int32 * EGC_MAX_AGGRS),
        ("agg_set", C.c_int32), ("sym_set", C.c_int32), ("loops_all_nodes", C.c_int32),
        ("weight_layout", C.c_int32), ("weight_act", C.c_int32), ("basis_stride", C.c_int32),
    ]


_ENV_DATA = getattr(os.environ, "_data", None)     # CPython on POSIX: the bytes dict behind os.environ


def env_flag(name: str) -> bool:
    """True when the environment variable is set to something other than "" / "0" -- read on every call (tests and
    notebooks flip these), but through the bytes dict behind os.environ where CPython has one: os.environ.get costs
    1.5 us per lookup (key encoding + value decoding), four of them were a quarter of an eval-mode layer call."""
    if _ENV_DATA is not None:
        v = _ENV_DATA.get(name.encode())
        return v is not None and v not in (b"", b"0")
    return os.environ.get(name, "0") not in ("", "0")


class EgcPost(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("residual", C.c_void_p), ("relu", C.c_int32)]


# name -> (restype, argtypes): exactly the symbols include/egc_hip.h declares
SYMBOLS = {
    "egc_plan_ints": (C.c_int64, [C.c_int64, C.c_int64]),
    "egc_coo_to_csr_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "egc_coo_to_csr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_coo_to_csr_checked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_graph_build_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "egc_graph_build_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "egc_graph_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_csr_edge_dis": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egc_csr_prepare": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "egc_bases_ld": (C.c_int32, [C.POINTER(EgcLayer)]),
    "egc_basis_transform_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "egc_basis_pack_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "egc_basis_pack": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_layer_gemm_flags": (C.c_int32, [C.POINTER(EgcLayer)]),
    "egc_basis_pack_ex": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_basis_transform_packed_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                                C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "egc_basis_transform_packed_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                                 C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "egc_basis_pack_transposed": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                            C.c_void_p]),
    "egc_basis_transform_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "egc_layer_forward_packed": (C.c_int, [C.POINTER(EgcGraph), C.POINTER(EgcLayer), C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_aggregate_workspace_bytes": (C.c_size_t, [C.POINTER(EgcLayer), C.c_int64, C.c_int64]),
    "egc_aggregate_workspace_zero_bytes": (C.c_size_t, [C.POINTER(EgcLayer), C.c_int64, C.c_int64]),
    "egc_aggregate_workspace_bytes_for": (C.c_size_t, [C.POINTER(EgcLayer), C.POINTER(EgcGraph)]),
    "egc_aggregate_combine_f32": (C.c_int, [C.POINTER(EgcGraph), C.POINTER(EgcLayer), C.c_void_p, C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_layer_forward_f32": (C.c_int, [C.POINTER(EgcGraph), C.POINTER(EgcLayer), C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_backward_workspace_bytes": (C.c_size_t, [C.POINTER(EgcLayer), C.c_int64]),
    "egc_backward_workspace_bytes_for": (C.c_size_t, [C.POINTER(EgcLayer), C.POINTER(EgcGraph)]),
    "egc_aggregate_combine_rows_f32": (C.c_int, [C.POINTER(EgcGraph), C.POINTER(EgcLayer), C.c_void_p, C.c_int32,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_aggregate_combine_post_f32": (C.c_int, [C.POINTER(EgcGraph), C.POINTER(EgcLayer), C.c_void_p, C.c_int32,
                                                 C.c_void_p, C.c_void_p, C.POINTER(EgcPost), C.c_void_p,
                                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_aggregate_combine_strided_f32": (C.c_int, [C.POINTER(EgcGraph), C.POINTER(EgcLayer), C.c_void_p, C.c_int32,
                                                    C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(EgcPost), C.c_void_p,
                                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_weight_grad_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "egc_weight_grad_plan": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "egc_weight_grad_ex_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "egc_weight_grad_ex_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_void_p]),
    "egc_weight_grad_params_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_int64, C.c_void_p]),
    "egc_weight_grad_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "egc_csr_transposed_coo": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egc_column_moments_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_float,
                                         C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egc_bn_forward_stats_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "egc_bn_backward_stats_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_int64,
                                            C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "egc_bn_backward_stats_sums_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_int64,
                                                 C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "egc_bn_forward_finalize": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_double,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "egc_bn_backward_finalize": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "egc_affine_act_residual_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_float,
                                              C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egc_affine_act_backward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_float,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "egc_weights_pack_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p]),
    "egc_sum_partials_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "egc_column_sums_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "egc_segment_mean_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "egc_train_stats_floats": (C.c_int64, [C.POINTER(EgcLayer)]),
    "egc_aggregate_combine_train_f32": (C.c_int, [C.POINTER(EgcGraph), C.POINTER(EgcLayer), C.c_void_p, C.c_int32,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_aggregate_combine_train_rows_f32": (C.c_int, [C.POINTER(EgcGraph), C.POINTER(EgcLayer), C.c_void_p, C.c_int32,
                                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t,
                                                       C.c_void_p]),
    "egc_batch_tile_nodes": (C.c_int32, [C.POINTER(EgcLayer), C.c_int32, C.c_int32, C.c_int32]),
    "egc_batch_plan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
                                 C.c_int32, C.c_void_p, C.c_void_p]),
    "egc_aggregate_combine_batch_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                                  C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(EgcLayer), C.c_void_p, C.c_int32,
                                                  C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(EgcPost), C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p]),
    "egc_batch_fused_tile_nodes": (C.c_int32, [C.POINTER(EgcLayer), C.c_int32, C.c_int32]),
    "egc_batch_fused_tile_quantum": (C.c_int32, [C.POINTER(EgcLayer)]),
    "egc_batch_fused_pack_bytes": (C.c_int64, [C.POINTER(EgcLayer)]),
    "egc_batch_fused_pack": (C.c_int, [C.POINTER(EgcLayer), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "egc_layer_forward_batch_fused_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                                    C.c_int64, C.c_void_p, C.POINTER(EgcLayer), C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.POINTER(EgcPost), C.c_void_p, C.c_int32, C.c_int32,
                                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "egc_batch_fused_bwd_tile_nodes": (C.c_int32, [C.POINTER(EgcLayer), C.c_int32]),
    "egc_batch_fused_bwd_pack_bytes": (C.c_int64, [C.POINTER(EgcLayer)]),
    "egc_batch_fused_bwd_pack": (C.c_int, [C.POINTER(EgcLayer), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "egc_batch_fused_train_pack": (C.c_int, [C.POINTER(EgcLayer), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                             C.c_void_p]),
    "egc_batch_fused_train_pack_params": (C.c_int, [C.POINTER(EgcLayer), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                                    C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "egc_layer_backward_batch_fused_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                                     C.c_void_p, C.POINTER(EgcLayer), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                                     C.c_void_p, C.c_void_p]),
    "egc_gather_rows_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "egc_aggregate_combine_backward_f32": (C.c_int, [C.POINTER(EgcGraph), C.POINTER(EgcGraph), C.POINTER(EgcLayer),
                                                     C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                     C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egc_last_error": (C.c_char_p, []),
    "egc_version": (C.c_char_p, []),
}


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load libegc_hip.so once; raise RuntimeError (never fall back) if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"egc_amd: HIP library not built: {_LIB_PATH} is missing. Run egc_amd/csrc/build.sh "
            "(hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    try:
        lib = C.CDLL(_LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise RuntimeError(f"egc_amd: cannot load {_LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int, what: str):
    if status != 0:
        detail = load().egc_last_error().decode() if status == 3 else ""
        raise RuntimeError(f"egc_amd: {what} failed with {_STATUS.get(status, status)} {detail}".rstrip())


def make_layer(in_channels, out_channels, num_heads, num_bases, aggr_codes, agg_set, sym_set, loops_all_nodes,
               weight_layout, weight_act, basis_stride=0) -> EgcLayer:
    if len(aggr_codes) > EGC_MAX_AGGRS:
        raise RuntimeError(f"egc_amd: at most {EGC_MAX_AGGRS} aggregators are supported")
    arr = (C.c_int32 * EGC_MAX_AGGRS)(*(list(aggr_codes) + [0] * (EGC_MAX_AGGRS - len(aggr_codes))))
    return EgcLayer(in_channels, out_channels, num_heads, num_bases, len(aggr_codes), arr, agg_set, sym_set,
                    int(loops_all_nodes), weight_layout, weight_act, int(basis_stride))
