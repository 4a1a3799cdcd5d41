"""ctypes binding of libhrag.so (include/hrag.h).  No fallback: if the library is missing and
cannot be built the import of the compute path fails loudly."""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhrag.so")

HRAG_OK, HRAG_EINVAL, HRAG_ENOMEM, HRAG_EHIP, HRAG_EBUSY, HRAG_ECAPACITY = range(6)
SEED_STRIDE = 32
HRAG_VERSION = 8      # HRAG_VERSION_MAJOR * 1000 + HRAG_VERSION_MINOR of include/hrag.h
FLAG_DPR_FALLBACK, FLAG_ZERO_MASS, FLAG_ZERO_PHRASE, FLAG_FP8_SATURATED, FLAG_NOT_CONVERGED = 1, 2, 4, 8, 16
# the convergence contract's error bound (include/hrag.h): error <= max(PPR_ERR_K * residual, floor of the state type);
# a tolerance below PPR_TOL_MIN is rejected (HRAG_EINVAL)
PPR_ERR_K, PPR_ERR_FLOOR_FP8, PPR_ERR_FLOOR_F16, PPR_ERR_FLOOR_F32, PPR_TOL_MIN = 3.5, 5e-6, 2e-6, 5e-7, 1e-7
# hrag_opts.flags (include/hrag.h HRAG_OPT_*)
OPT_NATURAL_ROW_ORDER, OPT_NT_CSR, OPT_NT_STORE, OPT_F32_STATE, OPT_TEMPORAL16, OPT_NO_FP8 = 1, 2, 4, 8, 16, 32
OPT_ROWS_BY_MINCOL, OPT_ROWS_BFS, OPT_SLABS_PER_WG_1, OPT_NO_F16, OPT_XCD_BLOCKED = 64, 128, 256, 1024, 2048
OPT_ACCEL = 4096


class HragError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libhrag status {status}: {message}")
        self.status = status


class GraphDesc(C.Structure):
    _fields_ = [("num_vertices", C.c_int64), ("row_offset", C.c_int64), ("n_rows", C.c_int64),
                ("nnz", C.c_int64), ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p),
                ("val", C.c_void_p), ("n_passages", C.c_int64), ("passage_vertex", C.c_void_p),
                ("col_sum", C.c_void_p)]


class EmbedDesc(C.Structure):
    _fields_ = [("rows", C.c_int64), ("row_offset", C.c_int64), ("dim", C.c_int32),
                ("dtype", C.c_int32), ("data", C.c_void_p)]


class FactDesc(C.Structure):
    _fields_ = [("n_facts", C.c_int64), ("subj_vertex", C.c_void_p), ("obj_vertex", C.c_void_p),
                ("num_chunks", C.c_void_p)]


class Opts(C.Structure):
    _fields_ = [("max_batch", C.c_int32), ("max_topk", C.c_int32), ("slab_width", C.c_int32),
                ("long_row_nnz", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32),
                ("segment_nnz", C.c_int32), ("sell_seg_len", C.c_int32), ("sell_sigma", C.c_int32), ("reserved", C.c_int32 * 7)]


class Timings(C.Structure):
    _fields_ = [("fact_sim_ms", C.c_float), ("pass_sim_ms", C.c_float), ("seed_ms", C.c_float),
                ("ppr_ms", C.c_float), ("rank_ms", C.c_float), ("total_ms", C.c_float),
                ("ppr_iters", C.c_int32), ("n_slabs", C.c_int32), ("slab_width", C.c_int32),
                ("n_long_rows", C.c_int32)]


class ShardLayout(C.Structure):
    _fields_ = [("n_slabs", C.c_int32), ("n_groups", C.c_int32), ("slabs_per_group", C.c_int32),
                ("reserved", C.c_int32), ("state_bytes", C.c_int64), ("group_bytes", C.c_int64),
                ("own_offset", C.c_int64), ("own_bytes", C.c_int64)]


class Stats(C.Structure):
    """hrag_stats of include/hrag.h."""
    _fields_ = [("is_workspace", C.c_int32), ("live_workspaces", C.c_int32), ("index_bytes", C.c_int64),
                ("workspace_bytes", C.c_int64), ("ppr_states", C.c_int32), ("fp8_unavailable", C.c_int32),
                ("last_ppr_state", C.c_int32), ("reserved", C.c_int32), ("calls_score_facts", C.c_int64),
                ("calls_retrieve", C.c_int64), ("calls_dense_retrieve", C.c_int64), ("calls_ppr", C.c_int64),
                ("calls_shard", C.c_int64), ("queries", C.c_int64)]


COMM_ALL_REDUCE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)
COMM_ALL_GATHER = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
COMM_EXCHANGE_BEGIN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)
COMM_EXCHANGE_WAIT = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_void_p)


class Comm(C.Structure):
    """hrag_comm of include/hrag.h: the collectives a host supplies to hrag_shard_score_facts_all / hrag_shard_retrieve."""
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int32), ("world", C.c_int32), ("all_reduce", COMM_ALL_REDUCE),
                ("all_gather", COMM_ALL_GATHER), ("exchange_begin", COMM_EXCHANGE_BEGIN), ("exchange_wait", COMM_EXCHANGE_WAIT)]


PPR_STATE_F32, PPR_STATE_F16, PPR_STATE_SMALL, PPR_STATE_FP8 = 1, 2, 4, 8
FP8_UNAVAILABLE = {1: "hrag_graph_desc.col_sum was not given", 2: "V + 1 > 2^24 vertices",
                   4: "the passage shard is not aligned with the row shard", 8: "disabled by HRAG_OPT_F32_STATE / HRAG_OPT_NO_FP8",
                   16: "max_batch <= 64 (never needed)"}

_P = C.c_void_p
_I32, _I64, _F32 = C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol include/hrag.h declares
SIGNATURES = {
    "hrag_last_error": (C.c_char_p, []),
    "hrag_version": (C.c_int, []),
    "hrag_engine_create": (C.c_int, [C.POINTER(GraphDesc), C.POINTER(EmbedDesc), C.POINTER(EmbedDesc),
                                     C.POINTER(FactDesc), C.POINTER(Opts), C.POINTER(_P)]),
    "hrag_engine_destroy": (C.c_int, [_P]),
    "hrag_workspace_create": (C.c_int, [_P, C.POINTER(_P)]),
    "hrag_engine_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "hrag_score_facts": (C.c_int, [_P, _P, _I32, _I32, _P, _P, _P]),
    "hrag_retrieve": (C.c_int, [_P, _P, _I32, _P, _P, _P, _I32, _I32, _F32, _F32, _I32, _I32, _F32, _I32, _P, _P, _P,
                                _P, _P, _P]),
    "hrag_retrieve_scored": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _I32, _I32, _F32, _F32, _I32, _I32, _F32, _I32, _P, _P,
                                       _P, _P, _P, _P]),
    "hrag_last_doc_scores": (C.c_int, [_P, _I32, _P, _I64, _P]),
    "hrag_dense_retrieve": (C.c_int, [_P, _P, _I32, _I32, _P, _P, _P]),
    "hrag_sim_scores": (C.c_int, [_P, _I32, _P, _I32, _P, _P]),
    "hrag_ppr": (C.c_int, [_P, _P, _I32, _F32, _I32, _P, _P, _P]),
    "hrag_topk_rows": (C.c_int, [_P, _I32, _I64, _I64, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "hrag_ppr_sweeps": (C.c_int, [_P, _I32, _I32, _F32, _I32, _P]),
    "hrag_ppr_layout": (C.c_int, [_P, _I32, C.POINTER(_I32), C.POINTER(_I32)]),
    "hrag_row_minmax": (C.c_int, [_P, _I32, _I64, _I64, _P, _P, _P]),
    "hrag_stage_seeds": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "hrag_stage_teleport": (C.c_int, [_P, _P, _I64, _P, _P, _F32, _P, _I32, _P, _P]),
    "hrag_stage_ppr_init": (C.c_int, [_P, _P, _P, _P, _P, _I32, _P, _P]),
    "hrag_stage_ppr_step": (C.c_int, [_P, _P, _P, _P, _P, _I32, _F32, _P, _P, _P]),
    "hrag_stage_colsum": (C.c_int, [_P, _P, _I32, _P, _P, _P]),
    "hrag_colsum_workspace_bytes": (C.c_int64, [_P, _I32]),
    "hrag_stage_doc_scores": (C.c_int, [_P, _P, _P, _I32, _P, _I64, _P, _P, _P, _P, _I64, _P]),
    "hrag_normalize_split_bf16": (C.c_int, [_P, _I64, _I32, _I32, _P, _P, _P]),
    "hrag_sim_gemm": (C.c_int, [_P, _I64, _I32, _P, _I32, _P, _I64, _I32, _I32, _P]),
    "hrag_sim_topk_workspace_bytes": (C.c_int64, [_I64, _I32]),
    "hrag_sim_topk": (C.c_int, [_P, _I64, _I32, _P, _I32, _I32, _I32, _P, _I64, _P, _P, _P]),
    "hrag_sim_topk_min_score": (C.c_int, [_P, _I64, _I32, _P, _I32, _I32, _I32, _I32, C.c_float, C.c_float, _P, _I64, _P, _P, _P, _P]),
    "hrag_split_f32": (C.c_int, [_P, _I64, _I32, _I32, _I32, _P, _P]),
    "hrag_shard_layout_query": (C.c_int, [_P, _I32, _I32, C.POINTER(ShardLayout)]),
    "hrag_shard_score_facts": (C.c_int, [_P, _P, _I32, _I32, _P, _P, _P, _P, _P]),
    "hrag_shard_passage_scores": (C.c_int, [_P, _P, _I32, _P, _P, _P]),
    "hrag_shard_prior_stats": (C.c_int, [_P, _P, _P, _F32, _P, _I32, _P, _P, _P]),
    "hrag_shard_ppr_begin": (C.c_int, [_P, _P, _P, _P, _P, _F32, _P, _P, _P, _P, _I32, _F32, _I32, _I32, _F32, _I32, _P, _P,
                                       _P, C.POINTER(_I32), _P]),
    "hrag_shard_ppr_sweep": (C.c_int, [_P, _I32, _I32, C.POINTER(_I32), C.POINTER(_I32), _P]),
    "hrag_shard_ppr_est": (C.c_int, [_P, _P, _I32, _P]),
    "hrag_shard_ppr_decide": (C.c_int, [_P, _I32, _P]),
    "hrag_shard_ppr_gate": (C.c_int, [_P, _I32, C.POINTER(_I32), _P]),
    "hrag_shard_finish": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P]),
    "hrag_shard_workspace_bytes": (C.c_int64, [_P, _I32, _I32, _I32]),
    "hrag_shard_score_facts_all": (C.c_int, [_P, C.POINTER(Comm), _P, _I32, _I32, _P, _I64, _P, _P, _P]),
    "hrag_shard_retrieve": (C.c_int, [_P, C.POINTER(Comm), _P, _I32, _P, _P, _P, _I32, _I32, _F32, _F32, _I32, _I32, _F32, _I32,
                                      _I32, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P]),
    "hrag_engine_set_flags": (C.c_int, [_P, _I32, _I32]),
    "hrag_engine_gather_embeddings": (C.c_int, [_P, _I32, _P, _I64, _P, _P, _P]),
    "hrag_get_timings": (C.c_int, [_P, C.POINTER(Timings)]),
    "hrag_set_profiling": (C.c_int, [_P, _I32]),
}

_lib = None


def build_library(force: bool = False) -> str:
    from .csrc.build import build
    return build(force=force)


def load(build_if_missing: bool = True):
    """Load libhrag.so (building it with hipcc first if it is not there)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It must
    # be in the process BEFORE libhrag.so so that libhrag's NEEDED libamdhip64.so.7 binds to the same
    # runtime instance torch uses; loading libhrag first pulls in /opt/rocm's copy as a second
    # runtime, which then sees "no ROCm-capable device" (observed on the GPU box).
    import torch  # noqa: F401
    if build_if_missing:
        # digest-stamped: a no-op when neither a source nor a header changed, so a stale library can never
        # be loaded against newer ctypes struct layouts.  Without hipcc (a deployment box that received the
        # built library) the existing file is used and the version check below is the guard.
        # A compile error propagates (never load a stale library over a source that no longer builds).
        from .csrc.build import ToolchainMissing
        try:
            build_library()
        except ToolchainMissing:
            if not os.path.exists(LIB_PATH):
                raise
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing; run `python -m hipporag_amd.csrc.build`")
    # RTLD_LAZY: the CPU-only container has no HIP driver; symbols resolve at first use on a GPU box
    lib = C.CDLL(LIB_PATH, mode=os.RTLD_LAZY if hasattr(os, "RTLD_LAZY") else C.DEFAULT_MODE)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => the library does not export what hrag.h declares
        fn.restype = res
        fn.argtypes = args
    if lib.hrag_version() != HRAG_VERSION:
        raise ImportError(f"{LIB_PATH} reports version {lib.hrag_version()}, this binding expects {HRAG_VERSION}: "
                          "rebuild with `python -m hipporag_amd.csrc.build --force`")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != HRAG_OK:
        msg = load().hrag_last_error()
        raise HragError(status, msg.decode() if msg else "")
