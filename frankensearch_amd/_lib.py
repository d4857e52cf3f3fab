"""ctypes binding of libfsgpu.so (the C ABI declared in include/fsgpu.h).

There is NO fallback: if the HIP library is missing this module raises at import of the symbol
table, and every compute entry point fails with FSGPU_ERR_NO_DEVICE when no GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfsgpu.so")

OK = 0
ERR_DIMENSION_MISMATCH = 1
ERR_INVALID_CONFIG = 2
ERR_INDEX_CORRUPTED = 3
ERR_INDEX_VERSION_MISMATCH = 4
ERR_IO = 5
ERR_DEVICE = 6
ERR_NO_DEVICE = 7
ERR_NULL_ARGUMENT = 8
ERR_EMBEDDING_FAILED = 9
ERR_MODEL_LOAD_FAILED = 10

ZERO_SIGNAL_NONE = 0
ZERO_SIGNAL_CALLER_REQUESTED_ZERO_K = 1
ZERO_SIGNAL_ZERO_NORM_QUERY = 2
ZERO_SIGNAL_FILTER_ELIMINATED_ALL = 3
ZERO_SIGNAL_NEWLY_CREATED_EMPTY = 4
ZERO_SIGNAL_ALL_TOMBSTONED = 5
ZERO_SIGNAL_WAL_ONLY_NO_LIVE_RECORDS = 6
ZERO_SIGNAL_NO_USABLE_VECTORS = 7

HREDUCE_SSE2 = 0
HREDUCE_AVX = 1
HREDUCE_SEQ = 2

_vp, _u32, _u64, _i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32

# name -> (restype, argtypes); must list every symbol include/fsgpu.h declares
SIGNATURES = {
    "fsgpu_version": (C.c_char_p, []),
    "fsgpu_device_count": (_i32, []),
    "fsgpu_last_error": (C.c_char_p, []),
    "fsgpu_last_main_pass_kernel": (C.c_char_p, []),
    "fsgpu_index_create": (_i32, [_i32, _u32, _u64, _vp, _vp, _u64, C.POINTER(_vp)]),
    "fsgpu_index_create_device": (_i32, [_i32, _u32, _u64, _vp, _vp, _u64, C.POINTER(_vp)]),
    "fsgpu_index_open_fsvi": (_i32, [C.c_char_p, _i32, C.POINTER(_vp)]),
    "fsgpu_index_destroy": (None, [_vp]),
    "fsgpu_index_record_count": (_u64, [_vp]),
    "fsgpu_index_dimension": (_u32, [_vp]),
    "fsgpu_index_set_hreduce": (_i32, [_vp, _i32]),
    "fsgpu_index_set_batched_filter": (_i32, [_vp, _i32]),
    "fsgpu_index_set_filter_rotation": (_i32, [_vp, _i32]),
    "fsgpu_index_filter_rotated": (_i32, [_vp]),
    "fsgpu_index_set_int8_latency": (_i32, [_vp, _i32]),
    "fsgpu_index_batched_filter_stats": (_i32, [_vp, _vp, _vp, _vp]),
    "fsgpu_index_int8_filter_bound": (_i32, [_vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp]),
    "fsgpu_index_doc_id": (_i32, [_vp, _u32, C.POINTER(_vp), C.POINTER(_u32)]),
    "fsgpu_index_soft_delete": (_i32, [_vp, C.c_char_p, _u32, C.POINTER(_i32)]),
    "fsgpu_index_wal_append": (_i32, [_vp, C.c_char_p, _u32, _vp, _u32]),
    "fsgpu_index_wal_record_count": (_u64, [_vp]),
    "fsgpu_index_set_live_bitmap": (_i32, [_vp, _vp]),
    "fsgpu_search_topk": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "fsgpu_search_topk_exact": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "fsgpu_allow_bitmap_create": (_i32, [_vp, _vp, C.POINTER(_vp)]),
    "fsgpu_allow_bitmap_destroy": (None, [_vp]),
    "fsgpu_allow_bitmap_allowed_rows": (_u64, [_vp]),
    "fsgpu_search_topk_filtered": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "fsgpu_search_topk_batched_filtered": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_search_topk_device": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp]),
    "fsgpu_search_topk_batched": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_search_topk_batched_device": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_search_topk_batched_packed_device": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_search_topk_batched_device_begin": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i32)]),
    "fsgpu_search_topk_batched_device_end": (_i32, [_vp, _i32, C.POINTER(_u32)]),
    "fsgpu_search_topk_batched_device_end_late": (_i32, [_vp, _i32, C.POINTER(_u32), C.POINTER(_u32)]),
    "fsgpu_search_topk_packed_device": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "fsgpu_merge_topk_device": (_i32, [_i32, _vp, _u32, _u32, _u32, _u64, _u64, _u32, _vp, _vp, _vp, _vp]),
    "fsgpu_sharded_create": (_i32, [_vp, _u32, _u32, _u64, _vp, _vp, _i32, C.POINTER(_vp)]),
    "fsgpu_sharded_create_device": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _i32, C.POINTER(_vp)]),
    "fsgpu_sharded_create_grouped": (_i32, [_vp, _u32, _u32, _u32, _u64, _vp, _vp, _i32, C.POINTER(_vp)]),
    "fsgpu_sharded_create_device_grouped": (_i32, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _i32, C.POINTER(_vp)]),
    "fsgpu_sharded_open_fsvi_grouped": (_i32, [C.c_char_p, _vp, _u32, _u32, _i32, C.POINTER(_vp)]),
    "fsgpu_sharded_query_groups": (_u32, [_vp]),
    "fsgpu_sharded_row_shards": (_u32, [_vp]),
    "fsgpu_sharded_set_int8_latency": (_i32, [_vp, _i32]),
    "fsgpu_sharded_search_parts": (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "fsgpu_sharded_destroy": (None, [_vp]),
    "fsgpu_sharded_record_count": (_u64, [_vp]),
    "fsgpu_sharded_dimension": (_u32, [_vp]),
    "fsgpu_sharded_shard_count": (_u32, [_vp]),
    "fsgpu_sharded_exchange_mode": (_i32, [_vp]),
    "fsgpu_sharded_device": (_i32, [_vp, _u32]),
    "fsgpu_sharded_shard_range": (_i32, [_vp, _u32, C.POINTER(_u64), C.POINTER(_u64)]),
    "fsgpu_sharded_set_hreduce": (_i32, [_vp, _i32]),
    "fsgpu_sharded_search_topk": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "fsgpu_sharded_search_topk_batched": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_sharded_search": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "fsgpu_sharded_search_begin": (_i32, [_vp, _vp, C.POINTER(_u64)]),
    "fsgpu_sharded_search_end": (_i32, [_vp, _u64, _vp, _vp, _vp, _vp]),
    "fsgpu_sharded_quant_scale_max": (C.c_float, [_vp]),
    "fsgpu_sharded_set_coalescing": (_i32, [_vp, _u32, _u32]),
    "fsgpu_sharded_coalescing_stats": (_i32, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "fsgpu_sharded_alignment_create": (_i32, [_vp, _vp, C.POINTER(_vp)]),
    "fsgpu_sharded_quality_scores_for_hits": (_i32, [_vp, _vp, _vp, _vp, _u32, _vp, _u32, _vp, _vp]),
    "fsgpu_sharded_open_fsvi": (_i32, [C.c_char_p, _vp, _u32, _i32, C.POINTER(_vp)]),
    "fsgpu_sharded_set_live_bitmap": (_i32, [_vp, _vp]),
    "fsgpu_sharded_soft_delete": (_i32, [_vp, C.c_char_p, _u32, C.POINTER(_i32)]),
    "fsgpu_sharded_wal_append": (_i32, [_vp, C.c_char_p, _u32, _vp, _u32]),
    "fsgpu_sharded_wal_record_count": (_u64, [_vp]),
    "fsgpu_sharded_doc_id": (_i32, [_vp, _u32, C.POINTER(_vp), C.POINTER(_u32)]),
    "fsgpu_sharded_search_hits": (_i32, [_vp, _vp, _u32, _u32, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_sharded_gather_dot": (_i32, [_vp, _vp, _u32, _vp, _u32, _vp]),
    "fsgpu_search_topk_classified": (_i32, [_vp, _vp, _u32, _u32, _vp, _vp, C.POINTER(_u32), C.POINTER(_i32)]),
    "fsgpu_search_topk_int8_two_pass": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_search_hits": (_i32, [_vp, _vp, _u32, _u32, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_gather_dot": (_i32, [_vp, _vp, _u32, _vp, _u32, _vp]),
    "fsgpu_lab_sort_keys_desc": (_i32, [_i32, _vp, _u64, _u64, _vp]),
    "fsgpu_bench_fixture_device": (_i32, [_i32, _u64, _u64, _u32, _u32, C.c_float, _u64, _i32, _vp, _vp]),
    "fsgpu_encode_f32_to_f16": (_i32, [_i32, _vp, _u64, _vp]),
    "fsgpu_widen_f16_to_f32": (_i32, [_i32, _vp, _u64, _vp]),
    "fsgpu_m2v_create": (_i32, [_i32, _vp, _u32, _u32, C.POINTER(_vp)]),
    "fsgpu_m2v_destroy": (None, [_vp]),
    "fsgpu_m2v_dimension": (_u32, [_vp]),
    "fsgpu_m2v_embed": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "fsgpu_bert_create": (_i32, [_i32, _vp, _vp, C.POINTER(_vp)]),
    "fsgpu_bert_create_safetensors": (_i32, [_i32, _vp, _u64, C.c_float, C.POINTER(_vp)]),
    "fsgpu_bert_destroy": (None, [_vp]),
    "fsgpu_bert_embed_device": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "fsgpu_m2v_embed_device": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "fsgpu_search_topk_batched_device_queries": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_search_topk_int8_two_pass_batched_device_queries": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_device_malloc": (_i32, [_i32, _u64, C.POINTER(_vp)]),
    "fsgpu_device_free": (_i32, [_i32, _vp]),
    "fsgpu_bert_device": (_i32, [_vp]),
    "fsgpu_m2v_device": (_i32, [_vp]),
    "fsgpu_index_device": (_i32, [_vp]),
    "fsgpu_bert_dimension": (_u32, [_vp]),
    "fsgpu_bert_embed": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "fsgpu_rrf_fuse": (_i32, [_vp, _u32, _vp, _u32, C.c_double, C.c_double, C.c_double, _i32, _u32, _u32, _vp,
                              C.POINTER(_u32)]),
    "fsgpu_blend_two_tier": (_i32, [_vp, _u32, _vp, _u32, C.c_float, _vp, C.POINTER(_u32)]),
    "fsgpu_blend_two_tier_aligned": (_i32, [_vp, _u32, _vp, _vp, C.c_float, _vp, C.POINTER(_u32)]),
    "fsgpu_alignment_create": (_i32, [_vp, _vp, C.POINTER(_vp)]),
    "fsgpu_alignment_destroy": (None, [_vp]),
    "fsgpu_alignment_kind": (_i32, [_vp]),
    "fsgpu_alignment_quality_row": (C.c_int64, [_vp, _u64]),
    "fsgpu_alignment_unmatched_quality_docs": (_u64, [_vp]),
    "fsgpu_quality_scores_for_hits": (_i32, [_vp, _vp, _vp, _vp, _u32, _vp, _u32, _vp, _vp]),
    "fsgpu_quality_scores_for_hits_batched": (_i32, [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "fsgpu_index_set_after_enqueue_hook": (_i32, [_vp, _vp, _vp]),
    "fsgpu_index_set_profiling": (_i32, [_vp, _i32]),
    "fsgpu_index_scan_time": (_i32, [_vp, C.POINTER(C.c_double), C.POINTER(_u64), _i32]),
    "fsgpu_search_topk_int8_two_pass_batched": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_search_topk_4bit_two_pass_batched": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_search_topk_4bit_two_pass": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_fsvi_write": (_i32, [C.c_char_p, C.c_char_p, C.c_char_p, _u32, _u64, _vp, _vp, _vp, C.c_uint8, _i32]),
    "fsgpu_fsvi_write_quant": (_i32, [C.c_char_p, C.c_char_p, C.c_char_p, _u32, _u64, _vp, _vp, _vp, C.c_uint8, _i32,
                               C.c_uint8]),
    "fsgpu_search_mrl_batched": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(_u32)]),
    "fsgpu_search_mrl": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, C.POINTER(_u32), _vp]),
    "fsgpu_index_set_coalescing": (_i32, [_vp, _u32, _u32]),
    "fsgpu_index_coalescing_stats": (_i32, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "fsgpu_m2v_set_coalescing": (_i32, [_vp, _u32, _u32]),
    "fsgpu_bert_set_coalescing": (_i32, [_vp, _u32, _u32]),
    "fsgpu_index_scan_stats": (_i32, [_vp, C.POINTER(C.c_double), C.POINTER(_u64), C.POINTER(_u64), _i32]),
    "fsgpu_index_allow_bitmap_for_hashes": (_i32, [_vp, C.POINTER(_u64), _u32, C.POINTER(_u64), C.POINTER(_u64)]),
    "fsgpu_index_filter_stats": (_i32, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "fsgpu_index_set_variant": (_i32, [_vp, _i32]),
}

_lib = None


def _share_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own copies of the ROCm runtime libraries (libamdhip64, librccl, hsa-runtime, ...) under
    the SONAMEs libfsgpu.so also resolves.  Whichever copy of a library is loaded first serves the whole process: with the system
    HIP runtime first, a later `import torch` finds no GPU; with only SOME of the system copies first (libfsgpu loaded, then torch
    loading the rest of its own set) the process has been seen to abort in the libraries' exit handlers ("double free or
    corruption" after `pytest tests/test_gpu_sharded.py` alone, round 5 — the whole suite, where an earlier test file imports
    torch first, was fine).  When torch is installed, import it BEFORE libfsgpu.so is loaded, so that the order of imports in the
    caller does not matter (bench.py / sharded.py hand torch device pointers to this library anyway).  Loading RCCL alone ahead of
    time is not an option: without a GPU its own exit handlers abort."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    try:
        import torch  # noqa: F401
        return
    except Exception:   # a broken torch install: fall back to its HIP runtime alone
        pass
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the loader's own resolution


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m frankensearch_amd.build` "
                "(the HIP library is the product; there is no CPU fallback)")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI and the header drift apart
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return (lib().fsgpu_last_error() or b"").decode("utf-8", "replace")
