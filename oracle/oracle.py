"""ctypes binding of the CPU oracle (oracle/fs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the
``cpu_baseline`` leg of bench.py -- never by frankensearch_amd/ (the product fails loudly
when libfsgpu.so is missing instead of falling back to this).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libfsoracle.so")

HREDUCE_SSE2 = 0
HREDUCE_AVX = 1
HREDUCE_SEQ = 2

OK = 0
ERR_DIMENSION_MISMATCH = 1
ERR_INVALID_CONFIG = 2
ERR_INDEX_CORRUPTED = 3
ERR_INDEX_VERSION_MISMATCH = 4
ERR_IO = 5

PARALLEL_THRESHOLD = 10_000  # search.rs:23
PARALLEL_CHUNK_SIZE = 1_024  # search.rs:25


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (idempotent)."""
    srcs = [os.path.join(_HERE, f) for f in ("fs_oracle.c", "fs_oracle_avx2.c", "bert_oracle_c.c", "fs_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p, f32p, u16p, u32p, u64p = (C.POINTER(t) for t in (C.c_uint8, C.c_float, C.c_uint16, C.c_uint32, C.c_uint64))
        L.fso_f16_to_f32.restype = C.c_float
        L.fso_f16_to_f32.argtypes = [C.c_uint16]
        L.fso_f32_to_f16.restype = C.c_uint16
        L.fso_f32_to_f16.argtypes = [C.c_float]
        L.fso_encode_f32_to_f16.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        for name in ("fso_dot_f16_f32", "fso_dot_f16_f32_fast"):
            fn = getattr(L, name)
            fn.restype = C.c_float
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.fso_has_avx2_f16c.restype = C.c_int
        L.fso_ranks_before.restype = C.c_int
        L.fso_ranks_before.argtypes = [C.c_uint64, C.c_float, C.c_uint64, C.c_float]
        L.fso_search_top_k.restype = C.c_size_t
        L.fso_search_top_k.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fso_classify_query.restype = C.c_int
        L.fso_classify_query.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_size_t, C.POINTER(C.c_int)]
        L.fso_gather_dot.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.fso_fnv1a64.restype = C.c_uint64
        L.fso_fnv1a64.argtypes = [C.c_char_p, C.c_size_t]
        L.fso_crc32.restype = C.c_uint32
        L.fso_crc32.argtypes = [C.c_char_p, C.c_size_t]
        L.fso_align_up.restype = C.c_uint64
        L.fso_align_up.argtypes = [C.c_uint64, C.c_uint64]
        L.fso_vector_signal_usable.restype = C.c_int
        L.fso_vector_signal_usable.argtypes = [C.c_void_p, C.c_size_t]
        L.fso_l2_normalize.argtypes = [C.c_void_p, C.c_size_t]
        L.fso_fsvi_write.restype = C.c_int
        L.fso_fsvi_write.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64,
                                     C.POINTER(C.c_char_p), C.c_void_p, C.c_uint8]
        L.fso_fsvi_open.restype = C.c_int
        L.fso_fsvi_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.fso_fsvi_close.argtypes = [C.c_void_p]
        L.fso_fsvi_record_count.restype = C.c_uint64
        L.fso_fsvi_record_count.argtypes = [C.c_void_p]
        L.fso_fsvi_dimension.restype = C.c_uint32
        L.fso_fsvi_dimension.argtypes = [C.c_void_p]
        L.fso_fsvi_vectors_offset.restype = C.c_uint64
        L.fso_fsvi_vectors_offset.argtypes = [C.c_void_p]
        L.fso_fsvi_slab.restype = C.c_void_p
        L.fso_fsvi_slab.argtypes = [C.c_void_p]
        L.fso_fsvi_doc_id.restype = C.c_uint32
        L.fso_fsvi_doc_id.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        L.fso_fsvi_flags.restype = C.c_uint16
        L.fso_fsvi_flags.argtypes = [C.c_void_p, C.c_uint64]
        L.fso_fsvi_set_flags.argtypes = [C.c_void_p, C.c_uint64, C.c_uint16]
        L.fso_fsvi_search.restype = C.c_size_t
        L.fso_fsvi_search.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.fso_dot_f32_f32.restype = C.c_float
        L.fso_dot_f32_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.fso_fsvi_soft_delete.restype = C.c_size_t
        L.fso_fsvi_soft_delete.argtypes = [C.c_void_p, C.c_char_p]
        L.fso_fsvi_append.restype = C.c_int
        L.fso_fsvi_append.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
        L.fso_fsvi_wal_count.restype = C.c_uint64
        L.fso_fsvi_wal_count.argtypes = [C.c_void_p]
        L.fso_fsvi_wal_doc_id.restype = C.c_uint32
        L.fso_fsvi_wal_doc_id.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        L.fso_quantize_slab_i8.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.fso_quantize_query_i8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.fso_dot_i8_i8.restype = C.c_int32
        L.fso_dot_i8_i8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.fso_search_int8_two_pass.restype = C.c_size_t
        L.fso_search_int8_two_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                               C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.fso_fixture_hashmix.restype = C.c_float
        L.fso_fixture_hashmix.argtypes = [C.c_uint64, C.c_uint64]
        L.fso_raw_vector.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
        L.fso_normalize_bench.argtypes = [C.c_void_p, C.c_uint32]
        L.fso_clustered_corpus_f16.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
        L.fso_clustered_query.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
        L.fso_m2v_embed.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def _p(a: np.ndarray) -> int:
    return a.ctypes.data


# ---- f16 ----
def f16_to_f32(h: int) -> float:
    return lib().fso_f16_to_f32(h)


def f32_to_f16(f: float) -> int:
    return lib().fso_f32_to_f16(f)


def encode_f32_to_f16(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.float32)
    dst = np.empty(src.shape, dtype=np.uint16)
    lib().fso_encode_f32_to_f16(_p(src), src.size, _p(dst))
    return dst


# ---- dot / search ----
def dot_f16_f32(row_u16: np.ndarray, q: np.ndarray, hreduce: int = HREDUCE_SSE2, fast: bool = False) -> float:
    row = np.ascontiguousarray(row_u16, dtype=np.uint16)
    q = np.ascontiguousarray(q, dtype=np.float32)
    assert row.size == q.size
    fn = lib().fso_dot_f16_f32_fast if fast else lib().fso_dot_f16_f32
    return fn(_p(row), _p(q), q.size, hreduce)


def dot_f32_f32(a: np.ndarray, b: np.ndarray, hreduce: int = HREDUCE_SSE2) -> float:
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.size == b.size
    return lib().fso_dot_f32_f32(_p(a), _p(b), a.size, hreduce)


def live_bitmap(live_bool: np.ndarray) -> np.ndarray:
    """bool[N] -> uint64 bitmap (bit r set = row r live)."""
    n = live_bool.size
    words = (n + 63) // 64
    padded = np.zeros(words * 64, dtype=np.uint8)
    padded[:n] = live_bool.astype(np.uint8)
    return np.packbits(padded, bitorder="little").view(np.uint64).copy()


def search_top_k(slab_u16: np.ndarray, q: np.ndarray, k: int, live: np.ndarray | None = None,
                 parallel_threshold: int = PARALLEL_THRESHOLD, chunk_size: int = PARALLEL_CHUNK_SIZE,
                 parallel_enabled: bool = True, nthreads: int = 1, hreduce: int = HREDUCE_SSE2):
    """Returns (rows uint32[count], scores float32[count]) best-first; slab_u16 is [N, dim] uint16."""
    slab = np.ascontiguousarray(slab_u16, dtype=np.uint16)
    n, dim = slab.shape
    q = np.ascontiguousarray(q, dtype=np.float32)
    if q.size != dim:
        raise ValueError(f"DimensionMismatch expected={dim} found={q.size}")
    cap = max(1, min(k, n))
    rows = np.empty(cap, dtype=np.uint32)
    scores = np.empty(cap, dtype=np.float32)
    bm = None
    if live is not None:
        bm = live_bitmap(np.asarray(live, dtype=bool)) if live.dtype != np.uint64 else live
    cnt = lib().fso_search_top_k(_p(slab), n, dim, _p(bm) if bm is not None else None, _p(q), k,
                                 parallel_threshold, chunk_size, int(parallel_enabled), nthreads, hreduce,
                                 _p(rows), _p(scores))
    return rows[:cnt].copy(), scores[:cnt].copy()


def pack_slab_4bit(slab_u16: np.ndarray) -> np.ndarray:
    """pack_f16_le_bytes_to_4bit (simd.rs:2153-2215): [N, ceil(dim/2)] uint8."""
    slab = np.ascontiguousarray(slab_u16, dtype=np.uint16)
    n, dim = slab.shape
    out = np.empty((n, (dim + 1) // 2), dtype=np.uint8)
    L = lib()
    L.fso_pack_slab_4bit.restype = None
    L.fso_pack_slab_4bit.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    L.fso_pack_slab_4bit(slab.ctypes.data, n, dim, out.ctypes.data)
    return out


def pack_query_4bit(q: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.float32)
    out = np.empty((q.size + 1) // 2, dtype=np.uint8)
    L = lib()
    L.fso_pack_query_4bit.restype = None
    L.fso_pack_query_4bit.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.fso_pack_query_4bit(q.ctypes.data, q.size, out.ctypes.data)
    return out


def dot_4bit(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    L = lib()
    L.fso_dot_4bit.restype = C.c_int32
    L.fso_dot_4bit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    return int(L.fso_dot_4bit(a.ctypes.data, b.ctypes.data, a.size))


def search_4bit_two_pass(slab_u16: np.ndarray, q: np.ndarray, k: int, candidate_multiplier: int,
                         live: np.ndarray | None = None, hreduce: int = HREDUCE_SSE2):
    """search_top_k_4bit_two_pass (search.rs:876-946) on a raw slab -> (rows, scores)."""
    slab = np.ascontiguousarray(slab_u16, dtype=np.uint16)
    n, dim = slab.shape
    q = np.ascontiguousarray(q, dtype=np.float32)
    if q.size != dim:
        raise ValueError(f"DimensionMismatch expected={dim} found={q.size}")
    nib = pack_slab_4bit(slab)
    cap = max(1, min(k, n))
    rows = np.empty(cap, dtype=np.uint32)
    scores = np.empty(cap, dtype=np.float32)
    bm = None
    if live is not None:
        bm = live_bitmap(np.asarray(live, dtype=bool)) if live.dtype != np.uint64 else live
    L = lib()
    L.fso_search_4bit_two_pass.restype = C.c_size_t
    L.fso_search_4bit_two_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    cnt = L.fso_search_4bit_two_pass(slab.ctypes.data, nib.ctypes.data, n, dim, bm.ctypes.data if bm is not None else None,
                                     q.ctypes.data, k, candidate_multiplier, hreduce, rows.ctypes.data, scores.ctypes.data)
    return rows[:cnt].copy(), scores[:cnt].copy()


def mrl_search(slab_u16: np.ndarray, q: np.ndarray, limit: int, search_dims: int, rescore_dims: int = 0,
               rescore_top_k: int = 0, live: np.ndarray | None = None, wal: list | None = None,
               hreduce: int = HREDUCE_SSE2):
    """VectorIndex::mrl_search (mrl.rs:241-395): truncated scan + rescore.  `wal` = resident f32 embeddings; their hits
    come back at the virtual index N + i.  Falls back to search_top_k when search_dims >= dim (mrl.rs:283-296) — only
    valid here without WAL entries.  Returns (rows uint32[count], scores float32[count])."""
    slab = np.ascontiguousarray(slab_u16, dtype=np.uint16)
    n, dim = slab.shape
    q = np.ascontiguousarray(q, dtype=np.float32)
    if q.size != dim:
        raise ValueError(f"DimensionMismatch expected={dim} found={q.size}")
    if search_dims == 0:
        raise ValueError("InvalidConfig search_dims must be at least 1")
    if search_dims >= dim:
        assert not wal
        return search_top_k(slab, q, limit, live=live, hreduce=hreduce)
    wal = [np.ascontiguousarray(w, dtype=np.float32) for w in (wal or [])]
    cap = max(1, limit)
    rows = np.empty(cap, dtype=np.uint32)
    scores = np.empty(cap, dtype=np.float32)
    bm = None
    if live is not None:
        bm = live_bitmap(np.asarray(live, dtype=bool)) if live.dtype != np.uint64 else live
    ptrs = (C.c_void_p * max(len(wal), 1))(*[w.ctypes.data for w in wal])
    L = lib()
    L.fso_mrl_search.restype = C.c_size_t
    L.fso_mrl_search.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                 C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    cnt = L.fso_mrl_search(slab.ctypes.data, n, dim, bm.ctypes.data if bm is not None else None, ptrs, len(wal),
                           q.ctypes.data, limit, search_dims, rescore_dims, rescore_top_k, hreduce, rows.ctypes.data,
                           scores.ctypes.data)
    return rows[:cnt].copy(), scores[:cnt].copy()


def quantize_slab_i8(slab_u16: np.ndarray) -> np.ndarray:
    slab = np.ascontiguousarray(slab_u16, dtype=np.uint16)
    out = np.empty(slab.shape, dtype=np.int8)
    lib().fso_quantize_slab_i8(_p(slab), slab.size, _p(out))
    return out


def quantize_query_i8(q: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.float32)
    out = np.empty(q.size, dtype=np.int8)
    lib().fso_quantize_query_i8(_p(q), q.size, _p(out))
    return out


def dot_i8_i8(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, dtype=np.int8)
    b = np.ascontiguousarray(b, dtype=np.int8)
    return lib().fso_dot_i8_i8(_p(a), _p(b), a.size)


def search_int8_two_pass(slab_u16: np.ndarray, q: np.ndarray, k: int, candidate_multiplier: int = 3,
                         live: np.ndarray | None = None, slab_i8: np.ndarray | None = None,
                         hreduce: int = HREDUCE_SSE2):
    """VectorIndex::search_top_k_int8_two_pass (search.rs:514-661) on a raw slab (no WAL, no doc-id dedup)."""
    slab = np.ascontiguousarray(slab_u16, dtype=np.uint16)
    n, dim = slab.shape
    q = np.ascontiguousarray(q, dtype=np.float32)
    if slab_i8 is None:
        slab_i8 = quantize_slab_i8(slab)
    cap = max(1, min(k, n))
    rows = np.empty(cap, dtype=np.uint32)
    scores = np.empty(cap, dtype=np.float32)
    bm = live_bitmap(np.asarray(live, dtype=bool)) if live is not None else None
    cnt = lib().fso_search_int8_two_pass(_p(slab), _p(slab_i8), n, dim, _p(bm) if bm is not None else None, _p(q), k,
                                         candidate_multiplier, hreduce, _p(rows), _p(scores))
    return rows[:cnt].copy(), scores[:cnt].copy()


def classify_query(q: np.ndarray, dim: int, k: int):
    q = np.ascontiguousarray(q, dtype=np.float32)
    z = C.c_int(0)
    st = lib().fso_classify_query(_p(q), q.size, dim, k, C.byref(z))
    return st, z.value


def gather_dot(slab_u16: np.ndarray, q: np.ndarray, rows: np.ndarray, hreduce: int = HREDUCE_SSE2) -> np.ndarray:
    slab = np.ascontiguousarray(slab_u16, dtype=np.uint16)
    q = np.ascontiguousarray(q, dtype=np.float32)
    rows = np.ascontiguousarray(rows, dtype=np.uint32)
    out = np.empty(rows.size, dtype=np.float32)
    lib().fso_gather_dot(_p(slab), slab.shape[1], _p(q), _p(rows), rows.size, hreduce, _p(out))
    return out


# ---- hashes ----
def fnv1a64(b: bytes) -> int:
    return lib().fso_fnv1a64(b, len(b))


def crc32(b: bytes) -> int:
    return lib().fso_crc32(b, len(b))


def align_up(v: int, a: int) -> int:
    return lib().fso_align_up(v, a)


def vector_signal_usable(v: np.ndarray) -> bool:
    v = np.ascontiguousarray(v, dtype=np.float32)
    return bool(lib().fso_vector_signal_usable(_p(v), v.size))


def l2_normalize(v: np.ndarray) -> np.ndarray:
    v = np.array(v, dtype=np.float32, copy=True)
    lib().fso_l2_normalize(_p(v), v.size)
    return v


# ---- FSVI ----
def fsvi_write(path: str, rows, embedder_id: str = "hash", revision: str = "test", compaction_gen: int = 1,
               quantization: int = 1) -> int:
    """rows: list of (doc_id, vector) like the reference test helper write_index (search.rs:1784-1799);
    quantization 1 = F16 (default), 0 = F32."""
    n = len(rows)
    dim = len(rows[0][1]) if n else 0
    if n == 0:
        return ERR_INVALID_CONFIG
    ids = (C.c_char_p * n)(*[r[0].encode() for r in rows])
    vecs = np.ascontiguousarray(np.array([r[1] for r in rows], dtype=np.float32))
    L = lib()
    L.fso_fsvi_write_quant.restype = C.c_int
    L.fso_fsvi_write_quant.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64,
                                       C.POINTER(C.c_char_p), C.c_void_p, C.c_uint8, C.c_uint8]
    return L.fso_fsvi_write_quant(path.encode(), embedder_id.encode(), revision.encode(), dim, n, ids, _p(vecs),
                                  compaction_gen, quantization)


def dot_f32_bytes_f32(row_f32: np.ndarray, q: np.ndarray, hreduce: int = HREDUCE_SSE2) -> float:
    """dot_product_f32_bytes_f32 (simd.rs:581-702) of one Quantization::F32 row."""
    row = np.ascontiguousarray(row_f32, dtype="<f4")
    q = np.ascontiguousarray(q, dtype=np.float32)
    L = lib()
    L.fso_dot_f32_bytes_f32.restype = C.c_float
    L.fso_dot_f32_bytes_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return float(L.fso_dot_f32_bytes_f32(row.ctypes.data, q.ctypes.data, q.size, hreduce))


def search_top_k_f32(slab_f32: np.ndarray, q: np.ndarray, k: int, live: np.ndarray | None = None, nthreads: int = 1,
                     hreduce: int = HREDUCE_SSE2):
    """search_top_k over a Quantization::F32 slab ([N, dim] float32): (rows, scores) best first."""
    slab = np.ascontiguousarray(slab_f32, dtype="<f4")
    n, dim = slab.shape
    q = np.ascontiguousarray(q, dtype=np.float32)
    if q.size != dim:
        raise ValueError(f"DimensionMismatch expected={dim} found={q.size}")
    cap = max(1, min(k, n))
    rows = np.empty(cap, dtype=np.uint32)
    scores = np.empty(cap, dtype=np.float32)
    bm = None
    if live is not None:
        bm = live_bitmap(np.asarray(live, dtype=bool)) if live.dtype != np.uint64 else live
    L = lib()
    L.fso_search_top_k_f32.restype = C.c_size_t
    L.fso_search_top_k_f32.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                       C.c_int, C.c_void_p, C.c_void_p]
    cnt = L.fso_search_top_k_f32(slab.ctypes.data, n, dim, bm.ctypes.data if bm is not None else None, q.ctypes.data, k,
                                 nthreads, hreduce, rows.ctypes.data, scores.ctypes.data)
    return rows[:cnt].copy(), scores[:cnt].copy()


class Fsvi:
    def __init__(self, path: str):
        self.h = None
        h = C.c_void_p()
        st = lib().fso_fsvi_open(path.encode(), C.byref(h))
        if st != OK:
            raise IOError(f"fso_fsvi_open failed: status {st}")
        self.h = h
        self.status = st

    def close(self):
        if self.h:
            lib().fso_fsvi_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    @property
    def record_count(self) -> int:
        return lib().fso_fsvi_record_count(self.h)

    @property
    def dimension(self) -> int:
        return lib().fso_fsvi_dimension(self.h)

    @property
    def vectors_offset(self) -> int:
        return lib().fso_fsvi_vectors_offset(self.h)

    def slab(self) -> np.ndarray:
        n, d = self.record_count, self.dimension
        ptr = lib().fso_fsvi_slab(self.h)
        buf = (C.c_uint16 * (n * d)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint16).reshape(n, d).copy()

    def doc_id(self, row: int) -> str:
        p = C.c_void_p()
        n = self.record_count
        if row >= n:  # WAL virtual row (search.rs:1579-1596)
            ln = lib().fso_fsvi_wal_doc_id(self.h, row - n, C.byref(p))
        else:
            ln = lib().fso_fsvi_doc_id(self.h, row, C.byref(p))
        return C.string_at(p.value, ln).decode()

    def append(self, doc_id: str, vector) -> int:
        v = np.ascontiguousarray(vector, dtype=np.float32)
        return lib().fso_fsvi_append(self.h, doc_id.encode(), _p(v), v.size)

    @property
    def wal_record_count(self) -> int:
        return lib().fso_fsvi_wal_count(self.h)

    def flags(self, row: int) -> int:
        return lib().fso_fsvi_flags(self.h, row)

    def soft_delete(self, doc_id: str) -> bool:
        # VectorIndex::soft_delete = soft_delete_batch(&[id]) > 0 (lib.rs:2303-2305): main rows AND resident WAL entries
        return lib().fso_fsvi_soft_delete(self.h, doc_id.encode()) > 0

    def search_top_k(self, q, k: int, hreduce: int = HREDUCE_SSE2):
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.size != self.dimension:
            raise ValueError(f"DimensionMismatch expected={self.dimension} found={q.size}")
        cap = max(1, min(k, self.record_count + self.wal_record_count))
        rows = np.empty(cap, dtype=np.uint32)
        scores = np.empty(cap, dtype=np.float32)
        cnt = lib().fso_fsvi_search(self.h, _p(q), k, hreduce, _p(rows), _p(scores))
        return [(int(rows[i]), float(scores[i]), self.doc_id(int(rows[i]))) for i in range(cnt)], scores[:cnt].copy()


# ---- fixtures ----
def fixture_hashmix(count: int, dim: int) -> np.ndarray:
    out = np.empty((count, dim), dtype=np.float32)
    L = lib()
    for i in range(count):
        for j in range(dim):
            out[i, j] = L.fso_fixture_hashmix(i, j)
    return out


def raw_vector(seed: int, dim: int) -> np.ndarray:
    out = np.empty(dim, dtype=np.float32)
    lib().fso_raw_vector(seed, dim, _p(out))
    return out


def clustered_corpus_f16(row0: int, n: int, dim: int, clusters: int = 64, noise: float = 0.30) -> np.ndarray:
    out = np.empty((n, dim), dtype=np.uint16)
    lib().fso_clustered_corpus_f16(row0, n, dim, clusters, noise, _p(out))
    return out


def clustered_query(q: int, dim: int, clusters: int = 64, noise: float = 0.30) -> np.ndarray:
    out = np.empty(dim, dtype=np.float32)
    lib().fso_clustered_query(q, dim, clusters, noise, _p(out))
    return out


def recall_fixture(dim: int = 384, count: int = 4000):
    """The reference's recall fixture (search.rs:1931-1967, int8_two_pass_maddubs_preserves_recall_vs_flat): 16 hash-mixed
    centroids, `count` rows = normalize(centroid[i % 16] + 0.15 jitter) — the same integer mixing.  -> (centroids, rows) f32."""
    j = np.arange(dim, dtype=np.uint64)

    def mix(a, mul_a, mul_j):
        s = (np.uint64(a) * np.uint64(mul_a)) ^ (j * np.uint64(mul_j))
        s ^= s >> np.uint64(13)
        return ((s & np.uint64(0xFFFF)).astype(np.float32) / np.float32(65535.0)) - np.float32(0.5)

    def normalize(v):
        n = max(float(np.sqrt(np.sum(v.astype(np.float32) ** 2, dtype=np.float32))), 1e-9)
        return (v / np.float32(n)).astype(np.float32)

    with np.errstate(over="ignore"):
        cent = np.stack([normalize(mix(c + 1, 0x9E37, 40503)) for c in range(16)])
        rows = np.stack([normalize(cent[i % 16] + np.float32(0.15) * mix(i + 1, 2654435761, 7)) for i in range(count)])
    return cent, rows


# ---- Model2Vec ----
def m2v_embed(table: np.ndarray, ids) -> np.ndarray:
    table = np.ascontiguousarray(table, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    out = np.empty(table.shape[1], dtype=np.float32)
    lib().fso_m2v_embed(_p(table), table.shape[0], table.shape[1], _p(ids), ids.size, _p(out))
    return out
