"""ctypes binding of libfshost.so — the C++ host-side mirror of the reference's two-phase searcher
(include/fshost.h; crates/frankensearch-fusion/src/sync_searcher.rs:616-943) and its native load generator.

libfshost.so contains no GPU code: it calls only the C ABI of libfsgpu.so, from native threads, the way the Rust
host of INTEGRATION.md would."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .errors import check
from .fusion import FusedHit

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfshost.so")
DOC_ID_MAX = 63


class _Config(C.Structure):
    _fields_ = [("quality_weight", C.c_float), ("rrf_k", C.c_double), ("candidate_multiplier", C.c_uint32),
                ("doc_id_mode", C.c_int32), ("fast_tier_int8_multiplier", C.c_uint32),
                ("prefetch_quality_embed", C.c_int32), ("quality_pool", C.c_int32), ("quality_int8_latency", C.c_int32)]


class _Hit(C.Structure):
    _fields_ = [("doc_id", C.c_char * (DOC_ID_MAX + 1)), ("rrf_score", C.c_double), ("lexical_rank", C.c_int64),
                ("semantic_rank", C.c_int64), ("semantic_index", C.c_uint32), ("lexical_score", C.c_float),
                ("semantic_score", C.c_float), ("in_both_sources", C.c_uint8)]


class _Metrics(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fast_embed_ms", "fast_search_ms", "phase1_total_ms", "quality_embed_ms",
                                          "quality_search_ms", "blend_ms", "phase2_total_ms")] + [("refinement_failed", C.c_int32)]


class _StreamResult(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("wall_seconds", "queries_per_sec", "mean_encode_ms", "mean_search_ms")] + \
               [(n, C.c_uint64) for n in ("queries", "groups", "exact_fallbacks", "device_resident_handoff", "encoders")] + \
               [("error_detail", C.c_char * 256)]


class _ScoredDoc(C.Structure):
    _fields_ = [("doc_id", C.c_char_p), ("doc_id_len", C.c_uint32), ("score", C.c_float), ("index", C.c_uint32)]


class _LoadConfig(C.Structure):
    _fields_ = [("threads", C.c_uint32), ("queries", C.c_uint32), ("warmup_queries", C.c_uint32), ("k", C.c_uint32),
                ("fast_vocab", C.c_uint32), ("quality_vocab", C.c_uint32), ("corpus_rows", C.c_uint64), ("seed", C.c_uint64)]


class _LoadResult(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "wall_seconds", "queries_per_sec", "phase0_p50_ms", "phase0_p95_ms", "phase0_p99_ms", "phase1_p50_ms",
        "phase1_p95_ms", "phase1_p99_ms", "mean_fast_embed_ms", "mean_fast_search_ms", "mean_quality_embed_ms",
        "mean_quality_search_ms", "mean_fusion_ms")] + [("completed", C.c_uint64), ("failed", C.c_uint64),
                                                  ("first_error", C.c_char * 160)]


class _ManyResult(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("wall_seconds", "queries_per_sec", "mean_fast_embed_ms", "mean_fast_search_ms",
                                          "mean_quality_embed_ms", "mean_quality_search_ms", "fusion_busy_ms_per_chunk",
                                          "first_chunk_initial_ms", "first_chunk_refined_ms")] + \
               [(n, C.c_uint64) for n in ("queries", "chunks", "chunk_queries", "fusion_threads", "refinement_failed", "fast_fallbacks",
                                          "quality_fallbacks", "device_resident_handoff", "queries_with_k_initial_and_refined_hits")] + \
               [("error_detail", C.c_char * 256)]


SYMBOLS = ("fshost_two_tier_create", "fshost_two_tier_create_sharded", "fshost_two_tier_destroy", "fshost_two_tier_search",
           "fshost_two_tier_search_many", "fshost_run_load_many", "fshost_two_tier_set_batching", "fshost_two_tier_batching_stats", "fshost_run_load", "fshost_embed_search_stream", "fshost_embed_search_stream_dp")
_handle = None


def lib() -> C.CDLL:
    global _handle
    if _handle is None:
        _lib.lib()  # libfsgpu.so first (and its loud failure when missing)
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -m frankensearch_amd.build`")
        h = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        h.fshost_two_tier_create.restype = C.c_int32
        h.fshost_two_tier_create.argtypes = [C.c_void_p] * 4 + [C.POINTER(_Config), C.POINTER(C.c_void_p)]
        h.fshost_two_tier_create_sharded.restype = C.c_int32
        h.fshost_two_tier_create_sharded.argtypes = [C.c_void_p] * 4 + [C.POINTER(_Config), C.POINTER(C.c_void_p)]
        h.fshost_embed_search_stream.restype = C.c_int32
        h.fshost_embed_search_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                 C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.POINTER(_StreamResult)]
        h.fshost_embed_search_stream_dp.restype = C.c_int32
        h.fshost_embed_search_stream_dp.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                    C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.POINTER(_StreamResult)]
        h.fshost_two_tier_destroy.restype = None
        h.fshost_two_tier_destroy.argtypes = [C.c_void_p]
        h.fshost_two_tier_search.restype = C.c_int32
        h.fshost_two_tier_search.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                             C.POINTER(_ScoredDoc), C.c_uint32, C.POINTER(_Hit), C.POINTER(C.c_uint32),
                                             C.POINTER(_Hit), C.POINTER(C.c_uint32), C.POINTER(_Metrics)]
        h.fshost_run_load.restype = C.c_int32
        h.fshost_run_load.argtypes = [C.c_void_p, C.POINTER(_LoadConfig), C.POINTER(_LoadResult)]
        h.fshost_two_tier_set_batching.restype = C.c_int32
        h.fshost_two_tier_set_batching.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        h.fshost_two_tier_batching_stats.restype = C.c_int32
        h.fshost_two_tier_batching_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        h.fshost_run_load_many.restype = C.c_int32
        h.fshost_run_load_many.argtypes = [C.c_void_p, C.POINTER(_LoadConfig), C.c_uint32, C.POINTER(_ManyResult)]
        h.fshost_two_tier_search_many.restype = C.c_int32
        h.fshost_two_tier_search_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                  C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_ManyResult)]
        _handle = h
    return _handle


@dataclass
class LoadResult:
    wall_seconds: float
    queries_per_sec: float
    phase0_p50_ms: float
    phase0_p95_ms: float
    phase0_p99_ms: float
    phase1_p50_ms: float
    phase1_p95_ms: float
    phase1_p99_ms: float
    mean_fast_embed_ms: float
    mean_fast_search_ms: float
    mean_quality_embed_ms: float
    mean_quality_search_ms: float
    mean_fusion_ms: float
    completed: int
    failed: int
    first_error: str = ""


def _fused(h: _Hit) -> FusedHit:
    return FusedHit(h.doc_id.decode(), h.rrf_score, None if h.lexical_rank < 0 else int(h.lexical_rank),
                    None if h.semantic_rank < 0 else int(h.semantic_rank),
                    None if h.semantic_index == 0xFFFFFFFF else int(h.semantic_index),
                    h.lexical_score if h.lexical_rank >= 0 else None,
                    h.semantic_score if h.semantic_rank >= 0 else None, bool(h.in_both_sources))


class NativeTwoTierSearcher:
    """SyncTwoTierSearcher in C++ over the C ABI (doc_id_mode: 0 = FSVI doc-id tables, 1 = "doc-%08u" of the row)."""

    def __init__(self, fast_index, quality_index, fast_embedder, quality_embedder, quality_weight: float = 0.7,
                 rrf_k: float = 60.0, candidate_multiplier: int = 3, doc_id_mode: int = 0,
                 fast_tier_int8_multiplier: int = 0, prefetch_quality_embed: bool = False, quality_pool: int = 0,
                 quality_int8_latency: bool = False):
        """quality_pool: 0 = Retrieved (independent quality-tier search; attested FSVI v2 pairs), 1 = RescoredFastPool
        (quality_scores_for_hits over the fast pool: every FSVI v1 pair) — sync_searcher.rs:810-818."""
        self._keep = (fast_index, quality_index, fast_embedder, quality_embedder)
        cfg = _Config(quality_weight, rrf_k, candidate_multiplier, doc_id_mode, fast_tier_int8_multiplier,
                      int(prefetch_quality_embed), int(quality_pool), int(quality_int8_latency))
        h = C.c_void_p()
        from .index import NativeShardedIndex
        # two row-sharded handles (fsgpu_sharded over the GPUs of the node) or two indexes: the same searcher either way
        create = lib().fshost_two_tier_create_sharded if isinstance(fast_index, NativeShardedIndex) else lib().fshost_two_tier_create
        check(create(fast_index._h, quality_index._h, fast_embedder._h, quality_embedder._h, C.byref(cfg), C.byref(h)))
        self._h = h

    def search(self, fast_token_ids: Sequence[int], quality_token_ids: Sequence[int], k: int,
               lexical: Optional[Sequence[Tuple[str, float]]] = None):
        f = np.ascontiguousarray(fast_token_ids, dtype=np.uint32)
        q = np.ascontiguousarray(quality_token_ids, dtype=np.int32)
        lex = list(lexical or [])
        ids = [d.encode() for d, _ in lex]
        arr = (_ScoredDoc * max(len(lex), 1))()
        for i, (b, (_, s)) in enumerate(zip(ids, lex)):
            arr[i] = _ScoredDoc(b, len(b), s, 0)
        ini, fin = (_Hit * max(k, 1))(), (_Hit * max(k, 1))()
        ni, nf, m = C.c_uint32(), C.c_uint32(), _Metrics()
        check(lib().fshost_two_tier_search(self._h, f.ctypes.data, f.size, q.ctypes.data, q.size, k, arr, len(lex), ini,
                                           C.byref(ni), fin, C.byref(nf), C.byref(m)))
        metrics = {n: getattr(m, n) for n, _ in _Metrics._fields_}
        return [_fused(ini[i]) for i in range(ni.value)], [_fused(fin[i]) for i in range(nf.value)], metrics

    def run_load(self, threads: int, queries: int, warmup_queries: int, k: int, fast_vocab: int, corpus_rows: int,
                 quality_vocab: int = 30000, seed: int = 1) -> LoadResult:
        cfg = _LoadConfig(threads, queries, warmup_queries, k, fast_vocab, quality_vocab, corpus_rows, seed)
        res = _LoadResult()
        check(lib().fshost_run_load(self._h, C.byref(cfg), C.byref(res)))
        d = {n: getattr(res, n) for n, _ in _LoadResult._fields_}
        d["first_error"] = d["first_error"].decode(errors="replace")
        return LoadResult(**d)

    def search_many(self, fast_token_ids: Sequence[Sequence[int]], quality_token_ids: Sequence[Sequence[int]], k: int,
                    lexical: Optional[Sequence[Sequence[Tuple[str, float]]]] = None, chunk: int = 0, fusion_threads: int = 0,
                    want_vectors: bool = False):
        """fshost_two_tier_search_many: the two-phase flow for a list of queries in one call -> (initial hit lists, final hit lists,
        refinement_failed flags, stats[, fast vectors, quality vectors])."""
        nq = len(fast_token_ids)
        assert len(quality_token_ids) == nq and (lexical is None or len(lexical) == nq)
        f_off = np.zeros(nq + 1, dtype=np.uint32)
        q_off = np.zeros(nq + 1, dtype=np.uint32)
        f_off[1:] = np.cumsum([len(x) for x in fast_token_ids])
        q_off[1:] = np.cumsum([len(x) for x in quality_token_ids])
        f = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.uint32) for x in fast_token_ids]) if nq else np.zeros(0, np.uint32))
        q = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.int32) for x in quality_token_ids]) if nq else np.zeros(0, np.int32))
        lex_arr = lex_off = None
        keep = []
        if lexical is not None:
            lex_off = np.zeros(nq + 1, dtype=np.uint32)
            lex_off[1:] = np.cumsum([len(x) for x in lexical])
            lex_arr = (_ScoredDoc * max(int(lex_off[-1]), 1))()
            j = 0
            for lst in lexical:
                for d, s_ in lst:
                    b = d.encode()
                    keep.append(b)
                    lex_arr[j] = _ScoredDoc(b, len(b), s_, 0)
                    j += 1
        ini, fin = (_Hit * max(nq * k, 1))(), (_Hit * max(nq * k, 1))()
        ni, nf = np.zeros(nq, dtype=np.uint32), np.zeros(nq, dtype=np.uint32)
        rf = np.zeros(nq, dtype=np.uint8)
        fdim = _lib.lib().fsgpu_m2v_dimension(self._keep[2]._h)
        qdim = _lib.lib().fsgpu_bert_dimension(self._keep[3]._h)
        fv = np.empty((nq, fdim), dtype=np.float32) if want_vectors else None
        qv = np.empty((nq, qdim), dtype=np.float32) if want_vectors else None
        res = _ManyResult()
        st = lib().fshost_two_tier_search_many(self._h, f.ctypes.data, f_off.ctypes.data, q.ctypes.data, q_off.ctypes.data, nq, k,
                                               C.cast(lex_arr, C.c_void_p) if lex_arr is not None else None,
                                               lex_off.ctypes.data if lex_off is not None else None, chunk, fusion_threads,
                                               C.cast(ini, C.c_void_p), ni.ctypes.data, C.cast(fin, C.c_void_p), nf.ctypes.data, rf.ctypes.data,
                                               fv.ctypes.data if want_vectors else None, qv.ctypes.data if want_vectors else None, C.byref(res))
        if st != 0 and res.error_detail:
            from .errors import error_for
            raise error_for(st, res.error_detail.decode(errors="replace"))
        check(st)
        initial = [[_fused(ini[i * k + j]) for j in range(int(ni[i]))] for i in range(nq)]
        final = [[_fused(fin[i * k + j]) for j in range(int(nf[i]))] for i in range(nq)]
        stats = {n: getattr(res, n) for n, _ in _ManyResult._fields_ if n != "error_detail"}
        stats["error_detail"] = res.error_detail.decode(errors="replace")
        out = (initial, final, rf.astype(bool), stats)
        return out + (fv, qv) if want_vectors else out

    def set_batching(self, max_chunk: int, max_wait_us: int = 200) -> None:
        """fshost_two_tier_set_batching: concurrent search() callers ride the many-queries pipeline (0 = off)."""
        check(lib().fshost_two_tier_set_batching(self._h, max_chunk, max_wait_us))

    def batching_stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(lib().fshost_two_tier_batching_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def run_load_many(self, queries: int, warmup_queries: int, k: int, fast_vocab: int, corpus_rows: int, chunk: int = 0,
                      fusion_threads: int = 0, quality_vocab: int = 30000, seed: int = 1) -> dict:
        """fshost_run_load_many: the load generator's synthetic queries through ONE fshost_two_tier_search_many call."""
        cfg = _LoadConfig(fusion_threads, queries, warmup_queries, k, fast_vocab, quality_vocab, corpus_rows, seed)
        res = _ManyResult()
        st = lib().fshost_run_load_many(self._h, C.byref(cfg), chunk, C.byref(res))
        if st != 0 and res.error_detail:
            from .errors import error_for
            raise error_for(st, res.error_detail.decode(errors="replace"))
        check(st)
        d = {n: getattr(res, n) for n, _ in _ManyResult._fields_ if n != "error_detail"}
        d["error_detail"] = res.error_detail.decode(errors="replace")
        return d

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().fshost_two_tier_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def embed_search_stream(encoder, index, ids: np.ndarray, offsets: np.ndarray, batch: int, k: int, group: int = 1,
                        overlap: bool = True, want_hits: bool = True, host_handoff: bool = False):
    """BASELINE config 5's serving loop in native code (fshost_embed_search_stream): token-id batches -> MiniLM on the GPU ->
    batched exact top-k, the encode of group g + 1 overlapped with the search of group g.  `index` is a VectorIndex or a
    NativeShardedIndex.  Returns (rows, scores, counts, stats dict)."""
    from .index import NativeShardedIndex
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    n_texts = offsets.size - 1
    if n_texts % batch:
        raise ValueError("the number of texts must be a multiple of the batch size")
    n_batches = n_texts // batch
    rows = np.empty((n_texts, k), dtype=np.uint32) if want_hits else None
    scores = np.empty((n_texts, k), dtype=np.float32) if want_hits else None
    counts = np.empty(n_texts, dtype=np.uint32) if want_hits else None
    res = _StreamResult()
    sharded = isinstance(index, NativeShardedIndex)
    out = (rows.ctypes.data if want_hits else None, scores.ctypes.data if want_hits else None, counts.ctypes.data if want_hits else None)
    if isinstance(encoder, (list, tuple)):
        # data-parallel encoders over a sharded handle (fshost_embed_search_stream_dp): one per device, each embeds a slice of every group
        if not sharded:
            raise TypeError("data-parallel encoders need a NativeShardedIndex")
        handles = (C.c_void_p * len(encoder))(*[e._h for e in encoder])
        st = lib().fshost_embed_search_stream_dp(handles, len(encoder), index._h, ids.ctypes.data, offsets.ctypes.data, batch, n_batches,
                                                 group, k, int(overlap), *out, C.byref(res))
    else:
        st = lib().fshost_embed_search_stream(encoder._h, None if sharded else index._h, index._h if sharded else None, ids.ctypes.data,
                                              offsets.ctypes.data, batch, n_batches, group, k, int(overlap) | (2 if host_handoff else 0),
                                              *out, C.byref(res))
    if st != 0 and res.error_detail:
        from .errors import error_for
        raise error_for(st, res.error_detail.decode(errors="replace"))
    check(st)
    stats = {n: getattr(res, n) for n, _ in _StreamResult._fields_ if n != "error_detail"}
    return rows, scores, counts, stats
