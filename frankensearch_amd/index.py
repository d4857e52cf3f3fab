"""VectorIndex — host-side mirror of the reference's `frankensearch_index::VectorIndex` method set
(crates/frankensearch-index/src/lib.rs:819, src/search.rs:192-494) over the C ABI of libfsgpu.so.

Same names, argument meaning and error behaviour as the reference so the parity tests read like the
reference's own tests; all compute happens in the HIP library.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .errors import check


@dataclass(frozen=True)
class VectorHit:
    """crates/frankensearch-core/src/types.rs:88-95"""
    index: int
    score: float
    doc_id: Optional[str] = None


@dataclass(frozen=True)
class ClassifiedHits:
    """search.rs:66-89: zero_signal is set iff hits is empty."""
    hits: List[VectorHit]
    zero_signal: Optional[str]


_ZERO_SIGNAL = {1: "CallerRequestedZeroK", 2: "ZeroNormQuery", 3: "FilterEliminatedAll", 4: "NewlyCreatedEmpty",
                5: "AllTombstoned", 6: "WalOnlyNoLiveRecords", 7: "NoUsableVectors"}


def _ptr(a) -> Optional[int]:
    return None if a is None else a.ctypes.data


def pack_bitmap(mask: np.ndarray) -> np.ndarray:
    """bool[N] -> uint64 words (bit r of word r//64 = mask[r]); uint64 input is taken as already packed."""
    if isinstance(mask, np.ndarray) and mask.dtype == np.uint64:
        return np.ascontiguousarray(mask)
    mask = np.asarray(mask, dtype=bool)
    words = (mask.size + 63) // 64
    padded = np.zeros(words * 64, dtype=np.uint8)
    padded[: mask.size] = mask
    return np.packbits(padded, bitorder="little").view(np.uint64).copy()


class _MrlStats(C.Structure):
    _fields_ = [("scan_dims", C.c_uint32), ("rescore_dims", C.c_uint32), ("candidates_rescored", C.c_uint32),
                ("records_scanned", C.c_uint64), ("fell_back_to_full", C.c_int32)]


class VectorIndex:
    # filter of search_batched applied to every new handle (0 = the library's automatic choice); the test-suite pins it to
    # run the same cases under the int8 and the f16 filter
    default_batched_filter = 0

    def __init__(self, handle: int, keepalive=None):
        self._h = C.c_void_p(handle)
        self._keepalive = keepalive
        if type(self).default_batched_filter:
            self.set_batched_filter(type(self).default_batched_filter)

    # ---- constructors -------------------------------------------------------------------------
    @classmethod
    def from_slab(cls, slab_f16: np.ndarray, live: Optional[np.ndarray] = None, device: int = 0,
                  row_base: int = 0) -> "VectorIndex":
        """slab_f16: [N, dim] uint16/float16 (little-endian f16 rows, the FSVI slab)."""
        slab = np.ascontiguousarray(slab_f16)
        if slab.dtype == np.float16:
            slab = slab.view(np.uint16)
        if slab.dtype != np.uint16 or slab.ndim != 2:
            raise TypeError("slab must be a 2-D uint16/float16 array")
        bm = pack_bitmap(live) if live is not None else None
        h = C.c_void_p()
        check(_lib.lib().fsgpu_index_create(device, slab.shape[1], slab.shape[0], _ptr(slab), _ptr(bm), row_base,
                                            C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_device_slab(cls, data_ptr: int, nrows: int, dim: int, live_ptr: Optional[int] = None, device: int = 0,
                         row_base: int = 0, keepalive=None) -> "VectorIndex":
        """Adopts a device-resident slab (e.g. a torch tensor's data_ptr()); `keepalive` pins its owner."""
        h = C.c_void_p()
        check(_lib.lib().fsgpu_index_create_device(device, dim, nrows, data_ptr, live_ptr, row_base, C.byref(h)))
        return cls(h.value, keepalive)

    @classmethod
    def open(cls, path: str, device: int = 0) -> "VectorIndex":
        """VectorIndex::open for an FSVI v1 / F16 file."""
        h = C.c_void_p()
        check(_lib.lib().fsgpu_index_open_fsvi(str(path).encode(), device, C.byref(h)))
        return cls(h.value)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().fsgpu_index_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- metadata -----------------------------------------------------------------------------
    def record_count(self) -> int:
        return _lib.lib().fsgpu_index_record_count(self._h)

    def dimension(self) -> int:
        return _lib.lib().fsgpu_index_dimension(self._h)

    def set_hreduce(self, mode: int) -> None:
        check(_lib.lib().fsgpu_index_set_hreduce(self._h, mode))

    def set_batched_filter(self, filter: int) -> None:
        """0 = automatic, 1 = f16 slab, 2 = int8 slab as the filter of search_batched (results identical either way)."""
        check(_lib.lib().fsgpu_index_set_batched_filter(self._h, filter))

    def set_filter_rotation(self, mode: int) -> None:
        """0 = automatic, 1 = never, 2 = always: the int8 filter's copy is built from rotated rows (fsgpu_index_set_filter_rotation)."""
        check(_lib.lib().fsgpu_index_set_filter_rotation(self._h, mode))

    def filter_rotated(self) -> bool:
        return bool(_lib.lib().fsgpu_index_filter_rotated(self._h))

    def set_int8_latency(self, enabled: bool, build_now: bool = False) -> None:
        """Unfiltered search_batch calls of a few queries go through the int8 filter + exact re-score (same hits, half the bytes).
        build_now: the int8 copy is built before the call returns (FSGPU_INT8_LATENCY_BUILD_NOW) instead of at the first such search."""
        check(_lib.lib().fsgpu_index_set_int8_latency(self._h, 2 if (enabled and build_now) else int(enabled)))

    def int8_filter_bound(self, queries: np.ndarray, want_slab: bool = False):
        """(delta[nq], query_scale[nq], slab_scale, queries_i8[nq, dim], slab_i8 or None): the int8 filter's certificate."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        q = q.reshape(-1, q.shape[-1])
        nq, dim = q.shape
        delta = np.empty(nq, np.float32)
        qscale = np.empty(nq, np.float32)
        sscale = C.c_float(0)
        qi8 = np.empty((nq, dim), np.int8)
        slab = np.empty((self.record_count(), dim), np.int8) if want_slab else None
        check(_lib.lib().fsgpu_index_int8_filter_bound(self._h, _ptr(q), nq, dim, _ptr(delta), _ptr(qscale), C.byref(sscale), _ptr(qi8),
                                                       _ptr(slab)))
        return delta, qscale, sscale.value, qi8, slab

    def batched_filter_stats(self) -> dict:
        q, r, a = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        check(_lib.lib().fsgpu_index_batched_filter_stats(self._h, C.byref(q), C.byref(r), C.byref(a)))
        return {"int8_queries": q.value, "refiltered_f16": r.value, "int8_active": bool(a.value)}

    def doc_id_at(self, row: int) -> str:
        p, n = C.c_void_p(), C.c_uint32()
        check(_lib.lib().fsgpu_index_doc_id(self._h, row, C.byref(p), C.byref(n)))
        return C.string_at(p.value, n.value).decode()

    def soft_delete(self, doc_id: str) -> bool:
        b = doc_id.encode()
        d = C.c_int32()
        check(_lib.lib().fsgpu_index_soft_delete(self._h, b, len(b), C.byref(d)))
        return bool(d.value)

    def append(self, doc_id: str, vector: Sequence[float]) -> None:
        """VectorIndex::append (lib.rs:2532): resident WAL entry, immediately searchable."""
        b = doc_id.encode()
        v = np.ascontiguousarray(vector, dtype=np.float32).reshape(-1)
        check(_lib.lib().fsgpu_index_wal_append(self._h, b, len(b), _ptr(v), v.size))

    def wal_record_count(self) -> int:
        return _lib.lib().fsgpu_index_wal_record_count(self._h)

    def set_live(self, live: Optional[np.ndarray]) -> None:
        bm = pack_bitmap(live) if live is not None else None
        check(_lib.lib().fsgpu_index_set_live_bitmap(self._h, _ptr(bm)))

    # ---- search -------------------------------------------------------------------------------
    def search_batch(self, queries: np.ndarray, limit: int, allow: Optional[np.ndarray] = None, exact: bool = False
                     ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """nq queries at once -> (rows [nq,limit] u32, scores [nq,limit] f32, counts [nq] u32).
        exact=True: fsgpu_search_topk_exact — the exact f16 kernels whatever copies the index holds (a lone query of an index that
        holds the int8 copy is otherwise answered by the certified pass over it: same rows and score bits)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq, qlen = q.shape
        rows = np.full((nq, max(limit, 1)), 0xFFFFFFFF, dtype=np.uint32)
        scores = np.full((nq, max(limit, 1)), np.nan, dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        if isinstance(allow, ResidentFilter):   # uploaded once (fsgpu_allow_bitmap): no per-call copy of the bitmap
            check(_lib.lib().fsgpu_search_topk_filtered(self._h, _ptr(q), nq, qlen, limit, allow._h, _ptr(rows), _ptr(scores),
                                                        _ptr(counts)))
            return rows[:, :limit], scores[:, :limit], counts
        bm = pack_bitmap(allow) if allow is not None else None
        fn = _lib.lib().fsgpu_search_topk_exact if exact else _lib.lib().fsgpu_search_topk
        check(fn(self._h, _ptr(q), nq, qlen, limit, _ptr(bm), _ptr(rows), _ptr(scores), _ptr(counts)))
        return rows[:, :limit], scores[:, :limit], counts

    def resident_filter(self, allow: np.ndarray) -> "ResidentFilter":
        """A precomputed SearchFilter (bool[N] allow mask) made resident on this index's device; pass it as `allow=` to
        search_batch / search_batched any number of times."""
        return ResidentFilter(self, allow)

    def search_batched(self, queries: np.ndarray, limit: int, allow: Optional[np.ndarray] = None):
        """Throughput path (64 queries per HBM pass on the matrix cores, exact results):
        -> (rows, scores, counts, fallbacks)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq, qlen = q.shape
        rows = np.full((nq, max(limit, 1)), 0xFFFFFFFF, dtype=np.uint32)
        scores = np.full((nq, max(limit, 1)), np.nan, dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        fb = C.c_uint32()
        if isinstance(allow, ResidentFilter):
            check(_lib.lib().fsgpu_search_topk_batched_filtered(self._h, _ptr(q), nq, qlen, limit, allow._h, _ptr(rows), _ptr(scores),
                                                                _ptr(counts), C.byref(fb)))
            return rows[:, :limit], scores[:, :limit], counts, fb.value
        bm = pack_bitmap(allow) if allow is not None else None
        check(_lib.lib().fsgpu_search_topk_batched(self._h, _ptr(q), nq, qlen, limit, _ptr(bm), _ptr(rows), _ptr(scores),
                                                   _ptr(counts), C.byref(fb)))
        return rows[:, :limit], scores[:, :limit], counts, fb.value

    def search_top_k(self, query: Sequence[float], limit: int, filter: Optional[np.ndarray] = None
                     ) -> List[VectorHit]:
        """VectorIndex::search_top_k(query, limit, filter) (search.rs:192-206).  `filter` is a precomputed
        allow mask over rows (bool[N]); doc-id dedup applies when the index has a doc-id table."""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        if filter is None and self._has_doc_ids():
            cap = max(limit, 1)
            rows = np.empty(cap, dtype=np.uint32)
            scores = np.empty(cap, dtype=np.float32)
            n = C.c_uint32()
            check(_lib.lib().fsgpu_search_hits(self._h, _ptr(q), q.size, limit, _ptr(rows), _ptr(scores), C.byref(n)))
            return [VectorHit(int(rows[i]), float(scores[i]), self.doc_id_at(int(rows[i]))) for i in range(n.value)]
        rows, scores, counts = self.search_batch(q, limit, filter)
        return [VectorHit(int(rows[0, i]), float(scores[0, i])) for i in range(int(counts[0]))]

    def search_top_k_classified(self, query: Sequence[float], limit: int) -> ClassifiedHits:
        """search.rs:227-261"""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        rows = np.empty(max(limit, 1), dtype=np.uint32)
        scores = np.empty(max(limit, 1), dtype=np.float32)
        n, z = C.c_uint32(), C.c_int32()
        check(_lib.lib().fsgpu_search_topk_classified(self._h, _ptr(q), q.size, limit, _ptr(rows), _ptr(scores),
                                                      C.byref(n), C.byref(z)))
        hits = [VectorHit(int(rows[i]), float(scores[i])) for i in range(n.value)]
        return ClassifiedHits(hits, _ZERO_SIGNAL.get(z.value))

    def mrl_search_batched(self, queries: np.ndarray, limit: int, search_dims: int = 64, rescore_dims: int = 0,
                           rescore_top_k: int = 0):
        """fsgpu_search_mrl_batched: mrl_search (mrl.rs:241-395) for a batch -> rows [nq, k], scores, counts, fallbacks."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq = q.shape[0]
        rows = np.full((nq, max(limit, 1)), 0xFFFFFFFF, dtype=np.uint32)
        scores = np.zeros((nq, max(limit, 1)), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        fb = C.c_uint32()
        check(_lib.lib().fsgpu_search_mrl_batched(self._h, _ptr(q), nq, q.shape[1], limit, search_dims, rescore_dims, rescore_top_k,
                                                  _ptr(rows), _ptr(scores), _ptr(counts), C.byref(fb)))
        return rows[:, :limit], scores[:, :limit], counts, fb.value

    def mrl_search(self, query: Sequence[float], limit: int, search_dims: int = 64, rescore_dims: int = 0,
                   rescore_top_k: int = 0, with_stats: bool = False):
        """VectorIndex::mrl_search / mrl_search_with_stats (mrl.rs:241-395; MrlConfig defaults :79-87)."""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        cap = max(limit, 1)
        rows = np.empty(cap, dtype=np.uint32)
        scores = np.empty(cap, dtype=np.float32)
        n = C.c_uint32()
        st = _MrlStats()
        check(_lib.lib().fsgpu_search_mrl(self._h, _ptr(q), q.size, limit, search_dims, rescore_dims, rescore_top_k,
                                          _ptr(rows), _ptr(scores), C.byref(n), C.addressof(st)))
        ids = self._has_doc_ids()
        nrec = self.record_count if not callable(self.record_count) else self.record_count()
        hits = [VectorHit(int(rows[i]), float(scores[i]),
                          self.doc_id_at(int(rows[i])) if ids and int(rows[i]) < nrec else None)
                for i in range(n.value)]
        if with_stats:
            return hits, {"scan_dims": st.scan_dims, "rescore_dims": st.rescore_dims,
                          "candidates_rescored": st.candidates_rescored, "records_scanned": st.records_scanned,
                          "fell_back_to_full": bool(st.fell_back_to_full)}
        return hits

    def search_top_k_int8_two_pass(self, query: Sequence[float], k: int, candidate_multiplier: int = 3
                                   ) -> List[VectorHit]:
        """search.rs:514-661 — int8 pass-1 + exact f16 rescore."""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        rows = np.empty(max(k, 1), dtype=np.uint32)
        scores = np.empty(max(k, 1), dtype=np.float32)
        n = C.c_uint32()
        check(_lib.lib().fsgpu_search_topk_int8_two_pass(self._h, _ptr(q), q.size, k, candidate_multiplier, _ptr(rows),
                                                         _ptr(scores), C.byref(n)))
        has_ids = self._has_doc_ids()
        return [VectorHit(int(rows[i]), float(scores[i]), self.doc_id_at(int(rows[i])) if has_ids else None)
                for i in range(n.value)]

    def search_4bit_two_pass_batched(self, queries: np.ndarray, limit: int, candidate_multiplier: int = 5):
        """Batched search_top_k_4bit_two_pass -> (rows [nq, limit], scores, counts, fallbacks)."""
        return self.search_int8_two_pass_batched(queries, limit, candidate_multiplier, bits=4)

    def search_int8_two_pass_batched(self, queries: np.ndarray, limit: int, candidate_multiplier: int = 3, bits: int = 8):
        """Batched search_top_k_int8_two_pass (int8 MFMA pass 1 shared by the whole batch):
        -> (rows [nq, limit], scores, counts, fallbacks)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq, qlen = q.shape
        rows = np.full((nq, max(limit, 1)), 0xFFFFFFFF, dtype=np.uint32)
        scores = np.full((nq, max(limit, 1)), np.nan, dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        fb = C.c_uint32()
        fn = _lib.lib().fsgpu_search_topk_4bit_two_pass_batched if bits == 4 else _lib.lib().fsgpu_search_topk_int8_two_pass_batched
        check(fn(self._h, _ptr(q), nq, qlen, limit, candidate_multiplier, _ptr(rows), _ptr(scores), _ptr(counts), C.byref(fb)))
        return rows[:, :limit], scores[:, :limit], counts, fb.value

    def search_top_k_4bit_two_pass(self, query: Sequence[float], k: int, candidate_multiplier: int = 5
                                   ) -> List[VectorHit]:
        """search.rs:876-946 — packed 4-bit pass-1 + exact f16 rescore."""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        rows = np.empty(max(k, 1), dtype=np.uint32)
        scores = np.empty(max(k, 1), dtype=np.float32)
        n = C.c_uint32()
        check(_lib.lib().fsgpu_search_topk_4bit_two_pass(self._h, _ptr(q), q.size, k, candidate_multiplier, _ptr(rows),
                                                         _ptr(scores), C.byref(n)))
        has_ids = self._has_doc_ids()
        return [VectorHit(int(rows[i]), float(scores[i]), self.doc_id_at(int(rows[i])) if has_ids else None)
                for i in range(n.value)]

    def dot_query_at(self, index: int, query: Sequence[float]) -> float:
        """lib.rs:3229-3239"""
        return float(self.gather_dot(query, [index])[0])

    def gather_dot(self, query: Sequence[float], rows: Sequence[int]) -> np.ndarray:
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        out = np.empty(r.size, dtype=np.float32)
        check(_lib.lib().fsgpu_gather_dot(self._h, _ptr(q), q.size, _ptr(r), r.size, _ptr(out)))
        return out

    # device-pointer path (torch tensors): everything stays in HBM, enqueued on `stream`
    def search_device(self, queries_ptr: int, nq: int, limit: int, out_rows_ptr: int, out_scores_ptr: int,
                      out_counts_ptr: int, stream: int = 0, allow_ptr: Optional[int] = None) -> None:
        check(_lib.lib().fsgpu_search_topk_device(self._h, queries_ptr, nq, self.dimension(), limit, allow_ptr,
                                                  out_rows_ptr, out_scores_ptr, out_counts_ptr, stream))

    # ---- instrumentation ----------------------------------------------------------------------
    def set_profiling(self, enabled) -> None:
        """True / False, or an int n > 1: time every n-th main launch of the batched search only."""
        check(_lib.lib().fsgpu_index_set_profiling(self._h, int(enabled)))

    def scan_time(self, reset: bool = True) -> Tuple[float, int]:
        ms, n = C.c_double(), C.c_uint64()
        check(_lib.lib().fsgpu_index_scan_time(self._h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def scan_stats(self, reset: bool = True) -> Tuple[float, int, int]:
        """(total ms, launches, slab rows streamed) of the timed scan launches since the last reset."""
        ms, n, rows = C.c_double(), C.c_uint64(), C.c_uint64()
        check(_lib.lib().fsgpu_index_scan_stats(self._h, C.byref(ms), C.byref(n), C.byref(rows), int(reset)))
        return ms.value, n.value, rows.value

    def allow_bitmap_for_hashes(self, hashes: Sequence[int]) -> Tuple[np.ndarray, int]:
        """`SearchFilter::candidate_hashes` (FNV-1a doc-id hashes) -> (packed allow bitmap, rows matched)."""
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        bm = np.zeros((self.record_count() + 63) // 64, dtype=np.uint64)
        m = C.c_uint64()
        check(_lib.lib().fsgpu_index_allow_bitmap_for_hashes(self._h, h.ctypes.data_as(C.POINTER(C.c_uint64)), h.size,
                                                             bm.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(m)))
        return bm, m.value

    def filter_stats(self) -> Tuple[int, int]:
        """(filtered searches answered by scoring only the allowed rows, by the masked full scan)."""
        g, s = C.c_uint64(), C.c_uint64()
        check(_lib.lib().fsgpu_index_filter_stats(self._h, C.byref(g), C.byref(s)))
        return g.value, s.value

    def set_coalescing(self, max_batch: int, max_wait_us: int = 200) -> None:
        """Gather concurrent single-query `search_top_k` callers into one batched pass (0 = off)."""
        check(_lib.lib().fsgpu_index_set_coalescing(self._h, max_batch, max_wait_us))

    def coalescing_stats(self) -> Tuple[int, int]:
        b, r = C.c_uint64(), C.c_uint64()
        check(_lib.lib().fsgpu_index_coalescing_stats(self._h, C.byref(b), C.byref(r)))
        return b.value, r.value

    def set_variant(self, variant: int) -> None:
        check(_lib.lib().fsgpu_index_set_variant(self._h, variant))

    def _has_doc_ids(self) -> bool:
        p, n = C.c_void_p(), C.c_uint32()
        if self.record_count() + self.wal_record_count() == 0:
            return False
        return _lib.lib().fsgpu_index_doc_id(self._h, 0, C.byref(p), C.byref(n)) == 0


class ResidentFilter:
    """fsgpu_allow_bitmap: a precomputed SearchFilter (filter.rs:19-56) uploaded ONCE to an index's device and reused by any number
    of searches — the per-call ABI copies the bitmap (1.25 MB at 10M rows) on every search."""

    def __init__(self, index: "VectorIndex", allow: np.ndarray):
        self._index = index
        bm = pack_bitmap(allow)
        h = C.c_void_p()
        check(_lib.lib().fsgpu_allow_bitmap_create(index._h, _ptr(bm), C.byref(h)))
        self._h = h

    def allowed_rows(self) -> int:
        return int(_lib.lib().fsgpu_allow_bitmap_allowed_rows(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().fsgpu_allow_bitmap_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeShardedIndex:
    """fsgpu_sharded: the row-sharded index behind ONE C-ABI handle (include/fsgpu.h): one shard and stream pair per
    device (optionally query groups x row shards), everything enqueued by the calling thread, RCCL all-gather of the packed per-shard top-k, merge on the root (search.rs:1013-1036,1704-1720 at GPU granularity).
    (The one-process-per-GPU form used by `bench.py --gpus N` is frankensearch_amd/sharded.py.)"""

    EXCHANGE_AUTO, EXCHANGE_RCCL, EXCHANGE_PEER_COPY = 0, 1, 2

    def __init__(self, handle: int, keepalive=None):
        self._h = C.c_void_p(handle)
        self._keepalive = keepalive

    @classmethod
    def from_slab(cls, slab_f16: np.ndarray, devices: Sequence[int], live: Optional[np.ndarray] = None,
                  exchange: int = 0, query_groups: int = 1) -> "NativeShardedIndex":
        slab = np.ascontiguousarray(slab_f16)
        if slab.dtype == np.float16:
            slab = slab.view(np.uint16)
        if slab.dtype != np.uint16 or slab.ndim != 2:
            raise TypeError("slab must be a 2-D uint16/float16 array")
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        bm = pack_bitmap(live) if live is not None else None
        h = C.c_void_p()
        check(_lib.lib().fsgpu_sharded_create_grouped(_ptr(devs), devs.size, query_groups, slab.shape[1], slab.shape[0], _ptr(slab),
                                                      _ptr(bm), exchange, C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_device_slabs(cls, devices: Sequence[int], dim: int, shard_rows: Sequence[int], slab_ptrs: Sequence[int],
                          exchange: int = 0, keepalive=None, query_groups: int = 1) -> "NativeShardedIndex":
        """Adopts per-device resident shards (e.g. torch tensors' data_ptr()); with query groups, device r holds row shard
        r % (len(devices) // query_groups)."""
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        rows = np.ascontiguousarray(shard_rows, dtype=np.uint64)
        ptrs = np.ascontiguousarray(slab_ptrs, dtype=np.uint64)
        h = C.c_void_p()
        check(_lib.lib().fsgpu_sharded_create_device_grouped(_ptr(devs), devs.size, query_groups, dim, _ptr(rows), _ptr(ptrs), None,
                                                             exchange, C.byref(h)))
        return cls(h.value, keepalive)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().fsgpu_sharded_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def record_count(self) -> int:
        return _lib.lib().fsgpu_sharded_record_count(self._h)

    def dimension(self) -> int:
        return _lib.lib().fsgpu_sharded_dimension(self._h)

    def shard_count(self) -> int:
        return _lib.lib().fsgpu_sharded_shard_count(self._h)

    def query_groups(self) -> int:
        return _lib.lib().fsgpu_sharded_query_groups(self._h)

    def row_shards(self) -> int:
        return _lib.lib().fsgpu_sharded_row_shards(self._h)

    def set_int8_latency(self, on: bool = True) -> None:
        """Lone exact queries through the certified int8 pass of every shard (fsgpu_sharded_set_int8_latency)."""
        check(_lib.lib().fsgpu_sharded_set_int8_latency(self._h, 1 if on else 0))

    def search_parts(self, parts: Sequence[Tuple[int, int, int]], dim: int, k: int, mode: int = 1):
        """Queries resident in parts on several devices: parts = [(device pointer, count, device), ...] (fsgpu_sharded_search_parts)."""
        ptrs = np.ascontiguousarray([p[0] for p in parts], dtype=np.uint64)
        counts = np.ascontiguousarray([p[1] for p in parts], dtype=np.uint32)
        devs = np.ascontiguousarray([p[2] for p in parts], dtype=np.int32)
        nq = int(counts.sum())
        rq = self._Request(None, nq, dim, k, mode, 0, None, None)
        rows, scores, cnts = self._outputs(nq, k)
        fb = C.c_uint32()
        check(_lib.lib().fsgpu_sharded_search_parts(self._h, C.byref(rq), _ptr(ptrs), _ptr(counts), _ptr(devs), len(parts), _ptr(rows),
                                                    _ptr(scores), _ptr(cnts), C.byref(fb)))
        return rows[:, :k], scores[:, :k], cnts, fb.value

    def exchange_mode(self) -> int:
        return _lib.lib().fsgpu_sharded_exchange_mode(self._h)

    def shard_range(self, shard: int) -> Tuple[int, int]:
        lo, hi = C.c_uint64(), C.c_uint64()
        check(_lib.lib().fsgpu_sharded_shard_range(self._h, shard, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def set_hreduce(self, mode: int) -> None:
        check(_lib.lib().fsgpu_sharded_set_hreduce(self._h, mode))

    def search_batch(self, queries: np.ndarray, k: int, batched: bool = False):
        """[nq, dim] f32 -> rows [nq, k] u32, scores [nq, k] f32, counts [nq] u32 (+ fallbacks when batched)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        rows = np.full((nq, max(k, 1)), 0xFFFFFFFF, dtype=np.uint32)
        scores = np.zeros((nq, max(k, 1)), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        if batched:
            fb = C.c_uint32()
            check(_lib.lib().fsgpu_sharded_search_topk_batched(self._h, _ptr(q), nq, q.shape[1], k, _ptr(rows), _ptr(scores),
                                                               _ptr(counts), C.byref(fb)))
            return rows[:, :k], scores[:, :k], counts, fb.value
        check(_lib.lib().fsgpu_sharded_search_topk(self._h, _ptr(q), nq, q.shape[1], k, _ptr(rows), _ptr(scores), _ptr(counts)))
        return rows[:, :k], scores[:, :k], counts

    # ---- the general form: fsgpu_sharded_search / _begin / _end ------------------------------------------------------
    EXACT, BATCHED, INT8_TWO_PASS, FOURBIT_TWO_PASS = 0, 1, 2, 3

    class _Request(C.Structure):
        _fields_ = [("queries", C.c_void_p), ("nq", C.c_uint32), ("query_len", C.c_uint32), ("k", C.c_uint32),
                    ("mode", C.c_int32), ("candidate_multiplier", C.c_uint32), ("allow_bitmap", C.c_void_p), ("queries_dev", C.c_void_p)]

    def _request(self, queries, k, mode, multiplier, allow):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        bm = pack_bitmap(allow) if allow is not None else None
        rq = self._Request(q.ctypes.data, q.shape[0], q.shape[1], k, mode, multiplier, bm.ctypes.data if bm is not None else None, None)
        return rq, (q, bm)

    @staticmethod
    def _outputs(nq, k):
        return (np.full((nq, max(k, 1)), 0xFFFFFFFF, dtype=np.uint32), np.zeros((nq, max(k, 1)), dtype=np.float32),
                np.zeros(nq, dtype=np.uint32))

    def search(self, queries: np.ndarray, k: int, mode: int = 0, candidate_multiplier: int = 0, allow: Optional[np.ndarray] = None):
        """search_top_k(query, limit, filter) / the two-pass searches for a batch -> rows, scores, counts, fallbacks."""
        rq, keep = self._request(queries, k, mode, candidate_multiplier, allow)
        rows, scores, counts = self._outputs(rq.nq, k)
        fb = C.c_uint32()
        check(_lib.lib().fsgpu_sharded_search(self._h, C.byref(rq), _ptr(rows), _ptr(scores), _ptr(counts), C.byref(fb)))
        return rows[:, :k], scores[:, :k], counts, fb.value

    def search_begin(self, queries: np.ndarray, k: int, mode: int = 0, candidate_multiplier: int = 0,
                     allow: Optional[np.ndarray] = None):
        rq, keep = self._request(queries, k, mode, candidate_multiplier, allow)
        t = C.c_uint64()
        check(_lib.lib().fsgpu_sharded_search_begin(self._h, C.byref(rq), C.byref(t)))
        return (t.value, rq.nq, k)

    def search_end(self, ticket):
        t, nq, k = ticket
        rows, scores, counts = self._outputs(nq, k)
        fb = C.c_uint32()
        check(_lib.lib().fsgpu_sharded_search_end(self._h, t, _ptr(rows), _ptr(scores), _ptr(counts), C.byref(fb)))
        return rows[:, :k], scores[:, :k], counts, fb.value

    def quant_scale_max(self) -> float:
        return float(_lib.lib().fsgpu_sharded_quant_scale_max(self._h))

    @classmethod
    def open(cls, path: str, devices: Sequence[int], exchange: int = 0, query_groups: int = 1) -> "NativeShardedIndex":
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        check(_lib.lib().fsgpu_sharded_open_fsvi_grouped(path.encode(), _ptr(devs), devs.size, query_groups, exchange, C.byref(h)))
        return cls(h.value)

    def set_live(self, live: Optional[np.ndarray]) -> None:
        bm = pack_bitmap(live) if live is not None else None
        check(_lib.lib().fsgpu_sharded_set_live_bitmap(self._h, _ptr(bm)))

    def soft_delete(self, doc_id: str) -> bool:
        b = doc_id.encode()
        d = C.c_int32()
        check(_lib.lib().fsgpu_sharded_soft_delete(self._h, b, len(b), C.byref(d)))
        return bool(d.value)

    def append(self, doc_id: str, vector: Sequence[float]) -> None:
        b = doc_id.encode()
        v = np.ascontiguousarray(vector, dtype=np.float32)
        check(_lib.lib().fsgpu_sharded_wal_append(self._h, b, len(b), _ptr(v), v.size))

    def wal_record_count(self) -> int:
        return _lib.lib().fsgpu_sharded_wal_record_count(self._h)

    def doc_id_at(self, row: int) -> str:
        p, n = C.c_void_p(), C.c_uint32()
        check(_lib.lib().fsgpu_sharded_doc_id(self._h, row, C.byref(p), C.byref(n)))
        return C.string_at(p.value, n.value).decode()

    def search_top_k(self, query: Sequence[float], limit: int) -> List[VectorHit]:
        """VectorIndex::search_top_k with the resident WAL, shadowing and doc-id dedup (fsgpu_sharded_search_hits)."""
        q = np.ascontiguousarray(query, dtype=np.float32)
        rows = np.zeros(max(limit, 1), dtype=np.uint32)
        scores = np.zeros(max(limit, 1), dtype=np.float32)
        n = C.c_uint32()
        check(_lib.lib().fsgpu_sharded_search_hits(self._h, _ptr(q), q.size, limit, _ptr(rows), _ptr(scores), C.byref(n)))
        return [VectorHit(int(rows[i]), float(scores[i]), self.doc_id_at(int(rows[i]))) for i in range(n.value)]

    def gather_dot(self, query: Sequence[float], rows: Sequence[int]) -> np.ndarray:
        q = np.ascontiguousarray(query, dtype=np.float32)
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        out = np.zeros(r.size, dtype=np.float32)
        check(_lib.lib().fsgpu_sharded_gather_dot(self._h, _ptr(q), q.size, _ptr(r), r.size, _ptr(out)))
        return out

    def set_coalescing(self, max_batch: int, max_wait_us: int) -> None:
        """Concurrent single-query searches ride one search of the shards (fsgpu_sharded_set_coalescing)."""
        check(_lib.lib().fsgpu_sharded_set_coalescing(self._h, max_batch, max_wait_us))

    def coalescing_stats(self) -> Tuple[int, int]:
        b, r = C.c_uint64(), C.c_uint64()
        check(_lib.lib().fsgpu_sharded_coalescing_stats(self._h, C.byref(b), C.byref(r)))
        return b.value, r.value


def write_fsvi(path: str, rows, embedder_id: str = "test", embedder_revision: str = "", compaction_gen: int = 0,
               device: int = 0, quantization: int = 1) -> None:
    """VectorIndexWriter (lib.rs:3637-3672, 3752-3943): rows = [(doc_id, vector), ...] -> FSVI v1 file
    (quantization 1 = F16, the default; 0 = F32)."""
    rows = list(rows)
    n = len(rows)
    dim = len(rows[0][1]) if n else 1
    ids = [d.encode() for d, _ in rows]
    vec = np.ascontiguousarray([v for _, v in rows], dtype=np.float32).reshape(n, dim) if n else np.zeros((0, dim), np.float32)
    arr = (C.c_char_p * max(n, 1))(*ids)
    lens = np.array([len(b) for b in ids], dtype=np.uint32)
    check(_lib.lib().fsgpu_fsvi_write_quant(path.encode(), embedder_id.encode(), embedder_revision.encode(), dim, n,
                                            C.cast(arr, C.c_void_p), _ptr(lens) if n else None, _ptr(vec) if n else None,
                                            compaction_gen, device, quantization))


def encode_f32_to_f16(src: np.ndarray, device: int = 0) -> np.ndarray:
    """encode_f32_to_f16_extend (simd.rs:2245-2305) on the GPU."""
    s = np.ascontiguousarray(src, dtype=np.float32)
    out = np.empty(s.shape, dtype=np.uint16)
    check(_lib.lib().fsgpu_encode_f32_to_f16(device, _ptr(s), s.size, _ptr(out)))
    return out


def widen_f16_to_f32(src: np.ndarray, device: int = 0) -> np.ndarray:
    s = np.ascontiguousarray(src, dtype=np.uint16)
    out = np.empty(s.shape, dtype=np.float32)
    check(_lib.lib().fsgpu_widen_f16_to_f32(device, _ptr(s), s.size, _ptr(out)))
    return out
