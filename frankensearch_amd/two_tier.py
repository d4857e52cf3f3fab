"""SyncTwoTierSearcher — mirror of the reference's two-phase query flow over the GPU tier
(crates/frankensearch-fusion/src/sync_searcher.rs:616-943; fsfs shape: crates/frankensearch-fsfs/src/runtime.rs:8185-8355).

  phase 0 / "Initial":  fast embed (potion Model2Vec) -> fast-tier scan (fetch = k * candidate_multiplier) ->
                        RRF with the lexical list
  phase 1 / "Refined":  quality embed (MiniLM) -> the quality pool (sync_searcher.rs:810-818) -> blend -> RRF with the lexical
                        list again.  The pool is `Retrieved` (an independent quality-tier scan + blend_two_tier) only when the
                        quality tier's space identity is ATTESTED (an admitted FSVI v2 artifact); every FSVI v1 artifact takes
                        `RescoredFastPool`: TwoTierIndex::quality_scores_for_hits gather-rescores the fast pool on the quality
                        tier (two_tier.rs:1566-1631) and blend_two_tier_aligned blends it (blend.rs:213-294).
Lexical (BM25) search is the caller's: it stays on the CPU in the reference and is passed in as a ranked list.
Defaults follow TwoTierConfig (crates/frankensearch-core/src/config.rs:169-176)."""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

import ctypes as C

from . import _lib, fusion
from .errors import check

POOL_RETRIEVED = 0   # SyncQualityPool::Retrieved: attested quality tier (FSVI v2 admission) — independent retrieval
POOL_RESCORED = 1    # SyncQualityPool::RescoredFastPool: legacy / unattested pair (every FSVI v1 artifact)


@dataclass
class TwoTierConfig:
    quality_weight: float = 0.7
    rrf_k: float = 60.0
    candidate_multiplier: int = 3
    # 0 = exact f16 scan of the fast tier; n = search_top_k_int8_two_pass(query, fetch, n) — the reference's default with
    # n = FAST_TIER_MULT = 3 (crates/frankensearch-index/src/two_tier.rs:1318-1337; sync_searcher::search_fast_hits)
    fast_tier_int8_multiplier: int = 0
    quality_pool: int = POOL_RETRIEVED


class TwoTierIndex:
    """The fast / quality pairing of crates/frankensearch-index/src/two_tier.rs: QualityAlignment (:404-409, computed at open by
    the merge walk :750-866) and quality_scores_for_hits (:1566-1631), over fsgpu_alignment_* / fsgpu_quality_scores_for_hits."""
    NONE, ALIGNED, MAPPING = 0, 1, 2

    def __init__(self, fast_index, quality_index):
        """Both unsharded (VectorIndex) or both row-sharded handles (NativeShardedIndex: fsgpu_sharded_alignment_create /
        fsgpu_sharded_quality_scores_for_hits — SURVEY 8e: the tiers shard identically, re-scored rows go to their shards)."""
        from .index import NativeShardedIndex
        self.fast, self.quality = fast_index, quality_index
        self.sharded = isinstance(fast_index, NativeShardedIndex)
        if self.sharded != isinstance(quality_index, NativeShardedIndex):
            raise TypeError("a fast / quality pair is either two indexes or two sharded handles")
        h = C.c_void_p()
        create = _lib.lib().fsgpu_sharded_alignment_create if self.sharded else _lib.lib().fsgpu_alignment_create
        check(create(fast_index._h, quality_index._h, C.byref(h)))
        self._a = h.value

    def close(self) -> None:
        if getattr(self, "_a", None):
            _lib.lib().fsgpu_alignment_destroy(self._a)
            self._a = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alignment_kind(self) -> int:
        return int(_lib.lib().fsgpu_alignment_kind(self._a))

    def quality_row(self, fast_row: int) -> Optional[int]:
        r = int(_lib.lib().fsgpu_alignment_quality_row(self._a, fast_row))
        return None if r < 0 else r

    def unmatched_quality_docs(self) -> int:
        return int(_lib.lib().fsgpu_alignment_unmatched_quality_docs(self._a))

    def quality_scores_for_hits(self, query: Sequence[float], hits: Sequence[Tuple[str, float, int]]) -> List[Optional[float]]:
        q = np.ascontiguousarray(query, dtype=np.float32)
        arr, keep = fusion._pack(hits)
        scores = np.zeros(max(len(hits), 1), dtype=np.float32)
        present = np.zeros(max(len(hits), 1), dtype=np.uint8)
        fn = _lib.lib().fsgpu_sharded_quality_scores_for_hits if self.sharded else _lib.lib().fsgpu_quality_scores_for_hits
        check(fn(self.fast._h, self.quality._h, self._a, q.ctypes.data, q.size, arr, len(hits), scores.ctypes.data,
                 present.ctypes.data))
        return [float(scores[i]) if present[i] else None for i in range(len(hits))]

    def quality_scores_for_hits_batched(self, queries: np.ndarray, hit_lists: Sequence[Sequence[Tuple[str, float, int]]]) -> List[List[Optional[float]]]:
        """fsgpu_quality_scores_for_hits_batched: quality_scores_for_hits for a chunk of queries, ONE gather launch (unsharded pairs)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq = len(hit_lists)
        assert q.ndim == 2 and q.shape[0] == nq and not self.sharded
        flat = [h for hits in hit_lists for h in hits]
        offs = np.zeros(nq + 1, dtype=np.uint32)
        offs[1:] = np.cumsum([len(h) for h in hit_lists])
        arr, keep = fusion._pack(flat)
        scores = np.zeros(max(len(flat), 1), dtype=np.float32)
        present = np.zeros(max(len(flat), 1), dtype=np.uint8)
        check(_lib.lib().fsgpu_quality_scores_for_hits_batched(self.fast._h, self.quality._h, self._a, q.ctypes.data, nq, q.shape[1], arr,
                                                               offs.ctypes.data, scores.ctypes.data, present.ctypes.data))
        return [[float(scores[i]) if present[i] else None for i in range(int(offs[j]), int(offs[j + 1]))] for j in range(nq)]


@dataclass
class TwoTierMetrics:
    fast_embed_ms: float = 0.0
    fast_search_ms: float = 0.0
    phase1_total_ms: float = 0.0     # library name of the Initial stage (config.rs:465-480)
    quality_embed_ms: float = 0.0
    quality_search_ms: float = 0.0
    blend_ms: float = 0.0
    phase2_total_ms: float = 0.0     # library name of the Refined stage


@dataclass
class SearchOutcome:
    initial_results: List[fusion.FusedHit]
    final_results: List[fusion.FusedHit]
    fast_hits: List[Tuple[str, float, int]]
    quality_hits: List[Tuple[str, float, int]]
    blended: List[Tuple[str, float, int]]
    metrics: TwoTierMetrics = field(default_factory=TwoTierMetrics)


class SyncTwoTierSearcher:
    def __init__(self, fast_index, quality_index, fast_embedder, quality_embedder,
                 doc_id_of: Callable[[int], str], config: Optional[TwoTierConfig] = None):
        self.fast_index, self.quality_index = fast_index, quality_index
        self.fast_embedder, self.quality_embedder = fast_embedder, quality_embedder
        self.doc_id_of = doc_id_of
        self.config = config or TwoTierConfig()
        self.pair = TwoTierIndex(fast_index, quality_index) if self.config.quality_pool == POOL_RESCORED else None

    def _hits(self, index, vec: np.ndarray, fetch: int, int8_multiplier: int = 0) -> List[Tuple[str, float, int]]:
        if int8_multiplier:
            return [(self.doc_id_of(h.index), h.score, h.index)
                    for h in index.search_top_k_int8_two_pass(vec, fetch, int8_multiplier)]
        rows, scores, counts = index.search_batch(vec, fetch)
        n = int(counts[0])
        return [(self.doc_id_of(int(rows[0, i])), float(scores[0, i]), int(rows[0, i])) for i in range(n)]

    def search(self, fast_token_ids: Sequence[int], quality_token_ids: Sequence[int], k: int,
               lexical: Optional[Sequence[Tuple[str, float]]] = None) -> SearchOutcome:
        cfg = self.config
        fetch = max(k * max(cfg.candidate_multiplier, 1), k)   # candidate_count (rrf.rs:113-115)
        m = TwoTierMetrics()
        lex = list(lexical or [])
        t0 = time.perf_counter()
        fast_vec = self.fast_embedder.embed_token_ids(fast_token_ids)
        t1 = time.perf_counter()
        fast_hits = self._hits(self.fast_index, fast_vec, fetch, cfg.fast_tier_int8_multiplier)
        t2 = time.perf_counter()
        initial = fusion.rrf_fuse(lex, fast_hits, k, 0, k=cfg.rrf_k)
        t3 = time.perf_counter()
        m.fast_embed_ms, m.fast_search_ms, m.phase1_total_ms = (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t0) * 1e3
        quality_vec = self.quality_embedder.embed_token_ids(quality_token_ids)
        t4 = time.perf_counter()
        if self.pair is not None:   # RescoredFastPool: the fast pool re-scored on the quality tier (a gather, not a scan)
            scores = self.pair.quality_scores_for_hits(quality_vec, fast_hits)
            quality_hits = [(d, q, i) for (d, _, i), q in zip(fast_hits, scores) if q is not None]
            t5 = time.perf_counter()
            blended = fusion.blend_two_tier_aligned(fast_hits, scores, cfg.quality_weight)
        else:
            quality_hits = self._hits(self.quality_index, quality_vec, fetch)
            t5 = time.perf_counter()
            blended = fusion.blend_two_tier(fast_hits, quality_hits, cfg.quality_weight)
            fast_index_of = {d: i for d, _, i in fast_hits}
            blended = [(d, s, fast_index_of.get(d, 0xFFFFFFFF)) for d, s, _ in blended]   # sync_searcher.rs:880-891
        t6 = time.perf_counter()
        final = fusion.rrf_fuse(lex, blended, k, 0, k=cfg.rrf_k)
        t7 = time.perf_counter()
        m.quality_embed_ms, m.quality_search_ms = (t4 - t3) * 1e3, (t5 - t4) * 1e3
        m.blend_ms, m.phase2_total_ms = (t6 - t5) * 1e3, (t7 - t3) * 1e3
        return SearchOutcome(initial, final, fast_hits, quality_hits, blended, m)
