"""SyncTwoTierSearcher — mirror of the reference's two-phase query flow over the GPU tier
(crates/frankensearch-fusion/src/sync_searcher.rs:616-943; fsfs shape: crates/frankensearch-fsfs/src/runtime.rs:8185-8355).

  phase 0 / "Initial":  fast embed (potion Model2Vec) -> fast-tier scan (fetch = k * candidate_multiplier) ->
                        RRF with the lexical list
  phase 1 / "Refined":  quality embed (MiniLM) -> quality-tier scan (the `Retrieved` pool, sync_searcher.rs:810-813)
                        -> blend_two_tier(fast, quality, quality_weight) -> RRF with the lexical list again
Lexical (BM25) search is the caller's: it stays on the CPU in the reference and is passed in as a ranked list.
Defaults follow TwoTierConfig (crates/frankensearch-core/src/config.rs:169-176)."""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import fusion


@dataclass
class TwoTierConfig:
    quality_weight: float = 0.7
    rrf_k: float = 60.0
    candidate_multiplier: int = 3
    # 0 = exact f16 scan of the fast tier; n = search_top_k_int8_two_pass(query, fetch, n) — the reference's default with
    # n = FAST_TIER_MULT = 3 (crates/frankensearch-index/src/two_tier.rs:1318-1337; sync_searcher::search_fast_hits)
    fast_tier_int8_multiplier: int = 0


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
