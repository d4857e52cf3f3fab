"""CPU oracle of the host-side fusion steps (pure Python; inputs are a few hundred hits).  TEST INFRASTRUCTURE ONLY.

rrf_fuse        crates/frankensearch-fusion/src/rrf.rs:113-122 (rank_contribution), :85-98,122-138 (sanitisers),
                :179-198 (cmp_for_ranking), :368-560 (fusion, window select, sort, offset/limit)
blend_two_tier  crates/frankensearch-fusion/src/blend.rs:24-75 (NormBounds), :107-195 (blend), :518-532 (sanitisers)
Pinned on the reference's known-answer tests (tests/test_oracle_fusion.py)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

DEFAULT_RRF_K = 60.0
F = np.float32


@dataclass
class FusedHit:
    doc_id: str
    rrf_score: float
    lexical_rank: Optional[int] = None
    semantic_rank: Optional[int] = None
    semantic_index: Optional[int] = None
    lexical_score: Optional[float] = None
    semantic_score: Optional[float] = None
    in_both_sources: bool = False


def _total_order_key(x: float) -> int:
    """f32::total_cmp as an integer key."""
    b = int(np.float32(x).view(np.int32))
    return b ^ ((b >> 31) & 0x7FFFFFFF)


def _total_order_key64(x: float) -> int:
    b = int(np.float64(x).view(np.int64))
    return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFF)


def _fnv(doc_id: str) -> int:
    h = 0xCBF29CE484222325
    for c in doc_id.encode():
        h = ((h ^ c) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def rrf_fuse(lexical: Sequence[Tuple[str, float]], semantic: Sequence[Tuple[str, float, int]], limit: int,
             offset: int = 0, k: float = DEFAULT_RRF_K, lexical_weight: float = 1.0, semantic_weight: float = 1.0,
             tiebreak: str = "lexical_then_id") -> List[FusedHit]:
    """lexical: [(doc_id, bm25 score)], semantic: [(doc_id, score, index)], both best-first."""
    k = k if (math.isfinite(k) and k >= 0.0) else DEFAULT_RRF_K
    lw = lexical_weight if (math.isfinite(lexical_weight) and lexical_weight > 0.0) else 1.0
    sw = semantic_weight if (math.isfinite(semantic_weight) and semantic_weight > 0.0) else 1.0
    hits = {}
    for rank, (doc, score) in enumerate(lexical):
        c = (1.0 / (k + float(min(rank, 0xFFFFFFFF)) + 1.0)) * lw
        h = hits.get(doc)
        if h is None:
            hits[doc] = FusedHit(doc, c, lexical_rank=rank, lexical_score=float(np.float32(score)))
        elif h.lexical_rank is None:
            h.rrf_score += c
            h.lexical_rank, h.lexical_score = rank, float(np.float32(score))
            if h.semantic_rank is not None:
                h.in_both_sources = True
    for rank, (doc, score, index) in enumerate(semantic):
        c = (1.0 / (k + float(min(rank, 0xFFFFFFFF)) + 1.0)) * sw
        h = hits.get(doc)
        if h is None:
            hits[doc] = FusedHit(doc, c, semantic_rank=rank, semantic_score=float(np.float32(score)), semantic_index=index)
        elif h.semantic_rank is None:
            h.rrf_score += c
            h.semantic_rank, h.semantic_score, h.semantic_index = rank, float(np.float32(score)), index
            if h.lexical_rank is not None:
                h.in_both_sources = True
    results = list(hits.values())
    window = limit + offset
    if window == 0:
        return []

    def sort_key(h: FusedHit):
        base = (-_total_order_key64(h.rrf_score), 0 if h.in_both_sources else 1)
        if tiebreak == "hash":
            return base + (_fnv(h.doc_id), h.doc_id.encode())
        lex = h.lexical_score if h.lexical_score is not None else -math.inf
        return base + (-_total_order_key(lex), h.doc_id.encode())

    results.sort(key=sort_key)
    return results[offset:offset + limit]


def blend_two_tier(fast: Sequence[Tuple[str, float, int]], quality: Sequence[Tuple[str, float, int]],
                   blend_factor: float) -> List[Tuple[str, float, int]]:
    """fast/quality: [(doc_id, score, index)] best-first -> [(doc_id, blended score, index)] best-first."""
    alpha = F(min(max(blend_factor, 0.0), 1.0)) if math.isfinite(blend_factor) else F(0.7)

    def bounds(hits):
        mn, mx, saw = F(np.inf), F(-np.inf), False
        for _, s, _ in hits:
            s = F(s)
            if np.isfinite(s):
                mn, mx, saw = min(mn, s), max(mx, s), True
        with np.errstate(invalid="ignore"):
            return mn, F(mx - mn), saw

    def apply(b, s):
        mn, rng, saw = b
        s = F(s)
        if not saw or not np.isfinite(s):
            return F(0.0)
        v = F(F(s - mn) / rng) if rng > F(1.1920929e-7) else F(1.0)
        return F(min(max(v, F(0.0)), F(1.0)))

    fb, qb = bounds(fast), bounds(quality)
    merged = {}
    for doc, s, index in fast:
        e = merged.setdefault(doc, {"fast": None, "quality": None, "index": index})
        if e["fast"] is None:
            e["fast"], e["index"] = apply(fb, s), index
    for doc, s, index in quality:
        e = merged.setdefault(doc, {"fast": None, "quality": None, "index": index})
        if e["quality"] is None:
            e["quality"] = apply(qb, s)
    out = []
    for doc, e in merged.items():
        f, q = e["fast"], e["quality"]
        if f is not None and q is not None:
            # alpha.mul_add(q, (1 - alpha) * f): the f32 product term, then one fused rounding
            t = F(F(F(1.0) - alpha) * f)
            score = F(np.float64(alpha) * np.float64(q) + np.float64(t))
        elif f is not None:
            score = f
        elif q is not None:
            score = q
        else:
            score = F(0.0)
        score = score if np.isfinite(score) else F(0.0)
        out.append((doc, float(score), e["index"]))
    out.sort(key=lambda h: (-_total_order_key(h[1]), h[0].encode()))
    return out
