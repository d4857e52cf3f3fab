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


def blend_two_tier_aligned(fast: Sequence[Tuple[str, float, int]], quality_scores: Sequence[Optional[float]],
                           blend_factor: float) -> List[Tuple[str, float, int]]:
    """blend_two_tier_aligned (blend.rs:213-294): quality_scores[i] is the Option<f32> quality score of fast[i].  The reference
    documents it as bit-identical to blend_two_tier(fast, Some-filtered projection of fast): that is how it is restated."""
    subset = [(doc, q, index) for (doc, _, index), q in zip(fast, quality_scores) if q is not None]
    return blend_two_tier(fast, subset, blend_factor)


def quality_alignment(fast_records: Sequence[Tuple[int, str, bool]], quality_records: Sequence[Tuple[int, str, bool]]):
    """The merge walk of TwoTierIndex::assemble_opened (two_tier.rs:750-866).  records: [(doc_id_hash, doc_id, tombstoned)] in
    table order (sorted by (hash, doc_id)).  Returns ("aligned", None) or ("mapping", [quality row or None per fast row])."""
    kind, mapping = "aligned", None
    f = q = 0
    fc, qc = len(fast_records), len(quality_records)

    def ensure_mapping(upto):
        nonlocal kind, mapping
        if kind == "aligned":
            kind, mapping = "mapping", list(range(upto))

    while f < fc and q < qc:
        fh, fd, ft = fast_records[f]
        qh, qd, qt = quality_records[q]
        if ft:
            ensure_mapping(f)
            mapping.append(None)
            f += 1
            continue
        if qt:
            q += 1
            continue
        if kind == "aligned" and f != q:
            ensure_mapping(f)
        if fh < qh or (fh == qh and fd.encode() < qd.encode()):
            ensure_mapping(f)
            mapping.append(None)
            f += 1
        elif fh > qh or (fh == qh and fd.encode() > qd.encode()):
            q += 1
        else:
            if kind == "mapping":
                mapping.append(q)
            f += 1
            q += 1
    if f < fc:
        ensure_mapping(f)
        mapping.extend([None] * (fc - len(mapping)))
    return kind, mapping


def quality_scores_for_hits(hits: Sequence[Tuple[str, float, int]], alignment, fast_count: int, quality_dot,
                            quality_find=None, fast_find=None, quality_wal=None) -> List[Optional[float]]:
    """TwoTierIndex::quality_scores_for_hits (two_tier.rs:1566-1631).  hits: [(doc_id, score, fast index)];
    alignment = quality_alignment(...) result; quality_dot(row) = dot_query_at; quality_find / fast_find(doc_id) = the index's
    find_index_by_doc_id; quality_wal(doc_id) = score of the quality WAL's latest entry of that doc or None."""
    kind, mapping = alignment
    out = []
    for doc, _, index in hits:
        score = quality_wal(doc) if quality_wal else None
        if score is None:
            fast_idx = None
            if index == 0xFFFFFFFF:
                fast_idx = fast_find(doc) if fast_find else None
            elif index < fast_count:
                fast_idx = index
            if fast_idx is not None and fast_idx < fast_count:
                qrow = fast_idx if kind == "aligned" else (mapping[fast_idx] if fast_idx < len(mapping) else None)
                if qrow is not None:
                    score = quality_dot(qrow)
        if score is None and quality_find:
            qrow = quality_find(doc)
            if qrow is not None:
                score = quality_dot(qrow)
        out.append(score)
    return out
