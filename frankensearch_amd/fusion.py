"""Host-side rank fusion — mirror of `frankensearch_fusion::{rrf_fuse, blend_two_tier}`
(crates/frankensearch-fusion/src/rrf.rs:368, blend.rs:107) over the C ABI (CPU code inside libfsgpu.so)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

from . import _lib
from .errors import check

DEFAULT_RRF_K = 60.0


class _ScoredDoc(C.Structure):
    _fields_ = [("doc_id", C.c_char_p), ("doc_id_len", C.c_uint32), ("score", C.c_float), ("index", C.c_uint32)]


class _FusedHit(C.Structure):
    _fields_ = [("doc_id", C.c_void_p), ("doc_id_len", C.c_uint32), ("rrf_score", C.c_double),
                ("lexical_rank", C.c_int64), ("semantic_rank", C.c_int64), ("semantic_index", C.c_uint32),
                ("lexical_score", C.c_float), ("semantic_score", C.c_float), ("in_both_sources", C.c_uint8)]


@dataclass
class FusedHit:
    """crates/frankensearch-core/src/types.rs:3892-3925"""
    doc_id: str
    rrf_score: float
    lexical_rank: Optional[int]
    semantic_rank: Optional[int]
    semantic_index: Optional[int]
    lexical_score: Optional[float]
    semantic_score: Optional[float]
    in_both_sources: bool


def _pack(hits: Sequence[Tuple]) -> Tuple["C.Array", list]:
    keep = [h[0].encode() for h in hits]
    arr = (_ScoredDoc * max(len(hits), 1))()
    for i, h in enumerate(hits):
        arr[i] = _ScoredDoc(keep[i], len(keep[i]), float(h[1]), int(h[2]) if len(h) > 2 else 0)
    return arr, keep


def rrf_fuse(lexical: Sequence[Tuple[str, float]], semantic: Sequence[Tuple[str, float, int]], limit: int,
             offset: int = 0, k: float = DEFAULT_RRF_K, lexical_weight: float = 1.0, semantic_weight: float = 1.0,
             tiebreak: str = "lexical_then_id") -> List[FusedHit]:
    la, lk = _pack(lexical)
    sa, sk = _pack(semantic)
    out = (_FusedHit * max(limit, 1))()
    n = C.c_uint32()
    check(_lib.lib().fsgpu_rrf_fuse(la, len(lexical), sa, len(semantic), k, lexical_weight, semantic_weight,
                                    1 if tiebreak == "hash" else 0, limit, offset, out, C.byref(n)))
    res = []
    for i in range(n.value):
        h = out[i]
        res.append(FusedHit(C.string_at(h.doc_id, h.doc_id_len).decode(), h.rrf_score,
                            h.lexical_rank if h.lexical_rank >= 0 else None,
                            h.semantic_rank if h.semantic_rank >= 0 else None,
                            h.semantic_index if h.semantic_index != 0xFFFFFFFF else None,
                            h.lexical_score if h.lexical_rank >= 0 else None,
                            h.semantic_score if h.semantic_rank >= 0 else None, bool(h.in_both_sources)))
    return res


def blend_two_tier(fast: Sequence[Tuple[str, float, int]], quality: Sequence[Tuple[str, float, int]],
                   blend_factor: float) -> List[Tuple[str, float, int]]:
    fa, fk = _pack(fast)
    qa, qk = _pack(quality)
    out = (_ScoredDoc * max(len(fast) + len(quality), 1))()
    n = C.c_uint32()
    check(_lib.lib().fsgpu_blend_two_tier(fa, len(fast), qa, len(quality), blend_factor, out, C.byref(n)))
    return [(C.string_at(out[i].doc_id, out[i].doc_id_len).decode(), out[i].score, out[i].index) for i in range(n.value)]


def blend_two_tier_aligned(fast: Sequence[Tuple[str, float, int]], quality_scores: Sequence[Optional[float]],
                           blend_factor: float) -> List[Tuple[str, float, int]]:
    """blend_two_tier_aligned (blend.rs:213-294): quality_scores[i] is the optional quality score of fast[i]."""
    import numpy as np
    fa, fk = _pack(fast)
    # a position past the end of quality_scores is None (quality_scores.get(i), blend.rs:246); extra scores have no hit to belong to
    quality_scores = (list(quality_scores) + [None] * len(fast))[:len(fast)]
    qs = np.array([0.0 if q is None else q for q in quality_scores], dtype=np.float32)
    qp = np.array([q is not None for q in quality_scores], dtype=np.uint8)
    out = (_ScoredDoc * max(len(fast), 1))()
    n = C.c_uint32()
    check(_lib.lib().fsgpu_blend_two_tier_aligned(fa, len(fast), qs.ctypes.data if len(fast) else None,
                                                  qp.ctypes.data if len(fast) else None, blend_factor, out, C.byref(n)))
    return [(C.string_at(out[i].doc_id, out[i].doc_id_len).decode(), out[i].score, out[i].index) for i in range(n.value)]
