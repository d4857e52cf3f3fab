"""Product host fusion (C++ in libfsgpu.so, through the C ABI) vs the Python oracle; no GPU needed."""
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def fusion():
    from frankensearch_amd.build import build
    build()
    from frankensearch_amd import fusion as f
    return f


def test_rrf_matches_oracle_randomised(fusion):
    from oracle import fusion_oracle as fo
    rng = np.random.default_rng(5)
    for trial in range(60):
        nl, ns = int(rng.integers(0, 60)), int(rng.integers(0, 60))
        pool = [f"doc-{i:03}" for i in range(80)]
        lex = [(str(rng.choice(pool)), float(np.float32(rng.normal()))) for _ in range(nl)]
        sem = [(str(rng.choice(pool)), float(np.float32(rng.normal())), int(rng.integers(0, 1000))) for _ in range(ns)]
        lex.sort(key=lambda h: -h[1]); sem.sort(key=lambda h: -h[1])
        kw = dict(k=float(rng.choice([60.0, 1.0, 0.0, -5.0, math.nan])), lexical_weight=float(rng.choice([1.0, 2.5, -1.0])),
                  semantic_weight=float(rng.choice([1.0, 0.3, math.inf])), tiebreak=str(rng.choice(["lexical_then_id", "hash"])))
        limit, offset = int(rng.integers(0, 40)), int(rng.integers(0, 5))
        got = fusion.rrf_fuse(lex, sem, limit, offset, **kw)
        want = fo.rrf_fuse(lex, sem, limit, offset, **kw)
        assert [h.doc_id for h in got] == [h.doc_id for h in want], trial
        for g, w in zip(got, want):
            assert g.rrf_score == w.rrf_score and g.in_both_sources == w.in_both_sources
            assert (g.lexical_rank, g.semantic_rank, g.semantic_index) == (w.lexical_rank, w.semantic_rank, w.semantic_index)
            assert g.lexical_score == w.lexical_score and g.semantic_score == w.semantic_score


def test_rrf_reference_known_answers(fusion):
    # rrf.rs:1864-1964, 2173-2214
    r = fusion.rrf_fuse([("doc-a", 10.0)], [], 10)
    assert abs(r[0].rrf_score - 1 / 61.0) < 1e-12
    r = fusion.rrf_fuse([("shared", 5.0)], [("shared", 0.9, 3)], 10)
    assert abs(r[0].rrf_score - 2 / 61.0) < 1e-12 and r[0].in_both_sources and r[0].semantic_index == 3
    assert [h.doc_id for h in fusion.rrf_fuse([("only-lex", 10.0)], [("only-sem", 0.9, 0)], 10)] == ["only-lex", "only-sem"]
    assert fusion.rrf_fuse([], [], 10) == []


def test_blend_matches_oracle_randomised(fusion):
    from oracle import fusion_oracle as fo
    rng = np.random.default_rng(8)
    for trial in range(60):
        nf, nqual = int(rng.integers(0, 50)), int(rng.integers(0, 50))
        pool = [f"d{i:03}" for i in range(70)]
        fast = [(str(rng.choice(pool)), float(np.float32(rng.normal())), int(rng.integers(0, 500))) for _ in range(nf)]
        qual = [(str(rng.choice(pool)), float(np.float32(rng.normal())), int(rng.integers(0, 500))) for _ in range(nqual)]
        if trial % 7 == 0 and fast:
            fast[0] = (fast[0][0], math.nan, fast[0][2])
        fast.sort(key=lambda h: -(h[1] if math.isfinite(h[1]) else -1e9)); qual.sort(key=lambda h: -h[1])
        alpha = float(rng.choice([0.7, 0.0, 1.0, 0.5, math.nan, 3.0]))
        got = fusion.blend_two_tier(fast, qual, alpha)
        want = fo.blend_two_tier(fast, qual, alpha)
        assert len(got) == len(want)
        gs, ws = dict((d, s) for d, s, _ in got), dict((d, s) for d, s, _ in want)
        assert set(gs) == set(ws)
        assert all(abs(gs[d] - ws[d]) <= 1.2e-7 for d in gs)           # fused multiply-add emulation: <= 1 ulp
        assert dict((d, i) for d, _, i in got) == dict((d, i) for d, _, i in want)
        assert all(a[1] >= b[1] for a, b in zip(got, got[1:]))


def test_blend_reference_known_answers(fusion):
    # blend.rs:708-897
    fast = [("a", 1.0, 0), ("b", 0.0, 1), ("c", 2.0, 2)]
    quality = [("a", 2.0, 0), ("b", 0.0, 1), ("c", 1.0, 2)]
    sc = dict((d, s) for d, s, _ in fusion.blend_two_tier(fast, quality, 0.7))
    assert abs(sc["a"] - 0.85) <= 1e-6
    b = fusion.blend_two_tier([("a", 10.0, 0), ("b", 1.0, 1)], [("a", 1.0, 0), ("b", 10.0, 1)], 0.7)
    assert [d for d, _, _ in b] == ["b", "a"]
    assert fusion.blend_two_tier([], [], 0.7) == []


def test_aligned_blend_matches_oracle_and_reference_known_answers(fusion):
    """fsgpu_blend_two_tier_aligned (blend.rs:213-294) vs the oracle: the reference's own cases (duplicate doc id, None scores,
    NaN blend factor: blend.rs:578-646) and randomised pools."""
    from oracle import fusion_oracle as fo
    fast = [("a", 0.90, 0), ("b", 0.70, 1), ("c", 0.50, 2), ("d", 0.30, 3), ("a", 0.20, 9), ("e", 0.10, 4)]
    scores = [0.10, None, 0.95, 0.40, 0.99, None]
    for alpha in (0.0, 0.3, 0.7, 1.0, math.nan):
        got, want = fusion.blend_two_tier_aligned(fast, scores, alpha), fo.blend_two_tier_aligned(fast, scores, alpha)
        assert [(d, i) for d, _, i in got] == [(d, i) for d, _, i in want]
        assert [np.float32(s).view(np.uint32) for _, s, _ in got] == [np.float32(s).view(np.uint32) for _, s, _ in want]
        # ... which the reference proves equal to the materialised path
        subset = [(d, q, i) for (d, _, i), q in zip(fast, scores) if q is not None]
        mat = fusion.blend_two_tier(fast, subset, alpha)
        assert [(d, np.float32(s).view(np.uint32), i) for d, s, i in got] == [(d, np.float32(s).view(np.uint32), i) for d, s, i in mat]
    assert fusion.blend_two_tier_aligned([], [], 0.7) == []
    rng = np.random.default_rng(21)
    for trial in range(60):
        n = int(rng.integers(1, 90))
        pool = [f"d{i:03}" for i in range(60 if trial % 3 else 400)]
        fast = sorted(((str(rng.choice(pool)), float(np.float32(rng.normal())), int(rng.integers(0, 900))) for _ in range(n)),
                      key=lambda h: -h[1])
        scores = [None if rng.random() < 0.3 else float(np.float32(rng.normal())) for _ in range(n)]
        if trial % 5 == 0:
            scores[0] = math.nan
        if trial % 11 == 0:
            scores = [None] * n
        alpha = float(rng.choice([0.7, 0.0, 1.0, 0.35, math.nan, -2.0]))
        got, want = fusion.blend_two_tier_aligned(fast, scores, alpha), fo.blend_two_tier_aligned(fast, scores, alpha)
        assert [(d, np.float32(s).view(np.uint32), i) for d, s, i in got] == \
               [(d, np.float32(s).view(np.uint32), i) for d, s, i in want], trial
