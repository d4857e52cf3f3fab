"""Pins oracle/fusion_oracle.py on the reference's RRF / blend known-answer tests
(crates/frankensearch-fusion/src/rrf.rs:1864-2270, blend.rs:708-897)."""
import math

from oracle import fusion_oracle as fo

EPS = 1e-6


def lex(doc, s):
    return (doc, s)


def sem(doc, s, index=0):
    return (doc, s, index)


def test_rrf_score_formulas():
    r = fo.rrf_fuse([lex("doc-a", 10.0)], [], 10)                       # rrf.rs:1864 k=60
    assert len(r) == 1 and abs(r[0].rrf_score - 1 / 61.0) < 1e-12
    r = fo.rrf_fuse([], [sem("first", 0.9), sem("second", 0.8)], 10, k=1.0)   # :1881
    assert abs(r[0].rrf_score - 0.5) < 1e-12 and abs(r[1].rrf_score - 1 / 3) < 1e-12
    r = fo.rrf_fuse([lex("doc-a", 10.0)], [], 10, k=0.0)                # :1898
    assert abs(r[0].rrf_score - 1.0) < 1e-12
    for bad in (math.nan, math.inf, -1.0, -100.0):                      # :1912
        assert abs(fo.rrf_fuse([lex("doc-a", 10.0)], [], 10, k=bad)[0].rrf_score - 1 / 61.0) < 1e-12


def test_rrf_multi_source_and_tiebreaks():
    r = fo.rrf_fuse([lex("shared", 5.0)], [sem("shared", 0.9)], 10)      # :1933
    assert len(r) == 1 and abs(r[0].rrf_score - 2 / 61.0) < 1e-12 and r[0].in_both_sources
    assert r[0].lexical_rank == 0 and r[0].semantic_rank == 0
    r = fo.rrf_fuse([lex("shared", 5.0), lex("lex-only", 4.0)], [sem("shared", 0.9), sem("sem-only", 0.8)], 10)
    assert len(r) == 3 and r[0].doc_id == "shared" and r[0].in_both_sources     # :1953
    r = fo.rrf_fuse([lex("only-lex", 10.0)], [sem("only-sem", 0.9)], 10)         # :2173
    assert [h.doc_id for h in r] == ["only-lex", "only-sem"]
    r = fo.rrf_fuse([lex("alpha", 10.0)], [sem("beta", 0.9)], 10)                # :2196
    assert [h.doc_id for h in r] == ["alpha", "beta"]
    r = fo.rrf_fuse([], [], 10)                                                  # :2114
    assert r == []


def test_rrf_limit_offset_and_preserved_scores():
    lexical = [lex(f"d{i}", 10.0 - i) for i in range(6)]
    r = fo.rrf_fuse(lexical, [], 3)                                   # :2122
    assert [h.doc_id for h in r] == ["d0", "d1", "d2"]
    r = fo.rrf_fuse(lexical, [], 10, offset=2)                        # :2138
    assert [h.doc_id for h in r] == ["d2", "d3", "d4", "d5"]
    r = fo.rrf_fuse(lexical, [], 2, offset=1)                         # :2154
    assert [h.doc_id for h in r] == ["d1", "d2"]
    r = fo.rrf_fuse([lex("s", 7.5)], [sem("s", 0.25, 42)], 10)         # :2233-2268
    assert r[0].lexical_score == 7.5 and r[0].semantic_score == 0.25 and r[0].semantic_index == 42
    big = fo.rrf_fuse([lex(f"l{i}", 1.0) for i in range(30)], [sem(f"l{i*2}", 0.5) for i in range(20)], 50)
    assert all(a.rrf_score >= b.rrf_score for a, b in zip(big, big[1:]))      # :2271
    # duplicate doc ids inside one lane: first (best) occurrence wins (rrf.rs:395-403)
    r = fo.rrf_fuse([lex("a", 3.0), lex("a", 2.0), lex("b", 1.0)], [], 10)
    assert [h.doc_id for h in r] == ["a", "b"] and r[1].lexical_rank == 2


def test_rrf_tier_weights():
    # rrf.rs:2030-2074: up-weighting the semantic tier flips the order of two single-source docs
    r = fo.rrf_fuse([lex("L", 1.0)], [sem("S", 0.5)], 10, semantic_weight=2.0)
    assert [h.doc_id for h in r] == ["S", "L"] and abs(r[0].rrf_score - 2 / 61.0) < 1e-12
    r = fo.rrf_fuse([lex("L", 1.0)], [sem("S", 0.5)], 10, semantic_weight=-3.0)   # bad weight -> 1.0
    assert abs(r[1].rrf_score - 1 / 61.0) < 1e-12


def score_for(doc, blended):
    return next(s for d, s, _ in blended if d == doc)


def test_blend_known_answers():
    fast = [sem("a", 1.0, 0), sem("b", 0.0, 1), sem("c", 2.0, 2)]
    quality = [sem("a", 2.0, 0), sem("b", 0.0, 1), sem("c", 1.0, 2)]
    assert abs(score_for("a", fo.blend_two_tier(fast, quality, 0.7)) - 0.85) <= EPS          # blend.rs:708
    fast = [sem("a", 10.0, 0), sem("b", 0.0, 1)]
    quality = [sem("a", 5.0, 0), sem("b", 15.0, 1)]
    b = fo.blend_two_tier(fast, quality, 1.0)                                                 # :724
    assert abs(score_for("a", b)) <= EPS and abs(score_for("b", b) - 1.0) <= EPS
    b = fo.blend_two_tier(fast, quality, 0.0)                                                 # :734
    assert abs(score_for("a", b) - 1.0) <= EPS and abs(score_for("b", b)) <= EPS
    b = fo.blend_two_tier([sem("fast-only", 10.0, 0)], [sem("quality-only", 10.0, 1)], 0.7)   # :744
    assert abs(score_for("fast-only", b) - 1.0) <= EPS and abs(score_for("quality-only", b) - 1.0) <= EPS
    b = fo.blend_two_tier([sem("same", 1.0, 0), sem("other", 1.0, 1)], [sem("same", 2.0, 0), sem("other", 2.0, 1)], 0.7)
    assert abs(score_for("same", b) - 1.0) <= EPS                                             # :759
    b = fo.blend_two_tier([sem("nan-doc", math.nan, 0), sem("ok-doc", 1.0, 1)], [], 0.3)      # :768
    assert all(math.isfinite(s) for _, s, _ in b)
    b = fo.blend_two_tier([sem("a", 10.0, 0), sem("b", 1.0, 1)], [sem("a", 1.0, 0), sem("b", 10.0, 1)], 0.7)
    assert [d for d, _, _ in b] == ["b", "a"]                                                 # :777
    for scores in ([-0.88, -0.89, -0.90], [1.02, 1.01, 1.00]):                                # :787
        b = fo.blend_two_tier([sem("z-best", scores[0], 0), sem("a-middle", scores[1], 1), sem("m-worst", scores[2], 2)], [], 0.7)
        assert [d for d, _, _ in b] == ["z-best", "a-middle", "m-worst"]
    b = fo.blend_two_tier([sem("a", 10.0, 0), sem("b", 0.0, 1)], [sem("a", 0.0, 0), sem("b", 10.0, 1)], 0.5)
    assert abs(score_for("a", b) - score_for("b", b)) <= EPS                                  # :873
    bn = fo.blend_two_tier([sem("a", 1.0, 0)], [sem("a", 1.0, 0)], math.nan)                  # :886
    bd = fo.blend_two_tier([sem("a", 1.0, 0)], [sem("a", 1.0, 0)], 0.7)
    assert abs(bn[0][1] - bd[0][1]) <= EPS
    assert fo.blend_two_tier([], [], 0.7) == []                                               # :851


def hit(doc, s, index):
    return (doc, s, index)


def test_aligned_blend_known_answers():
    # blend.rs:578-629: blend_two_tier_aligned == blend_two_tier(fast, Some-filtered projection), duplicate doc id included
    fast = [hit("a", 0.90, 0), hit("b", 0.70, 1), hit("c", 0.50, 2), hit("d", 0.30, 3), hit("a", 0.20, 9), hit("e", 0.10, 4)]
    scores = [0.10, None, 0.95, 0.40, 0.99, None]
    for alpha in (0.0, 0.3, 0.7, 1.0, math.nan):
        b = fo.blend_two_tier_aligned(fast, scores, alpha)
        assert len(b) == 5 and sorted(d for d, _, _ in b) == ["a", "b", "c", "d", "e"]
        assert next(i for d, _, i in b if d == "a") == 0            # first occurrence's index
    # quality only where present; the bounds run over ALL present scores — the ignored duplicate's 0.99 included
    b = fo.blend_two_tier_aligned(fast, scores, 1.0)
    assert b[0][0] == "c" and abs(b[0][1] - (0.95 - 0.10) / (0.99 - 0.10)) <= EPS
    # blend.rs:631-646: no quality scores -> the fast-only blend; empty inputs
    fast2 = [hit("a", 0.9, 0), hit("b", 0.1, 1)]
    assert fo.blend_two_tier_aligned(fast2, [None, None], 0.7) == fo.blend_two_tier(fast2, [], 0.7)
    assert fo.blend_two_tier_aligned([], [], 0.7) == []


def _records(ids, tomb=()):
    recs = sorted(((fo._fnv(d), d) for d in ids), key=lambda r: (r[0], r[1].encode()))
    return [(h, d, d in tomb) for h, d in recs]


def test_quality_alignment_known_answers():
    # two_tier.rs:3337-3398 (quality_alignment_handles_partial_coverage): the quality tier omits doc-b
    fast = _records(["doc-a", "doc-b", "doc-c"])
    qual = _records(["doc-c", "doc-a"])
    kind, mapping = fo.quality_alignment(fast, qual)
    qrow = {d: i for i, (_, d, _) in enumerate(qual)}
    for i, (_, d, _) in enumerate(fast):
        got = i if kind == "aligned" else mapping[i]
        assert got == qrow.get(d), (d, kind, mapping)
    dots = {qrow["doc-a"]: 1.0, qrow["doc-c"]: 0.0}
    hits = [(d, 0.0, i) for i, (_, d, _) in enumerate(fast)]
    scores = fo.quality_scores_for_hits(hits, (kind, mapping), len(fast), lambda r: dots[r])
    by_doc = {d: s for (d, _, _), s in zip(hits, scores)}
    assert by_doc == {"doc-a": 1.0, "doc-b": None, "doc-c": 0.0}
    # two_tier.rs:5630-5672 (full coverage): identical id sets stay ALIGNED
    kind, mapping = fo.quality_alignment(_records(["doc-a", "doc-b"]), _records(["doc-a", "doc-b"]))
    assert kind == "aligned" and mapping is None
    # tombstoned rows on either side are skipped; a tombstoned fast row has no quality row
    fast = _records([f"d{i}" for i in range(8)], tomb={"d3"})
    qual = _records([f"d{i}" for i in range(8)], tomb={"d5"})
    kind, mapping = fo.quality_alignment(fast, qual)
    assert kind == "mapping"
    for i, (_, d, t) in enumerate(fast):
        want = None if (t or d == "d5") else next(j for j, (_, qd, _) in enumerate(qual) if qd == d)
        assert mapping[i] == want, (d, mapping)
