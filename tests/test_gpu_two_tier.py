"""End-to-end two-tier query flow on the GPU vs the same flow assembled from the oracles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_tier_flow_matches_oracle_pipeline(oracle):
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.two_tier import SyncTwoTierSearcher, TwoTierConfig
    from oracle import bert_oracle, fusion_oracle

    build()
    rng = np.random.default_rng(3)
    n = 20000
    fast_slab = rng.standard_normal((n, 256)).astype(np.float16).view(np.uint16)
    qual_slab = rng.standard_normal((n, 384)).astype(np.float16).view(np.uint16)
    table = rng.standard_normal((5000, 256)).astype(np.float32)
    w = bert_oracle.random_weights(5, 3000, 384, 6, 1536)
    doc = lambda r: f"doc-{r:08d}"
    s = SyncTwoTierSearcher(fa.VectorIndex.from_slab(fast_slab), fa.VectorIndex.from_slab(qual_slab),
                            fa.Model2VecEmbedder(table), fa.NativeEmbedder(w), doc, TwoTierConfig())
    for trial in range(5):
        fast_ids = rng.integers(0, 5000, 9).tolist()
        qual_ids = [101] + rng.integers(1000, 3000, 10).tolist() + [102]
        lexical = [(doc(int(r)), float(30 - i)) for i, r in enumerate(rng.choice(n, 25, replace=False))]
        k = 10
        out = s.search(fast_ids, qual_ids, k, lexical)
        # oracle pipeline; the quality query vector is taken from the GPU encoder (its own tolerance test is separate)
        fv = oracle.m2v_embed(table, fast_ids)
        fr, fs = oracle.search_top_k(fast_slab, fv, 30)
        fast_hits = [(doc(int(r)), float(x), int(r)) for r, x in zip(fr, fs)]
        assert out.fast_hits == fast_hits
        qv = s.quality_embedder.embed_token_ids(qual_ids)
        assert np.sum(qv * bert_oracle.embed_forward(w, [qual_ids], 6)[0]) > 0.999
        qr, qs = oracle.search_top_k(qual_slab, qv, 30)
        qual_hits = [(doc(int(r)), float(x), int(r)) for r, x in zip(qr, qs)]
        assert out.quality_hits == qual_hits
        want_initial = fusion_oracle.rrf_fuse(lexical, fast_hits, k)
        assert [h.doc_id for h in out.initial_results] == [h.doc_id for h in want_initial]
        blended = fusion_oracle.blend_two_tier(fast_hits, qual_hits, 0.7)
        fidx = {d: i for d, _, i in fast_hits}
        blended = [(d, sc, fidx.get(d, 0xFFFFFFFF)) for d, sc, _ in blended]
        want_final = fusion_oracle.rrf_fuse(lexical, blended, k)
        assert [h.doc_id for h in out.final_results] == [h.doc_id for h in want_final]
        assert [h.rrf_score for h in out.final_results] == [h.rrf_score for h in want_final]
        assert out.metrics.phase2_total_ms > 0 and out.metrics.phase1_total_ms > 0


def _small_two_tier(fa, rng, n=40000):
    from frankensearch_amd.synthetic import random_bert_weights
    fast_slab = rng.standard_normal((n, 256)).astype(np.float16).view(np.uint16)
    qual_slab = rng.standard_normal((n, 384)).astype(np.float16).view(np.uint16)
    table = rng.standard_normal((5000, 256)).astype(np.float32)
    w = random_bert_weights(5, 3000, 384, 6, 1536)
    return (fa.VectorIndex.from_slab(fast_slab), fa.VectorIndex.from_slab(qual_slab), fa.Model2VecEmbedder(table),
            fa.NativeEmbedder(w))


def test_native_host_searcher_equals_python_mirror():
    # libfshost.so (C++ over the C ABI, native threads) must deliver exactly what the Python mirror does
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.host import NativeTwoTierSearcher
    from frankensearch_amd.two_tier import SyncTwoTierSearcher, TwoTierConfig

    build()
    rng = np.random.default_rng(11)
    n = 40000
    fast, qual, m2v, bert = _small_two_tier(fa, rng, n)
    doc = lambda r: f"doc-{r:08d}"
    py = SyncTwoTierSearcher(fast, qual, m2v, bert, doc, TwoTierConfig())
    native = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1)
    for trial in range(6):
        fast_ids = rng.integers(0, 5000, int(rng.integers(1, 20))).tolist()
        qual_ids = [101] + rng.integers(1000, 3000, int(rng.integers(2, 25))).tolist() + [102]
        lexical = [(doc(int(r)), float(30 - i)) for i, r in enumerate(rng.choice(n, 30, replace=False))]
        if trial == 5:
            lexical = []
        want = py.search(fast_ids, qual_ids, 10, lexical)
        ini, fin, metrics = native.search(fast_ids, qual_ids, 10, lexical)
        assert ini == want.initial_results
        assert fin == want.final_results
        assert metrics["phase2_total_ms"] > 0
    # the reference's default fast tier (int8 two-pass, multiplier 3) through both hosts
    py8 = SyncTwoTierSearcher(fast, qual, m2v, bert, doc, TwoTierConfig(fast_tier_int8_multiplier=3))
    native8 = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3)
    for trial in range(3):
        fast_ids = rng.integers(0, 5000, 12).tolist()
        qual_ids = [101] + rng.integers(1000, 3000, 9).tolist() + [102]
        lexical = [(doc(int(r)), float(30 - i)) for i, r in enumerate(rng.choice(n, 30, replace=False))]
        want = py8.search(fast_ids, qual_ids, 10, lexical)
        ini, fin, _ = native8.search(fast_ids, qual_ids, 10, lexical)
        assert ini == want.initial_results and fin == want.final_results
    # the quality embedding computed on a helper thread while phase 0 runs: same results; a failing embedding (a token
    # id outside the vocabulary) is reported from the helper thread with its message
    eager = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3,
                                  prefetch_quality_embed=True)
    for trial in range(4):
        fast_ids = rng.integers(0, 5000, 12).tolist()
        qual_ids = [101] + rng.integers(1000, 3000, 9).tolist() + [102]
        lexical = [(doc(int(r)), float(30 - i)) for i, r in enumerate(rng.choice(n, 30, replace=False))]
        want = py8.search(fast_ids, qual_ids, 10, lexical)
        ini, fin, _ = eager.search(fast_ids, qual_ids, 10, lexical)
        assert ini == want.initial_results and fin == want.final_results
    # ... and the quality tier's search as well (prefetch_quality_embed = 2): same results, same error reporting
    speculative = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3,
                                        prefetch_quality_embed=2)
    for trial in range(4):
        fast_ids = rng.integers(0, 5000, 12).tolist()
        qual_ids = [101] + rng.integers(1000, 3000, 9).tolist() + [102]
        lexical = [(doc(int(r)), float(30 - i)) for i, r in enumerate(rng.choice(n, 30, replace=False))]
        want = py8.search(fast_ids, qual_ids, 10, lexical)
        ini, fin, _ = speculative.search(fast_ids, qual_ids, 10, lexical)
        assert ini == want.initial_results and fin == want.final_results
    for searcher in (native8, eager, speculative):
        with pytest.raises(Exception) as err:
            searcher.search([1, 2, 3], [101, 10_000_000, 102], 10, [])
        assert str(err.value)


def test_native_host_searcher_over_fsvi_files_with_doc_id_tables(tmp_path):
    # doc ids come from the indexes' own FSVI tables (doc_id_mode 0); the files are written by the product writer
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.host import NativeTwoTierSearcher
    from frankensearch_amd.synthetic import random_bert_weights
    from frankensearch_amd.two_tier import SyncTwoTierSearcher, TwoTierConfig

    build()
    rng = np.random.default_rng(19)
    n = 3000
    ids = [f"note-{i:05d}-{'x' * (i % 4)}" for i in range(n)]
    fast_rows = rng.standard_normal((n, 256)).astype(np.float32)
    qual_rows = rng.standard_normal((n, 384)).astype(np.float32)
    pf, pq = str(tmp_path / "vector.fast.idx"), str(tmp_path / "vector.quality.idx")
    fa.write_fsvi(pf, list(zip(ids, fast_rows)), "potion", "r1")
    fa.write_fsvi(pq, list(zip(ids, qual_rows)), "minilm", "r1")
    fast, qual = fa.VectorIndex.open(pf), fa.VectorIndex.open(pq)
    m2v = fa.Model2VecEmbedder(rng.standard_normal((5000, 256)).astype(np.float32))
    bert = fa.NativeEmbedder(random_bert_weights(5, 3000, 384, 6, 1536))
    native = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=0)
    # the Python mirror resolves a tier's rows through that tier's own table, as the native searcher does
    class Py(SyncTwoTierSearcher):
        def _hits(self, index, vec, fetch, int8_multiplier=0):
            rows, scores, counts = index.search_batch(vec, fetch)
            return [(index.doc_id_at(int(rows[0, i])), float(scores[0, i]), int(rows[0, i])) for i in range(int(counts[0]))]
    py = Py(fast, qual, m2v, bert, lambda r: "", TwoTierConfig())
    for trial in range(4):
        fast_ids = rng.integers(0, 5000, 8).tolist()
        qual_ids = [101] + rng.integers(1000, 3000, 7).tolist() + [102]
        lexical = [(ids[int(r)], float(30 - i)) for i, r in enumerate(rng.choice(n, 30, replace=False))]
        want = py.search(fast_ids, qual_ids, 10, lexical)
        ini, fin, _ = native.search(fast_ids, qual_ids, 10, lexical)
        assert ini == want.initial_results and fin == want.final_results
        assert all(h.doc_id.startswith("note-") for h in fin)


def test_coalesced_concurrent_callers_get_identical_results():
    # many threads calling the per-query ABI at once are served by shared batched passes; every caller must still get
    # exactly the answer the unbatched call gives
    import threading

    import frankensearch_amd as fa
    from frankensearch_amd.build import build

    build()
    rng = np.random.default_rng(13)
    n, dim = 120_000, 384
    slab = rng.standard_normal((n, dim)).astype(np.float16).view(np.uint16)
    idx = fa.VectorIndex.from_slab(slab)
    nq = 96
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    want = [idx.search_batch(q[i], 10 if i % 3 else 7) for i in range(nq)]
    idx.set_coalescing(64, 20_000)
    got = [None] * nq
    errs = []

    def call(i):
        try:
            got[i] = idx.search_batch(q[i], 10 if i % 3 else 7)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=call, args=(i,)) for i in range(nq)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(nq):
        for a, b in zip(got[i], want[i]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), i
    batches, requests = idx.coalescing_stats()
    assert requests == nq and batches < nq // 2, (batches, requests)
    # the int8 two-pass rides the same coalescer (the reference's fast-tier default: fetch 30, multiplier 3)
    idx.set_coalescing(0, 0)
    want8 = [idx.search_top_k_int8_two_pass(q[i], 30, 3) for i in range(nq)]
    idx.set_coalescing(64, 20_000)
    got8 = [None] * nq

    def call8(i):
        try:
            got8[i] = idx.search_top_k_int8_two_pass(q[i], 30, 3)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=call8, args=(i,)) for i in range(nq)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(nq):
        assert [(h.index, np.float32(h.score).view(np.uint32)) for h in got8[i]] == \
               [(h.index, np.float32(h.score).view(np.uint32)) for h in want8[i]], i
    b2, r2 = idx.coalescing_stats()
    assert r2 == 2 * nq and b2 - batches < nq // 2
    idx.set_coalescing(0, 0)


def test_native_load_generator_runs_with_coalescing():
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.host import NativeTwoTierSearcher

    build()
    rng = np.random.default_rng(17)
    fast, qual, m2v, bert = _small_two_tier(fa, rng, 40000)
    for h, mb, wait in ((fast, 128, 300), (qual, 128, 300)):
        h.set_coalescing(mb, wait)
    m2v.set_coalescing(256, 100)
    bert.set_coalescing(256, 300)
    native = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1)
    res = native.run_load(threads=64, queries=640, warmup_queries=64, k=10, fast_vocab=5000, corpus_rows=40000,
                          quality_vocab=3000)
    assert res.completed == 640 and res.failed == 0, res.first_error
    assert res.queries_per_sec > 0 and res.phase1_p50_ms >= res.phase0_p50_ms > 0
    b, r = qual.coalescing_stats()
    assert r >= 640 and b < r


def _fsvi_pair(fa, tmp_path, rng, fast_ids, qual_ids, tag):
    fast_rows = rng.standard_normal((len(fast_ids), 64)).astype(np.float32)
    qual_rows = rng.standard_normal((len(qual_ids), 128)).astype(np.float32)
    pf, pq = str(tmp_path / f"{tag}.fast.idx"), str(tmp_path / f"{tag}.quality.idx")
    fa.write_fsvi(pf, list(zip(fast_ids, fast_rows)), "potion", "r1")
    fa.write_fsvi(pq, list(zip(qual_ids, qual_rows)), "minilm", "r1")
    return fa.VectorIndex.open(pf), fa.VectorIndex.open(pq)


def _oracle_pairing(oracle, fusion_oracle, fast, qual, fast_dead=(), qual_dead=()):
    """Alignment + the accessors quality_scores_for_hits needs, from the product indexes' own tables through the oracle."""
    def records(idx, dead):
        out = []
        for r in range(idx.record_count()):
            d = idx.doc_id_at(r)
            out.append((fusion_oracle._fnv(d), d, d in dead))
        return out
    frec, qrec = records(fast, fast_dead), records(qual, qual_dead)
    align = fusion_oracle.quality_alignment(frec, qrec)

    def find(recs):
        def f(doc):
            for i, (_, d, t) in enumerate(recs):
                if d == doc and not t:
                    return i
            return None
        return f
    return align, find(frec), find(qrec), len(frec)


def test_quality_alignment_and_rescored_fast_pool_match_the_reference_flow(oracle, tmp_path):
    """TwoTierIndex's pairing (two_tier.rs:404-409, 750-866) and quality_scores_for_hits (:1566-1631) through the C ABI, against
    the oracle's restatement: the reference's own cases first (two_tier.rs:3337-3398, 5630-5672), then a pair with missing
    documents, tombstones on both sides, a quality-side WAL entry and hits without a fast row."""
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.two_tier import TwoTierIndex
    from oracle import fusion_oracle

    build()
    rng = np.random.default_rng(41)
    # quality_alignment_handles_partial_coverage: the quality tier omits doc-b
    pf, pq = str(tmp_path / "kat.fast.idx"), str(tmp_path / "kat.quality.idx")
    fa.write_fsvi(pf, [("doc-a", [1.0, 0, 0, 0, 0, 0, 0, 0]), ("doc-b", [0, 1.0, 0, 0, 0, 0, 0, 0]), ("doc-c", [0, 0, 1.0, 0, 0, 0, 0, 0])])
    fa.write_fsvi(pq, [("doc-c", [0, 1.0, 0, 0, 0, 0, 0, 0]), ("doc-a", [1.0, 0, 0, 0, 0, 0, 0, 0])])
    fast, qual = fa.VectorIndex.open(pf), fa.VectorIndex.open(pq)
    pair = TwoTierIndex(fast, qual)
    row_of = {fast.doc_id_at(r): r for r in range(3)}
    assert pair.quality_row(row_of["doc-a"]) is not None and pair.quality_row(row_of["doc-c"]) is not None
    assert pair.quality_row(row_of["doc-b"]) is None and pair.quality_row(3) is None
    hits = [(d, 0.0, row_of[d]) for d in ("doc-a", "doc-b", "doc-c")]
    s = pair.quality_scores_for_hits([1.0, 0, 0, 0, 0, 0, 0, 0], hits)
    assert abs(s[0] - 1.0) < 1e-6 and s[1] is None and abs(s[2]) < 1e-6
    with pytest.raises(fa.DimensionMismatch):
        pair.quality_scores_for_hits([1.0, 0.0], hits)                     # two_tier.rs:4892
    assert pair.quality_scores_for_hits([1.0, 0, 0, 0, 0, 0, 0, 0], []) == []   # :5614
    # quality_scores_full_coverage: identical id sets stay ALIGNED
    f2, q2 = _fsvi_pair(fa, tmp_path, rng, ["doc-a", "doc-b"], ["doc-a", "doc-b"], "full")
    assert TwoTierIndex(f2, q2).alignment_kind() == TwoTierIndex.ALIGNED
    # a larger pair: 3,000 documents, the quality tier lacks every 7th and has 40 of its own; tombstones on both sides
    ids = [f"note-{i:05d}-{'y' * (i % 3)}" for i in range(3000)]
    q_ids = [d for i, d in enumerate(ids) if i % 7 != 3] + [f"extra-{i:03d}" for i in range(40)]
    fast, qual = _fsvi_pair(fa, tmp_path, rng, ids, q_ids, "big")
    fast_dead = {ids[i] for i in (5, 77, 1500, 2999)}
    qual_dead = {ids[i] for i in (8, 77, 2000)}
    for d in fast_dead:
        assert fast.soft_delete(d)
    for d in qual_dead:
        assert qual.soft_delete(d)
    pair = TwoTierIndex(fast, qual)     # (computed at open in the reference: after the tombstones are in place)
    align, fast_find, qual_find, fcount = _oracle_pairing(oracle, fusion_oracle, fast, qual, fast_dead, qual_dead)
    assert pair.alignment_kind() == (TwoTierIndex.ALIGNED if align[0] == "aligned" else TwoTierIndex.MAPPING) == TwoTierIndex.MAPPING
    for r in range(fcount):
        want = r if align[0] == "aligned" else align[1][r]
        assert pair.quality_row(r) == want, r
    assert pair.unmatched_quality_docs() >= 40
    qslab = oracle.Fsvi(str(tmp_path / "big.quality.idx")).slab()
    query = rng.standard_normal(128).astype(np.float32)
    # a quality-side WAL entry shadows the main row of its document (the latest entry wins)
    wal_doc = ids[10]
    wal_vec = rng.standard_normal(128).astype(np.float32)
    qual.append(wal_doc, rng.standard_normal(128).astype(np.float32))
    qual.append(wal_doc, wal_vec)
    wal_score = oracle.dot_f32_f32(wal_vec, query)
    picks = rng.choice(fcount, 60, replace=False).tolist() + [fast_find(ids[10])]
    hits = [(fast.doc_id_at(r), float(rng.normal()), r) for r in picks]
    hits += [(ids[20], 0.0, 0xFFFFFFFF), ("extra-007", 0.0, 0xFFFFFFFF), ("nowhere", 0.0, 0xFFFFFFFF), (ids[30], 0.0, fcount + 5)]
    got = pair.quality_scores_for_hits(query, hits)
    want = fusion_oracle.quality_scores_for_hits(
        hits, align, fcount, lambda r: float(oracle.dot_f16_f32(qslab[r], query)), quality_find=qual_find, fast_find=fast_find,
        quality_wal=lambda d: float(wal_score) if d == wal_doc else None)
    assert [g is None for g in got] == [w is None for w in want]
    assert [np.float32(g).view(np.uint32) for g in got if g is not None] == [np.float32(w).view(np.uint32) for w in want if w is not None]
    assert any(g is None for g in got)
    assert np.float32(got[len(picks) - 1]).view(np.uint32) == np.float32(wal_score).view(np.uint32)   # the WAL's latest entry
    # the chunked form (fsgpu_quality_scores_for_hits_batched: one multi-query gather launch for a whole chunk of the many-queries flow)
    # gives every query the per-query call's scores, bit for bit — WAL hits, missing rows, rowless hits and empty lists included
    queries = rng.standard_normal((9, 128)).astype(np.float32)
    queries[0] = query
    lists = [hits, [], hits[:5], hits[::-1]] + [[(fast.doc_id_at(int(r)), 0.0, int(r)) for r in rng.choice(fcount, int(rng.integers(1, 40)), replace=False)]
                                               for _ in range(5)]
    got_b = pair.quality_scores_for_hits_batched(queries, lists)
    for qi, hl in enumerate(lists):
        one = pair.quality_scores_for_hits(queries[qi], hl)
        assert [g is None for g in got_b[qi]] == [w is None for w in one], qi
        assert [np.float32(g).view(np.uint32) for g in got_b[qi] if g is not None] == [np.float32(w).view(np.uint32) for w in one if w is not None], qi
    assert [None if g is None else int(np.float32(g).view(np.uint32)) for g in got_b[0]] == [None if g is None else int(np.float32(g).view(np.uint32)) for g in got]
    with pytest.raises(fa.DimensionMismatch):
        pair.quality_scores_for_hits_batched(queries[:, :64], lists)


def test_rescored_fast_pool_searchers_equal_the_oracle_pipeline(oracle):
    """The phase-2 flow of an unattested (FSVI v1) pair end to end — SyncQualityPool::RescoredFastPool, sync_searcher.rs:814-818,
    862-866: Python mirror == libfshost (native) == the oracle pipeline, for aligned raw slabs and a shorter quality tier."""
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.host import NativeTwoTierSearcher
    from frankensearch_amd.two_tier import POOL_RESCORED, SyncTwoTierSearcher, TwoTierConfig
    from oracle import fusion_oracle

    build()
    rng = np.random.default_rng(43)
    n, nq_rows = 30000, 27000                      # the quality tier covers the first 27,000 documents only
    fast_slab = rng.standard_normal((n, 256)).astype(np.float16).view(np.uint16)
    qual_slab = rng.standard_normal((nq_rows, 384)).astype(np.float16).view(np.uint16)
    fast, qual, m2v, bert = _small_two_tier(fa, rng, 64)
    fast, qual = fa.VectorIndex.from_slab(fast_slab), fa.VectorIndex.from_slab(qual_slab)
    doc = lambda r: f"doc-{r:08d}"
    py = SyncTwoTierSearcher(fast, qual, m2v, bert, doc, TwoTierConfig(quality_pool=POOL_RESCORED))
    native = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, quality_pool=1)
    native_pre = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, quality_pool=1, prefetch_quality_embed=2)
    missing = 0
    for trial in range(6):
        fast_ids = rng.integers(0, 5000, 9).tolist()
        qual_ids = [101] + rng.integers(1000, 3000, 10).tolist() + [102]
        lexical = [(doc(int(r)), float(30 - i)) for i, r in enumerate(rng.choice(n, 25, replace=False))]
        k = 10
        out = py.search(fast_ids, qual_ids, k, lexical)
        fv = m2v.embed_token_ids(fast_ids)
        fr, fs = oracle.search_top_k(fast_slab, fv, 30)
        fast_hits = [(doc(int(r)), float(x), int(r)) for r, x in zip(fr, fs)]
        assert out.fast_hits == fast_hits
        qv = bert.embed_token_ids(qual_ids)
        scores = [float(oracle.dot_f16_f32(qual_slab[r], qv)) if r < nq_rows else None for _, _, r in fast_hits]
        missing += sum(s is None for s in scores)
        blended = fusion_oracle.blend_two_tier_aligned(fast_hits, scores, 0.7)
        assert [(d, np.float32(s).view(np.uint32), i) for d, s, i in out.blended] == \
               [(d, np.float32(s).view(np.uint32), i) for d, s, i in blended]
        want_final = fusion_oracle.rrf_fuse(lexical, blended, k)
        assert [h.doc_id for h in out.final_results] == [h.doc_id for h in want_final]
        assert [h.rrf_score for h in out.final_results] == [h.rrf_score for h in want_final]
        for s in (native, native_pre):
            ini, fin, metrics = s.search(fast_ids, qual_ids, k, lexical)
            assert ini == out.initial_results and fin == out.final_results
            assert metrics["quality_search_ms"] > 0
    assert missing > 0   # some fast hits had no quality vector (the reference's None)


def _same_hits(a, b):
    return [(h.index, np.float32(h.score).view(np.uint32)) for h in a] == [(h.index, np.float32(h.score).view(np.uint32)) for h in b]


def test_two_tier_flow_over_eight_virtual_shards_equals_the_unsharded_flow():
    """The whole phase-0 + phase-1 flow (sync_searcher.rs:616-943) with BOTH tiers behind row-sharded handles — eight shards on
    this one GPU, exchanged by peer copies (the rehearsal of the 8-GPU form, SURVEY 8e: the tiers shard identically) — must deliver
    the unsharded flow's fused hits bit for bit: doc ids, rrf scores, ranks.  Fast tier = int8 two-pass (corpus-wide candidate
    set), quality pool Retrieved (sharded exact search) and RescoredFastPool (fsgpu_sharded_quality_scores_for_hits: the gather
    routed to the owning shards)."""
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.host import NativeTwoTierSearcher

    build()
    rng = np.random.default_rng(51)
    n, nq_rows = 41_003, 39_000                          # ragged shards; the quality tier covers fewer documents
    fast_slab = rng.standard_normal((n, 256)).astype(np.float16).view(np.uint16)
    qual_slab = rng.standard_normal((n, 384)).astype(np.float16).view(np.uint16)
    _, _, m2v, bert = _small_two_tier(fa, rng, 64)
    P = fa.NativeShardedIndex.EXCHANGE_PEER_COPY
    doc = lambda r: f"doc-{r:08d}"
    for pool, qrows in ((0, n), (1, nq_rows), (1, n)):
        fast, qual = fa.VectorIndex.from_slab(fast_slab), fa.VectorIndex.from_slab(qual_slab[:qrows])
        sfast = fa.NativeShardedIndex.from_slab(fast_slab, [0] * 8, exchange=P)
        squal = fa.NativeShardedIndex.from_slab(qual_slab[:qrows], [0] * 8, exchange=P)
        assert sfast.shard_count() == 8
        one = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3, quality_pool=pool)
        eight = NativeTwoTierSearcher(sfast, squal, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3, quality_pool=pool)
        exact8 = NativeTwoTierSearcher(sfast, squal, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=0, quality_pool=pool,
                                       prefetch_quality_embed=2)
        exact1 = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=0, quality_pool=pool)
        for trial in range(5):
            fast_ids = rng.integers(0, 5000, int(rng.integers(2, 20))).tolist()
            qual_ids = [101] + rng.integers(1000, 3000, int(rng.integers(3, 25))).tolist() + [102]
            lexical = [(doc(int(r)), float(30 - i)) for i, r in enumerate(rng.choice(n, 30, replace=False))]
            k = 10
            ini1, fin1, _ = one.search(fast_ids, qual_ids, k, lexical)
            ini8, fin8, m8 = eight.search(fast_ids, qual_ids, k, lexical)
            assert ini8 == ini1 and fin8 == fin1, (pool, qrows, trial)
            assert [h.rrf_score for h in fin8] == [h.rrf_score for h in fin1]
            assert m8["refinement_failed"] == 0 and m8["phase2_total_ms"] > 0
            ie1, fe1, _ = exact1.search(fast_ids, qual_ids, k, lexical)
            ie8, fe8, _ = exact8.search(fast_ids, qual_ids, k, lexical)
            assert ie8 == ie1 and fe8 == fe1
        for s in (one, eight, exact1, exact8):
            s.close()
        for h in (fast, qual, sfast, squal):
            h.close()


def test_sharded_two_tier_over_fsvi_catalogs_with_tombstones_and_a_quality_wal(tmp_path):
    """doc_id_mode 0 over fsgpu_sharded_open_fsvi handles: doc ids, tombstones and the WAL live in the handles' catalogs; the
    alignment walk (two_tier.rs:750-866) runs over them and the re-scoring resolves WAL entries / doc ids there.  Equal to the
    unsharded pair opened from the same files, after the same soft deletes and appends."""
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.host import NativeTwoTierSearcher
    from frankensearch_amd.synthetic import random_bert_weights
    from frankensearch_amd.two_tier import TwoTierIndex

    build()
    rng = np.random.default_rng(53)
    n = 3000
    ids = [f"note-{i:05d}-{'z' * (i % 4)}" for i in range(n)]
    q_ids = [d for i, d in enumerate(ids) if i % 9 != 4] + [f"only-quality-{i:03d}" for i in range(25)]
    fast_rows = rng.standard_normal((n, 256)).astype(np.float32)
    qual_rows = rng.standard_normal((len(q_ids), 384)).astype(np.float32)
    pf, pq = str(tmp_path / "vector.fast.idx"), str(tmp_path / "vector.quality.idx")
    fa.write_fsvi(pf, list(zip(ids, fast_rows)), "potion", "r1")
    fa.write_fsvi(pq, list(zip(q_ids, qual_rows)), "minilm", "r1")
    P = fa.NativeShardedIndex.EXCHANGE_PEER_COPY
    fast, qual = fa.VectorIndex.open(pf), fa.VectorIndex.open(pq)
    sfast, squal = fa.NativeShardedIndex.open(pf, [0] * 3, exchange=P), fa.NativeShardedIndex.open(pq, [0] * 3, exchange=P)
    for d in (ids[5], ids[77], ids[1500]):
        assert fast.soft_delete(d) and sfast.soft_delete(d)
    for d in (ids[8], ids[2000]):
        assert qual.soft_delete(d) and squal.soft_delete(d)
    wal_vec = rng.standard_normal(384).astype(np.float32)
    for h in (qual, squal):
        h.append(ids[10], rng.standard_normal(384).astype(np.float32))
        h.append(ids[10], wal_vec)
    p1, p3 = TwoTierIndex(fast, qual), TwoTierIndex(sfast, squal)
    assert p1.alignment_kind() == p3.alignment_kind() == TwoTierIndex.MAPPING
    assert [p1.quality_row(r) for r in range(n + 2)] == [p3.quality_row(r) for r in range(n + 2)]
    assert p1.unmatched_quality_docs() == p3.unmatched_quality_docs() >= 25
    query = rng.standard_normal(384).astype(np.float32)
    row10 = next(r for r in range(n) if fast.doc_id_at(r) == ids[10])
    hits = [(fast.doc_id_at(r), 0.0, r) for r in rng.choice(n, 50, replace=False).tolist() + [row10]]
    hits += [(ids[20], 0.0, 0xFFFFFFFF), ("only-quality-007", 0.0, 0xFFFFFFFF), ("nowhere", 0.0, 0xFFFFFFFF), (ids[30], 0.0, n + 5)]
    s1, s3 = p1.quality_scores_for_hits(query, hits), p3.quality_scores_for_hits(query, hits)
    assert [x is None for x in s1] == [x is None for x in s3] and any(x is None for x in s1)
    assert [np.float32(x).view(np.uint32) for x in s1 if x is not None] == [np.float32(x).view(np.uint32) for x in s3 if x is not None]
    m2v = fa.Model2VecEmbedder(rng.standard_normal((5000, 256)).astype(np.float32))
    bert = fa.NativeEmbedder(random_bert_weights(5, 3000, 384, 6, 1536))
    for pool in (0, 1):
        one = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=0, quality_pool=pool)
        three = NativeTwoTierSearcher(sfast, squal, m2v, bert, doc_id_mode=0, quality_pool=pool)
        for trial in range(4):
            fast_ids = rng.integers(0, 5000, 8).tolist()
            qual_ids = [101] + rng.integers(1000, 3000, 7).tolist() + [102]
            lexical = [(ids[int(r)], float(30 - i)) for i, r in enumerate(rng.choice(n, 30, replace=False))]
            i1, f1, _ = one.search(fast_ids, qual_ids, 10, lexical)
            i3, f3, _ = three.search(fast_ids, qual_ids, 10, lexical)
            assert i3 == i1 and f3 == f1, (pool, trial)
            assert all(h.doc_id.startswith(("note-", "only-quality-")) for h in f3)
        one.close()
        three.close()
    # a search begun on the handle is in flight: soft_delete / append must refuse to rewrite the shards' bitmaps under it
    t = sfast.search_begin(rng.standard_normal((4, 256)).astype(np.float32), 5)
    with pytest.raises(fa.InvalidConfig):
        sfast.soft_delete(ids[40])
    with pytest.raises(fa.InvalidConfig):
        sfast.append(ids[41], rng.standard_normal(256).astype(np.float32))
    sfast.search_end(t)
    assert sfast.soft_delete(ids[40])


def test_refinement_failure_returns_the_initial_results():
    """sync_searcher.rs:820-839: a quality pool that cannot be produced (here: the quality index has another dimension than the
    quality embedder, so search_top_k / quality_scores_for_hits answer DimensionMismatch) is a RefinementFailed outcome — the search
    succeeds and final_results are the phase-0 results.  The same mismatch on the FAST tier fails the search."""
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.host import NativeTwoTierSearcher

    build()
    rng = np.random.default_rng(57)
    n = 20_000
    fast_slab = rng.standard_normal((n, 256)).astype(np.float16).view(np.uint16)
    wrong_quality = rng.standard_normal((n, 128)).astype(np.float16).view(np.uint16)   # the encoder emits 384 dimensions
    _, _, m2v, bert = _small_two_tier(fa, rng, 64)
    P = fa.NativeShardedIndex.EXCHANGE_PEER_COPY
    pairs = [(fa.VectorIndex.from_slab(fast_slab), fa.VectorIndex.from_slab(wrong_quality)),
             (fa.NativeShardedIndex.from_slab(fast_slab, [0] * 2, exchange=P), fa.NativeShardedIndex.from_slab(wrong_quality, [0] * 2, exchange=P))]
    doc = lambda r: f"doc-{r:08d}"
    for fast, qual in pairs:
        for pool in (0, 1):
            for prefetch in (0, 2):
                s = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, quality_pool=pool, prefetch_quality_embed=prefetch)
                lexical = [(doc(int(r)), float(30 - i)) for i, r in enumerate(rng.choice(n, 20, replace=False))]
                ini, fin, m = s.search(rng.integers(0, 5000, 9).tolist(), [101, 1500, 1600, 102], 10, lexical)
                assert m["refinement_failed"] == 1 and len(ini) == 10 and fin == ini
                s.close()
        bad = NativeTwoTierSearcher(qual, fast, m2v, bert, doc_id_mode=1)   # the FAST tier's dimension is wrong: an error
        with pytest.raises(fa.DimensionMismatch):
            bad.search([1, 2, 3], [101, 1500, 102], 10, [])
        bad.close()


def test_sharded_handle_coalesces_concurrent_single_query_callers():
    """fsgpu_sharded_set_coalescing: single-query exact and int8 two-pass searches in flight together ride ONE search of the shards
    (the batched mode / one two-pass batch) and every caller gets exactly the hits of the lone call."""
    import threading

    import frankensearch_amd as fa
    from frankensearch_amd.build import build

    build()
    rng = np.random.default_rng(59)
    n, dim, nq = 90_001, 384, 80
    slab = rng.standard_normal((n, dim)).astype(np.float16).view(np.uint16)
    idx = fa.NativeShardedIndex.from_slab(slab, [0] * 4, exchange=fa.NativeShardedIndex.EXCHANGE_PEER_COPY)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    want = [idx.search(q[i], 10) for i in range(nq)]
    want8 = [idx.search(q[i], 30, idx.INT8_TWO_PASS, 3) for i in range(nq)]
    idx.set_coalescing(64, 20_000)
    got, got8, errs = [None] * nq, [None] * nq, []

    def call(i):
        try:
            got[i] = idx.search(q[i], 10)
            got8[i] = idx.search(q[i], 30, idx.INT8_TWO_PASS, 3)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=call, args=(i,)) for i in range(nq)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(nq):
        for g, w in ((got[i], want[i]), (got8[i], want8[i])):
            assert np.array_equal(g[0], w[0]) and np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32)) and np.array_equal(g[2], w[2]), i
    batches, requests = idx.coalescing_stats()
    assert requests == 2 * nq and batches < nq, (batches, requests)
    idx.set_coalescing(0, 0)
    idx.close()


def test_embed_search_stream_overlapped_equals_the_stages_in_turn():
    """fshost_embed_search_stream (BASELINE config 5's loop): token-id batches -> MiniLM on the GPU -> batched exact top-k, with the
    encode of group g + 1 running under the search of group g.  The overlapped pipeline, the serial one, the sharded handle and a
    step-by-step drive of the same C ABI calls all return the same rows and score bits."""
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.host import embed_search_stream

    build()
    rng = np.random.default_rng(61)
    n, batch, nb, k = 60_000, 64, 5, 10
    slab = rng.standard_normal((n, 384)).astype(np.float16).view(np.uint16)
    _, _, _, bert = _small_two_tier(fa, rng, 64)
    texts = [[101] + rng.integers(1000, 3000, int(rng.integers(6, 31))).tolist() + [102] for _ in range(batch * nb)]
    offs = np.zeros(len(texts) + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(t) for t in texts])
    ids = np.concatenate([np.asarray(t, dtype=np.int32) for t in texts])
    idx = fa.VectorIndex.from_slab(slab)
    sh = fa.NativeShardedIndex.from_slab(slab, [0] * 8, exchange=fa.NativeShardedIndex.EXCHANGE_PEER_COPY)
    # step by step: one embed call and one batched search per encoder batch
    want_r, want_s = [], []
    for b in range(nb):
        lo, hi = b * batch, (b + 1) * batch
        emb = np.empty((batch, 384), dtype=np.float32)
        bert.embed_flat(ids[offs[lo]:offs[hi]], (offs[lo:hi + 1] - offs[lo]).astype(np.uint32), emb)
        r, s, c, _ = idx.search_batched(emb, k)
        assert np.all(c == k)
        want_r.append(r)
        want_s.append(s)
    want_r, want_s = np.concatenate(want_r), np.concatenate(want_s)
    for target in (idx, sh):
        for group in (1, 2):
            for overlap in (False, True):
                for host_handoff in (False, True):   # the embeddings handed over in device memory / through host vectors
                    r, s, c, stats = embed_search_stream(bert, target, ids, offs, batch, k, group=group, overlap=overlap,
                                                         host_handoff=host_handoff)
                    assert np.array_equal(r, want_r) and np.array_equal(s.view(np.uint32), want_s.view(np.uint32)), (group, overlap, host_handoff)
                    assert np.all(c == k) and stats["queries"] == batch * nb and stats["groups"] == (nb + group - 1) // group
                    assert stats["queries_per_sec"] > 0 and stats["device_resident_handoff"] == (0 if host_handoff else 1)
    # data-parallel encoders (fshost_embed_search_stream_dp): one encoder handle per device — three here, all on device 0,
    # over 2 query groups x 4 row shards — each embeds its slice of every group, the search fetches the slices peer to peer
    from frankensearch_amd.synthetic import random_bert_weights  # noqa: F401  (the helper's weights are reused below)
    hy = fa.NativeShardedIndex.from_slab(slab, [0] * 8, exchange=fa.NativeShardedIndex.EXCHANGE_PEER_COPY, query_groups=2)
    encoders = [bert, _small_two_tier(fa, np.random.default_rng(61), 64)[3], _small_two_tier(fa, np.random.default_rng(61), 64)[3]]
    for target in (sh, hy):
        for group in (1, 2):
            for overlap in (False, True):
                r, s, c, stats = embed_search_stream(encoders, target, ids, offs, batch, k, group=group, overlap=overlap)
                assert stats["encoders"] == 3 and stats["device_resident_handoff"] == 1
                # (an encoder's slice is a batch of its own: the embeddings agree to the encoder's tolerance, so compare through it)
                assert np.mean(r[:, 0] == want_r[:, 0]) > 0.98 and np.allclose(s, want_s, atol=2e-3), (group, overlap)
    hy.close()
    idx.close()
    sh.close()
