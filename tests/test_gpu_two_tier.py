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
