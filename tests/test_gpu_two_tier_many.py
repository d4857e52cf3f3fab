"""fshost_two_tier_search_many — the two-phase flow (crates/frankensearch-fusion/src/sync_searcher.rs:616-943) for MANY queries in one
call: batched embeds, batched tier searches, pipelined over chunks, per-query fusion on host threads.  Checked three ways:

* against the ORACLE PIPELINE on the vectors the call itself searched with (every tier answer is the per-query search's — rows and
  score bits —, every fused list the oracle fusion's: doc ids, rrf scores, ranks), 1,000 queries, ragged chunks;
* the vectors: Model2Vec bit-identical to the per-text embedder, MiniLM within the encoder's tolerance of the per-text embedding
  (the encoder picks kernels by batch shape — tests/test_gpu_bert.py holds it to cos >= 0.999 / 2e-3 against the f32 oracle);
* against fshost_two_tier_search called query by query: the Initial lists identical for every query (nothing in phase 0 depends on the
  batch), the Refined lists identical except where the embedding's last bits move a near-tie."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def _queries(rng, nq, n, fast_vocab=5000, quality_vocab=3000, lex=30):
    doc = lambda r: f"doc-{int(r):08d}"
    fast = [rng.integers(0, fast_vocab, int(rng.integers(1, 24))).tolist() for _ in range(nq)]
    qual = [[101] + rng.integers(1000, quality_vocab, int(rng.integers(2, 30))).tolist() + [102] for _ in range(nq)]
    lexical = [[(doc(r), float(lex - i)) for i, r in enumerate(rng.choice(n, lex, replace=False))] for _ in range(nq)]
    return fast, qual, lexical


def _small(fa, rng, n):
    from frankensearch_amd.synthetic import random_bert_weights
    fast_slab = rng.standard_normal((n, 256)).astype(np.float16).view(np.uint16)
    qual_slab = rng.standard_normal((n, 384)).astype(np.float16).view(np.uint16)
    table = rng.standard_normal((5000, 256)).astype(np.float32)
    return fast_slab, qual_slab, table, random_bert_weights(5, 3000, 384, 6, 1536)


@pytest.mark.parametrize("int8_mult", [3, 0])
def test_many_form_equals_the_oracle_pipeline_and_the_per_query_searcher_on_1000_queries(oracle, int8_mult):
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.host import NativeTwoTierSearcher
    from oracle import fusion_oracle

    build()
    rng = np.random.default_rng(101 + int8_mult)
    n, nq, k = 40_000, 1000, 10
    fetch = 3 * k
    fast_slab, qual_slab, table, w = _small(fa, rng, n)
    fast, qual = fa.VectorIndex.from_slab(fast_slab), fa.VectorIndex.from_slab(qual_slab)
    m2v, bert = fa.Model2VecEmbedder(table), fa.NativeEmbedder(w)
    doc = lambda r: f"doc-{int(r):08d}"
    s = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=int8_mult)
    fq, qq, lex = _queries(rng, nq, n)
    lex[7] = []                                             # a query without lexical hits
    # chunk 256 -> four chunks, the last one ragged (232); vectors handed back (host path of the embeddings)
    ini, fin, rf, st, fv, qv = s.search_many(fq, qq, k, lex, chunk=256, want_vectors=True)
    assert st["chunks"] == 4 and st["queries"] == nq and not rf.any() and st["refinement_failed"] == 0
    # ... and the same call with the embeddings left in device memory between the stages: identical in every field
    ini_d, fin_d, rf_d, st_d = s.search_many(fq, qq, k, lex, chunk=256)
    assert st_d["device_resident_handoff"] == 3
    assert ini_d == ini and fin_d == fin
    # vectors: Model2Vec bit-identical to the per-text call; MiniLM within the encoder's tolerance of the per-text call
    for qi in range(0, nq, 37):
        assert np.array_equal(_bits(fv[qi]), _bits(m2v.embed_token_ids(fq[qi])))
        one = bert.embed_token_ids(qq[qi])
        assert np.max(np.abs(one - qv[qi])) <= 2e-3 and float(np.sum(one * qv[qi])) >= 0.999
    # the oracle pipeline on those vectors (quantisers, scans and fusion all restated on the CPU)
    slab_i8 = oracle.quantize_slab_i8(fast_slab) if int8_mult else None
    for qi in list(range(0, nq, 9)) + [7, 255, 256, 767, 768, 999]:
        if int8_mult:
            fr, fs = oracle.search_int8_two_pass(fast_slab, fv[qi], fetch, int8_mult, slab_i8=slab_i8)
        else:
            fr, fs = oracle.search_top_k(fast_slab, fv[qi], fetch)
        fh = [(doc(r), float(x), int(r)) for r, x in zip(fr, fs)]
        want_i = fusion_oracle.rrf_fuse(lex[qi], fh, k)
        assert [h.doc_id for h in ini[qi]] == [h.doc_id for h in want_i], qi
        assert [h.rrf_score for h in ini[qi]] == [h.rrf_score for h in want_i], qi
        assert [h.semantic_index for h in ini[qi]] == [h.semantic_index for h in want_i], qi
        qr, qs = oracle.search_top_k(qual_slab, qv[qi], fetch)
        qh = [(doc(r), float(x), int(r)) for r, x in zip(qr, qs)]
        blended = fusion_oracle.blend_two_tier(fh, qh, 0.7)
        fidx = {d: i for d, _, i in fh}
        blended = [(d, sc, fidx.get(d, 0xFFFFFFFF)) for d, sc, _ in blended]
        want_f = fusion_oracle.rrf_fuse(lex[qi], blended, k)
        assert [h.doc_id for h in fin[qi]] == [h.doc_id for h in want_f], qi
        assert [h.rrf_score for h in fin[qi]] == [h.rrf_score for h in want_f], qi
        assert [None if h.semantic_score is None else int(_bits(h.semantic_score)) for h in fin[qi]] == \
               [None if h.semantic_score is None else int(_bits(h.semantic_score)) for h in want_f], qi
    # every query against the product's own per-query searches on the same vectors (the tier answers the fusion consumed)
    for qi in range(0, nq, 5):
        hits = fast.search_top_k_int8_two_pass(fv[qi], fetch, int8_mult) if int8_mult else None
        if hits is not None:
            fh = [(doc(h.index), float(h.score), int(h.index)) for h in hits]
        else:
            r_, s_, c_ = fast.search_batch(fv[qi], fetch)
            fh = [(doc(r_[0, i]), float(s_[0, i]), int(r_[0, i])) for i in range(int(c_[0]))]
        want_i = fusion_oracle.rrf_fuse(lex[qi], fh, k)
        assert [(h.doc_id, h.rrf_score) for h in ini[qi]] == [(h.doc_id, h.rrf_score) for h in want_i], qi
    # fshost_two_tier_search, query by query: phase 0 identical for every query; phase 1 identical unless the embedding's last bits
    # (single-text kernels vs batch kernels) move a near-tie
    same_final = 0
    for qi in range(nq):
        i1, f1, m1 = s.search(fq[qi], qq[qi], k, lex[qi])
        assert i1 == ini[qi], qi
        same_final += int([h.doc_id for h in f1] == [h.doc_id for h in fin[qi]])
    assert same_final >= 0.97 * nq, same_final
    for h in (s, fast, qual, m2v, bert):
        h.close()


def test_many_form_over_sharded_tiers_rescored_pool_doc_id_tables_and_failures(tmp_path):
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.host import NativeTwoTierSearcher
    from frankensearch_amd.synthetic import random_bert_weights

    build()
    rng = np.random.default_rng(202)
    n, nq, k = 41_003, 300, 10
    fast_slab, qual_slab, table, w = _small(fa, rng, n)
    fast, qual = fa.VectorIndex.from_slab(fast_slab), fa.VectorIndex.from_slab(qual_slab)
    m2v, bert = fa.Model2VecEmbedder(table), fa.NativeEmbedder(w)
    fq, qq, lex = _queries(rng, nq, n)
    one = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3)
    base = one.search_many(fq, qq, k, lex, chunk=128)
    # one chunk for everything, a chunk of one query, more fusion threads: the same lists
    # (phase 0 does not depend on the chunking at all; phase 1 does through the MiniLM batch shape: the encoder's tolerance)
    def same_docs(a, b):
        return sum(int([h.doc_id for h in x] == [h.doc_id for h in y]) for x, y in zip(a, b))
    for chunk, threads in ((128, 1), (128, 5)):
        got = one.search_many(fq, qq, k, lex, chunk=chunk, fusion_threads=threads)
        assert got[0] == base[0] and got[1] == base[1], (chunk, threads)
    for chunk, threads in ((0, 0), (1000, 1), (77, 5)):
        got = one.search_many(fq, qq, k, lex, chunk=chunk, fusion_threads=threads)
        assert got[0] == base[0] and same_docs(got[1], base[1]) >= 0.97 * nq, (chunk, threads)
    got = one.search_many(fq[:3], qq[:3], k, lex[:3], chunk=1)
    assert got[0] == base[0][:3] and got[3]["chunks"] == 3
    # no lexical source at all; an empty call
    nolex = one.search_many(fq[:40], qq[:40], k, None)
    assert all(len(x) == k and all(h.lexical_rank is None for h in x) for x in nolex[1])
    empty = one.search_many([], [], k, [])
    assert empty[0] == [] and empty[3]["queries"] == 0
    # both tiers behind row-sharded handles (virtual shards, peer copies; 2 query groups x 2 row shards): the unsharded lists
    P = fa.NativeShardedIndex.EXCHANGE_PEER_COPY
    sfast = fa.NativeShardedIndex.from_slab(fast_slab, [0] * 4, exchange=P, query_groups=2)
    squal = fa.NativeShardedIndex.from_slab(qual_slab, [0] * 4, exchange=P, query_groups=2)
    four = NativeTwoTierSearcher(sfast, squal, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3)
    got = four.search_many(fq, qq, k, lex, chunk=128)
    assert got[0] == base[0] and got[1] == base[1]
    four.close()
    # RescoredFastPool (sync_searcher.rs:814-818): phase 1 = quality_scores_for_hits per query on the fusion threads
    resc = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3, quality_pool=1)
    got = resc.search_many(fq, qq, k, lex, chunk=128)
    assert got[0] == base[0]
    same = 0
    for qi in range(0, nq, 3):
        i1, f1, _ = resc.search(fq[qi], qq[qi], k, lex[qi])
        assert i1 == got[0][qi]
        same += int([h.doc_id for h in f1] == [h.doc_id for h in got[1][qi]])
    assert same >= 0.95 * len(range(0, nq, 3)), same
    sresc = NativeTwoTierSearcher(sfast, squal, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3, quality_pool=1)
    got_s = sresc.search_many(fq, qq, k, lex, chunk=128)
    assert got_s[0] == got[0] and got_s[1] == got[1]
    for h in (resc, sresc, sfast, squal):
        h.close()
    # refinement failure (sync_searcher.rs:820-839): a quality index of the wrong dimension -> final = initial, flagged per query;
    # the same mismatch on the FAST tier fails the call
    wrong = fa.VectorIndex.from_slab(rng.standard_normal((n, 128)).astype(np.float16).view(np.uint16))
    bad_q = NativeTwoTierSearcher(fast, wrong, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3)
    ini, fin, rf, st = bad_q.search_many(fq[:50], qq[:50], k, lex[:50], chunk=20)
    assert rf.all() and st["refinement_failed"] == 50 and fin == ini and ini == base[0][:50]
    bad_q.close()
    bad_f = NativeTwoTierSearcher(wrong, qual, m2v, bert, doc_id_mode=1)
    with pytest.raises(fa.DimensionMismatch):
        bad_f.search_many(fq[:5], qq[:5], k, lex[:5])
    bad_f.close()
    # a token id outside the vocabulary fails the call with the embedder's message
    with pytest.raises(Exception) as err:
        one.search_many(fq[:4], [[101, 10_000_000, 102]] * 4, k, lex[:4])
    assert str(err.value)
    one.close()
    # doc ids from FSVI tables (doc_id_mode 0): exact tier searches go through search_hits query by query, as the per-query flow does
    m = 3000
    ids = [f"note-{i:05d}-{'x' * (i % 4)}" for i in range(m)]
    pf, pq = str(tmp_path / "vector.fast.idx"), str(tmp_path / "vector.quality.idx")
    fa.write_fsvi(pf, list(zip(ids, rng.standard_normal((m, 256)).astype(np.float32))), "potion", "r1")
    fa.write_fsvi(pq, list(zip(ids, rng.standard_normal((m, 384)).astype(np.float32))), "minilm", "r1")
    ffast, fqual = fa.VectorIndex.open(pf), fa.VectorIndex.open(pq)
    for mult in (0, 3):
        t = NativeTwoTierSearcher(ffast, fqual, m2v, bert, doc_id_mode=0, fast_tier_int8_multiplier=mult)
        lex_n = [[(ids[int(r)], float(30 - i)) for i, r in enumerate(rng.choice(m, 30, replace=False))] for _ in range(60)]
        ini, fin, rf, st = t.search_many(fq[:60], qq[:60], k, lex_n, chunk=25)
        same = 0
        for qi in range(60):
            i1, f1, _ = t.search(fq[qi], qq[qi], k, lex_n[qi])
            assert i1 == ini[qi] and all(h.doc_id.startswith("note-") for h in fin[qi])
            same += int([h.doc_id for h in f1] == [h.doc_id for h in fin[qi]])
        assert same >= 55, same
        t.close()
    for h in (ffast, fqual, fast, qual, wrong, m2v, bert):
        h.close()


def test_concurrent_per_query_callers_ride_the_engine_with_dynamic_batching():
    """fshost_two_tier_set_batching: fshost_two_tier_search calls from many host threads are collected into chunks and answered by the
    many-queries engine, one wake-up per query.  Every caller gets ITS query's lists: the Initial list identical to the unbatched call's,
    the Refined list identical up to the encoder's batch-shape tolerance; different k in flight together; errors reach their caller."""
    import threading
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.host import NativeTwoTierSearcher

    build()
    rng = np.random.default_rng(303)
    n, nq = 40_000, 600
    fast_slab, qual_slab, table, w = _small(fa, rng, n)
    fast, qual = fa.VectorIndex.from_slab(fast_slab), fa.VectorIndex.from_slab(qual_slab)
    m2v, bert = fa.Model2VecEmbedder(table), fa.NativeEmbedder(w)
    fq, qq, lex = _queries(rng, nq, n)
    s = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3)
    ks = [10 if i % 7 else 5 for i in range(nq)]
    want = [s.search(fq[i], qq[i], ks[i], lex[i]) for i in range(nq)]
    s.set_batching(256, 300)
    got = [None] * nq
    errors = []

    def caller(tid, nthreads):
        try:
            for i in range(tid, nq, nthreads):
                got[i] = s.search(fq[i], qq[i], ks[i], lex[i])
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    for nthreads in (1, 48):
        got = [None] * nq
        threads = [threading.Thread(target=caller, args=(t, nthreads)) for t in range(nthreads)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors[:2]
        same = 0
        for i in range(nq):
            assert got[i][0] == want[i][0], (nthreads, i)
            assert len(got[i][1]) == ks[i]
            same += int([h.doc_id for h in got[i][1]] == [h.doc_id for h in want[i][1]])
            assert got[i][2]["phase1_total_ms"] > 0 and got[i][2]["refinement_failed"] == 0
        assert same >= 0.97 * nq, (nthreads, same)
    chunks, requests = s.batching_stats()
    assert requests == 2 * nq and chunks < requests     # the 48 callers shared chunks
    # an out-of-vocabulary token fails ITS caller (and whoever shared the chunk's embedding call), not the searcher
    with pytest.raises(Exception) as err:
        s.search(fq[0], [101, 10_000_000, 102], 10, lex[0])
    assert str(err.value)
    assert s.search(fq[3], qq[3], ks[3], lex[3])[0] == want[3][0]
    s.set_batching(0)
    assert s.search(fq[4], qq[4], ks[4], lex[4])[0] == want[4][0]
    for h in (s, fast, qual, m2v, bert):
        h.close()
