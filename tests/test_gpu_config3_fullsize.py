"""BASELINE config 3 at its stated size on one GPU: 10M x 256 fast tier (potion) + 10M x 384 quality tier (MiniLM), the
two-phase flow of SyncTwoTierSearcher (crates/frankensearch-fusion/src/sync_searcher.rs:616-943) through libfshost / the C ABI,
checked against the oracle pipeline: fast-tier hits of search_top_k_int8_two_pass (the reference's fast-tier default,
two_tier.rs:1318-1337), initial RRF, the quality pool of BOTH branches (sync_searcher.rs:800-918: Retrieved = an independent
quality-tier search + blend_two_tier; RescoredFastPool = quality_scores_for_hits + blend_two_tier_aligned), refined RRF — doc ids,
rrf scores and blend scores.  Runs on the GPU box only; the oracle passes run on the host copy of each slab in turn."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, K = 10_000_000, 10


def test_config3_two_tier_flow_at_10m_rows_matches_the_oracle_pipeline(oracle):
    import torch
    import frankensearch_amd as fa
    from frankensearch_amd.build import build
    from frankensearch_amd.host import NativeTwoTierSearcher
    from frankensearch_amd.synthetic import random_bert_weights
    from oracle import fusion_oracle

    build()
    sys.path.insert(0, ROOT)
    import bench

    dev = torch.device("cuda", 0)
    fast_slab = bench.gen_corpus(0, N, 256, dev)
    qual_slab = bench.gen_corpus(0, N, 384, dev)
    fast = fa.VectorIndex.from_device_slab(fast_slab.data_ptr(), N, 256, device=0, keepalive=fast_slab)
    qual = fa.VectorIndex.from_device_slab(qual_slab.data_ptr(), N, 384, device=0, keepalive=qual_slab)
    rng = np.random.default_rng(71)
    table = rng.standard_normal((20_000, 256)).astype(np.float32)
    m2v = fa.Model2VecEmbedder(table)
    bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536))
    doc = lambda r: f"doc-{int(r):08d}"
    fetch = 3 * K
    queries = []
    for _ in range(4):
        fast_ids = rng.integers(0, 20_000, int(rng.integers(4, 20))).tolist()
        qual_ids = [101] + rng.integers(1000, 30000, int(rng.integers(6, 30))).tolist() + [102]
        lexical = [(doc(r), float(fetch - i)) for i, r in enumerate(rng.choice(N, fetch, replace=False))]
        queries.append((fast_ids, qual_ids, lexical))
    retrieved = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3, quality_int8_latency=True)
    rescored = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3, quality_pool=1)
    got = [(retrieved.search(f, q, K, lex), rescored.search(f, q, K, lex)) for f, q, lex in queries]
    nthreads = min(os.cpu_count() or 1, 16)
    # ---- fast tier on the host copy: m2v pool (bit-exact oracle), int8 two-pass with the reference's quantisers
    host = fast_slab.view(torch.int16).cpu().numpy().view(np.uint16)
    host_i8 = oracle.quantize_slab_i8(host)
    fast_hits = []
    for fast_ids, _, _ in queries:
        fv = oracle.m2v_embed(table, fast_ids)
        fr, fs = oracle.search_int8_two_pass(host, fv, fetch, 3, slab_i8=host_i8)
        fast_hits.append([(doc(r), float(s), int(r)) for r, s in zip(fr, fs)])
    del host, host_i8
    # ---- quality tier on the host copy (the query vector is the GPU encoder's: its own tolerance tests are in test_gpu_bert.py)
    host = qual_slab.view(torch.int16).cpu().numpy().view(np.uint16)
    for (fast_ids, qual_ids, lexical), fh, ((ini_r, fin_r, m_r), (ini_s, fin_s, m_s)) in zip(queries, fast_hits, got):
        want_initial = fusion_oracle.rrf_fuse(lexical, fh, K)
        for ini in (ini_r, ini_s):
            assert [h.doc_id for h in ini] == [h.doc_id for h in want_initial]
            assert [h.rrf_score for h in ini] == [h.rrf_score for h in want_initial]
            assert [h.semantic_index for h in ini] == [h.semantic_index for h in want_initial]
        qv = bert.embed_token_ids(qual_ids)
        # Retrieved: an independent search of the quality tier, blend_two_tier, the fast row carried by blended hits
        qr, qs = oracle.search_top_k(host, qv, fetch, nthreads=nthreads)
        qual_hits = [(doc(r), float(s), int(r)) for r, s in zip(qr, qs)]
        blended = fusion_oracle.blend_two_tier(fh, qual_hits, 0.7)
        fidx = {d: i for d, _, i in fh}
        blended = [(d, sc, fidx.get(d, 0xFFFFFFFF)) for d, sc, _ in blended]
        want = fusion_oracle.rrf_fuse(lexical, blended, K)
        assert [h.doc_id for h in fin_r] == [h.doc_id for h in want]
        assert [h.rrf_score for h in fin_r] == [h.rrf_score for h in want]
        assert [np.float32(h.semantic_score).view(np.uint32) if h.semantic_score is not None else None for h in fin_r] == \
               [np.float32(h.semantic_score).view(np.uint32) if h.semantic_score is not None else None for h in want]
        assert m_r["refinement_failed"] == 0
        # RescoredFastPool: the fast pool's rows scored on the quality slab (aligned raw slabs: fast row i = quality row i)
        scores = [float(oracle.dot_f16_f32(host[r], qv)) for _, _, r in fh]
        blended_a = fusion_oracle.blend_two_tier_aligned(fh, scores, 0.7)
        want_a = fusion_oracle.rrf_fuse(lexical, blended_a, K)
        assert [h.doc_id for h in fin_s] == [h.doc_id for h in want_a]
        assert [h.rrf_score for h in fin_s] == [h.rrf_score for h in want_a]
        assert [np.float32(h.semantic_score).view(np.uint32) if h.semantic_score is not None else None for h in fin_s] == \
               [np.float32(h.semantic_score).view(np.uint32) if h.semantic_score is not None else None for h in want_a]
        assert m_s["refinement_failed"] == 0
    del host
    for h in (retrieved, rescored, fast, qual, m2v, bert):
        h.close()
