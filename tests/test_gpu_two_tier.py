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
