"""BASELINE config 1 at its stated size: 100,000 synthetic token-id documents -> Model2Vec pool over a 32,768-row
formula table -> FSVI v1 write -> reopen -> exact top-10, for dim 128 and 256.  Every stage is compared with the oracle pipeline:
pooled embeddings bit for bit, the FSVI file byte for byte, hits by (row, doc id) and f32 score bits.

The table is the reference's synthetic Model2Vec model (`val = row * 0.1 + col * 0.01`,
crates/frankensearch-embed/src/model2vec_embedder.rs:691-849) — every pooled vector points nearly the same way, so after the
f16 encode the scan meets long runs of tied scores and the (score desc, row asc) order decides — and a folded variant of it whose
documents are spread out (the miniature of tests/test_gpu_parity.py, at full size)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VOCAB, NDOCS = 32_768, 100_000


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def fa():
    import frankensearch_amd as fa
    from frankensearch_amd.build import build

    build()
    return fa


def formula_table(dim: int, folded: bool) -> np.ndarray:
    r = np.arange(VOCAB, dtype=np.float64)[:, None]
    c = np.arange(dim, dtype=np.float64)[None, :]
    v = r * 0.1 + c * 0.01
    if folded:
        v = (v % 1.7) - 0.8
    return v.astype(np.float32)


@pytest.mark.parametrize("dim,folded", [(128, False), (256, False), (128, True), (256, True)])
def test_config1_at_size_pool_write_reopen_search(fa, oracle, tmp_path, dim, folded):
    rng = np.random.default_rng(1000 + dim + folded)
    table = formula_table(dim, folded)
    lens = rng.integers(3, 40, NDOCS)
    docs = [rng.integers(0, VOCAB + (64 if i % 97 == 0 else 0), int(lens[i])).tolist() for i in range(NDOCS)]   # a few out-of-vocabulary ids
    m = fa.Model2VecEmbedder(table)
    emb = m.embed_batch_token_ids(docs)
    assert emb.shape == (NDOCS, dim)
    # the pool: every 7th document against the oracle's per-text restatement (model2vec_embedder.rs:310-335), plus the ends
    for i in list(range(0, NDOCS, 7)) + [NDOCS - 1]:
        assert np.array_equal(bits(emb[i]), bits(oracle.m2v_embed(table, docs[i]))), i
    named = [(f"doc-{(i * 7919) % NDOCS:06d}", emb[i]) for i in range(NDOCS)]   # ids in a scrambled order: the writer sorts by FNV-1a
    p_gpu, p_ref = str(tmp_path / "vector.fast.idx"), str(tmp_path / "ref.idx")
    fa.write_fsvi(p_gpu, named, "potion-multilingual-128M", "a28f4ee", 0)
    assert oracle.fsvi_write(p_ref, named, "potion-multilingual-128M", "a28f4ee", 0) == 0
    a, b = open(p_gpu, "rb").read(), open(p_ref, "rb").read()
    assert len(a) == len(b) and a == b
    g, o = fa.VectorIndex.open(p_gpu), oracle.Fsvi(p_ref)
    assert g.record_count() == NDOCS and g.dimension() == dim
    queries = [oracle.m2v_embed(table, rng.integers(0, VOCAB, int(rng.integers(2, 12))).tolist()) for _ in range(24)]
    queries.append(emb[12345].copy())                     # a document as its own query
    for q in queries:
        oh, os_ = o.search_top_k(q, 10)
        gh = g.search_top_k(q, 10)
        assert [(h.index, h.doc_id) for h in gh] == [(h[0], h[2]) for h in oh]
        assert np.array_equal(bits([h.score for h in gh]), bits(os_))
    # the same index through the batched entry point (row level): the oracle's rows and bits again
    slab = o.slab()
    qs = np.stack(queries).astype(np.float32)
    rows, scores, counts, _ = g.search_batched(qs, 10)
    for qi in range(qs.shape[0]):
        er, es = oracle.search_top_k(slab, qs[qi], 10)
        assert counts[qi] == 10 and np.array_equal(rows[qi], er) and np.array_equal(bits(scores[qi]), bits(es)), qi
    g.close()
    o.close()
