"""BASELINE config 5 at full size on one GPU: 50M x 384 f16 (38.4 GB resident in HBM, generated on the GPU with the
reference's bench recipe) + batch-256 MiniLM-shaped query encoding, checked through size-independent properties and a
filtered-subset comparison against the oracle.  (The 8-GPU form of config 5 shards these rows 8 ways; the sharding logic
itself is covered by tests/test_gpu_sharded.py and tests/test_sharded_gloo.py.)  Runs on the GPU box only."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, DIM, K = 50_000_000, 384, 10


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def env():
    import torch
    import frankensearch_amd as fa
    from frankensearch_amd.build import build

    build()
    assert torch.cuda.is_available()
    free, _ = torch.cuda.mem_get_info(0)
    if free < 60 << 30:
        pytest.skip("needs ~45 GB of free HBM")
    sys.path.insert(0, ROOT)
    import bench

    dev = torch.device("cuda", 0)
    slab = bench.gen_corpus(0, N, DIM, dev)
    queries = bench.gen_queries(300, DIM, dev)
    idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), N, DIM, device=0, keepalive=slab)
    yield {"torch": torch, "fa": fa, "slab": slab, "queries": queries, "idx": idx, "dev": dev, "bench": bench}
    idx.close()
    del slab
    torch.cuda.empty_cache()


def test_corpus_is_the_reference_recipe_at_both_ends(env, oracle):
    # rows 0..255 and the last 256 rows of the 50M-row slab against the oracle's restatement of the bench generator
    slab = env["slab"]
    head = slab[:256].view(env["torch"].int16).cpu().numpy().view(np.uint16)
    tail = slab[N - 256:].view(env["torch"].int16).cpu().numpy().view(np.uint16)
    assert np.array_equal(head, oracle.clustered_corpus_f16(0, 256, DIM))
    assert np.array_equal(tail, oracle.clustered_corpus_f16(N - 256, 256, DIM))


def test_needles_at_both_ends_and_rescoring_agreement(env, oracle):
    torch, slab, idx = env["torch"], env["slab"], env["idx"]
    q = env["queries"][7].cpu().numpy()
    pos = np.array([0, 1, 24_999_999, 25_000_000, N - 2, N - 1])
    coef = np.array([1.5, 1.2, 1.45, 1.25, 1.3, 1.4])
    where = torch.from_numpy(pos).to(slab.device)
    saved = slab[where].clone()
    planted = np.stack([(q * c).astype(np.float16) for c in coef])
    slab[where] = torch.from_numpy(planted).to(slab.device)
    try:
        rows, scores, counts = idx.search_batch(q, 6)
        order = np.argsort(-coef)
        assert rows[0].tolist() == pos[order].tolist()
        want = [oracle.dot_f16_f32(planted[i].view(np.uint16), q) for i in order]
        assert np.array_equal(bits(scores[0]), bits(want))
        # the batched matrix-core path sees the same needles (the last row sits in a ragged tile of the wide pass)
        many = np.repeat(q[None, :], 260, axis=0)
        br, bs, bc, _ = idx.search_batched(many, 6)
        assert np.array_equal(br[0], rows[0]) and np.array_equal(br[259], rows[0]) and np.array_equal(bits(bs[131]), bits(scores[0]))
    finally:
        slab[where] = saved
    for k in (1, 10, 100):
        rows, scores, counts = idx.search_batch(env["queries"][:3].cpu().numpy(), k)
        assert np.all(counts == k) and np.all(np.diff(scores, axis=1) <= 0)
        for qi in range(3):
            assert len(set(rows[qi].tolist())) == k
            assert np.array_equal(bits(idx.gather_dot(env["queries"][qi].cpu().numpy(), rows[qi])), bits(scores[qi]))


def test_filtered_subset_of_200k_rows_matches_the_oracle(env, oracle):
    torch, slab, idx = env["torch"], env["slab"], env["idx"]
    q = env["queries"].cpu().numpy()
    rng = np.random.default_rng(50)
    sel = np.sort(rng.choice(N, 200_000, replace=False))
    sel[0], sel[-1] = 0, N - 1
    allow = np.zeros(N, bool)
    allow[sel] = True
    host = slab[torch.from_numpy(sel).to(slab.device)].view(torch.int16).cpu().numpy().view(np.uint16)
    rows, scores, counts = idx.search_batch(q[:3], K, allow=allow)     # 1/250 of the rows: the gather path
    for qi in range(3):
        er, es = oracle.search_top_k(host, q[qi], K, nthreads=8)
        assert np.array_equal(rows[qi], sel[er]) and np.array_equal(bits(scores[qi]), bits(es))


def test_batched_equals_per_query_on_a_sample(env):
    idx = env["idx"]
    q = env["queries"].cpu().numpy()         # 300 queries: one 256-query wide pass + a 64-query tail
    br, bs, bc, fb = idx.search_batched(q, K)
    assert np.all(bc == K) and fb < 30
    pick = [0, 1, 63, 64, 127, 128, 255, 256, 257, 299]
    er, es, _ = idx.search_batch(q[pick], K)
    assert np.array_equal(br[pick], er) and np.array_equal(bits(bs[pick]), bits(es))
    r8, s8, c8, _ = idx.search_int8_two_pass_batched(q[:260], K, 3)
    for qi in (0, 255, 259):
        hits = idx.search_top_k_int8_two_pass(q[qi], K, 3)
        assert [h.index for h in hits] == r8[qi].tolist() and np.array_equal(bits([h.score for h in hits]), bits(s8[qi]))


def test_config5_batch256_minilm_encode_then_scan(env, oracle):
    """256 token-id queries (lengths 8..32 incl. [CLS] / [SEP]) -> MiniLM-L6-shaped encoder on the GPU -> batched
    scan of the 50M rows: unit-norm embeddings (within the encoder tolerance of the f32 oracle on a few of them) and hits
    that equal the exact per-query search on those same embeddings."""
    fa, idx = env["fa"], env["idx"]
    from frankensearch_amd.synthetic import random_bert_weights
    from oracle import bert_oracle

    w = random_bert_weights(1, 30522, 384, 6, 1536)
    bert = fa.NativeEmbedder(w, device=0)
    rng = np.random.default_rng(5)
    batch = [[101] + rng.integers(1000, 30000, int(rng.integers(6, 31))).tolist() + [102] for _ in range(256)]
    emb = bert.embed_batch_token_ids(batch)
    assert emb.shape == (256, 384) and np.allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-4)
    ref = bert_oracle.embed_forward(w, [batch[0], batch[100], batch[255]], 6)
    for j, i in enumerate((0, 100, 255)):
        assert float(np.dot(ref[j], emb[i])) >= 0.999 and np.max(np.abs(ref[j] - emb[i])) <= 2e-3   # f32 oracle, seeded random weights
    br, bs, bc, fb = idx.search_batched(emb, K)
    assert np.all(bc == K)
    er, es, _ = idx.search_batch(emb[[0, 77, 255]], K)
    assert np.array_equal(br[[0, 77, 255]], er) and np.array_equal(bits(bs[[0, 77, 255]]), bits(es))
