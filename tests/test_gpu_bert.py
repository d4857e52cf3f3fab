"""GPU MiniLM-class encoder vs the f32 oracle and the transformers golden vectors.
Tolerance: cosine >= 0.999 and max-abs <= 2e-3 on unit vectors (f16 MFMA linears, f32 elsewhere)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bert_golden.npz")
COS_MIN, ABS_MAX = 0.999, 2e-3


@pytest.fixture(scope="module")
def fa():
    import frankensearch_amd as fa_mod
    from frankensearch_amd.build import build
    build()
    return fa_mod


def check(got, want):
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= ABS_MAX, np.max(np.abs(got - want))
    nz = np.linalg.norm(want, axis=1) > 0
    assert np.all(np.sum(got[nz] * want[nz], axis=1) >= COS_MIN)
    assert np.all(got[~nz] == 0)


def golden_batch(g):
    out, o = [], 0
    for n in g["batch_lens"]:
        out.append(g["batch_ids"][o:o + n].tolist())
        o += n
    return out


@pytest.mark.parametrize("name", ["tiny", "minilm_shape"])
def test_matches_transformers_golden(fa, name):
    from oracle import bert_oracle
    g = np.load(GOLD)
    seed, vocab, hidden, layers, inter = (int(x) for x in g[f"{name}_config"])
    w = bert_oracle.random_weights(seed, vocab, hidden, layers, inter)
    m = fa.NativeEmbedder(w)
    check(m.embed_batch_token_ids(golden_batch(g)), g[f"{name}_expected"])


def test_matches_oracle_on_ragged_batches(fa):
    from oracle import bert_oracle
    rng = np.random.default_rng(7)
    w = bert_oracle.random_weights(21, 2000, 384, 6, 1536)
    m = fa.NativeEmbedder(w)
    lens = [1, 2, 3, 8, 16, 17, 31, 32, 33, 63, 64, 65, 100, 128, 200, 0, 5]
    batch = [[101] + rng.integers(1000, 2000, max(n - 2, 0)).tolist() + ([102] if n > 1 else []) if n else [] for n in lens]
    batch = [b[:n] for b, n in zip(batch, lens)]
    got = m.embed_batch_token_ids(batch)
    want = bert_oracle.embed_forward(w, batch, 6)
    check(got, want)
    assert np.allclose(np.linalg.norm(got[[i for i, n in enumerate(lens) if n]], axis=1), 1.0, atol=1e-4)
    # single == batch (native_embedder.rs:308-333)
    single = m.embed_token_ids(batch[9])
    assert np.sum(single * got[9]) > 0.99999
    # all-empty batch -> zeros
    assert np.all(m.embed_batch_token_ids([[], []]) == 0)


def test_max_length_512_and_errors(fa):
    from oracle import bert_oracle
    rng = np.random.default_rng(9)
    w = bert_oracle.random_weights(22, 500, 128, 2, 512)
    m = fa.NativeEmbedder(w)
    long = [101] + rng.integers(1, 500, 510).tolist() + [102]
    got = m.embed_batch_token_ids([long, long[:300]])
    check(got, bert_oracle.embed_forward(w, [long, long[:300]], 2))
    with pytest.raises(fa.InvalidConfig):
        m.embed_token_ids(long + [5])          # > 512 tokens: caller must truncate (native.rs:45)
    with pytest.raises(fa.InvalidConfig):
        m.embed_token_ids([101, 9999, 102])    # id outside the vocabulary


def test_batch_256_queries_config5_shape(fa):
    """BASELINE config 5 shape: 256 queries, lengths uniform 8..32 incl. [CLS]=101/[SEP]=102."""
    from oracle import bert_oracle
    rng = np.random.default_rng(11)
    w = bert_oracle.random_weights(23, 30522, 384, 6, 1536)
    m = fa.NativeEmbedder(w)
    batch = [[101] + rng.integers(1000, 30000, int(rng.integers(6, 31))).tolist() + [102] for _ in range(256)]
    got = m.embed_batch_token_ids(batch)
    want = bert_oracle.embed_forward(w, batch[:24], 6)
    check(got[:24], want)


@pytest.mark.parametrize("hidden,inter,layers", [(256, 1024, 2), (128, 384, 2), (384, 1280, 1)])
def test_batch_path_other_widths(fa, hidden, inter, layers):
    """The fragment-order batch kernels (bert_gemm_w.hip) at the other supported widths: hidden 256 (the one-launch
    post-attention kernel with 2-tile chunks), inter 384 (not a multiple of 256: FFN up and FFN down + LayerNorm as
    separate launches), inter 1280 (chunks of two); token counts that are not multiples of the 32 / 64-row tiles."""
    from oracle import bert_oracle
    rng = np.random.default_rng(hidden + inter)
    w = bert_oracle.random_weights(31, 3000, hidden, layers, inter)
    m = fa.NativeEmbedder(w)
    lens = [int(x) for x in rng.integers(3, 40, 40)] + [1, 129]
    batch = [[101] + rng.integers(1000, 3000, n - 1).tolist() for n in lens]
    assert sum(lens) > 256 and sum(lens) % 32 != 0
    got = m.embed_batch_token_ids(batch)
    check(got, bert_oracle.embed_forward(w, batch, layers))
    # a text alone (query path or small-batch kernels) agrees with the same text inside the batch
    for i in (0, 7, 41):
        assert float(np.sum(m.embed_token_ids(batch[i]) * got[i])) > 0.9999


def test_large_batch_equals_its_parts(fa):
    """A call too large for the pinned staging block and the graph cache (130 x 512 tokens: pageable H2D / D2H, eager
    launches) returns, text for text, the bits of the same texts embedded 26 at a time: every row's arithmetic is independent
    of how many rows share the launch."""
    from frankensearch_amd.synthetic import random_bert_weights
    rng = np.random.default_rng(17)
    m = fa.NativeEmbedder(random_bert_weights(3, 3000, 384, 2, 1536))
    docs = [[101] + rng.integers(1000, 3000, int(n) - 2).tolist() + [102] for n in rng.integers(400, 513, 130)]
    whole = m.embed_batch_token_ids(docs)
    parts = np.concatenate([m.embed_batch_token_ids(docs[i:i + 26]) for i in range(0, 130, 26)])
    assert np.array_equal(whole.view(np.uint32), parts.view(np.uint32))
    assert np.allclose(np.linalg.norm(whole, axis=1), 1.0, atol=1e-4)


def test_outlier_channels_heavy_tails_and_512_token_documents(fa):
    """Weights with the statistics of TRAINED checkpoints (oracle.bert_oracle.heavy_tailed_weights: outlier channels with
    LayerNorm gains of 8-20, Student-t weights, activations of +-50 next to a median of 0.6): the f16 operand tiles and the f16
    intermediate tile of the batch kernels, the query path and 512-token documents must hold the same tolerance."""
    from oracle import bert_oracle
    rng = np.random.default_rng(29)
    w = bert_oracle.heavy_tailed_weights(41, 3000, 384, 6, 1536)
    m = fa.NativeEmbedder(w)
    ref = bert_oracle.CForward(w, 6)
    queries = [[101] + rng.integers(1000, 3000, int(n)).tolist() + [102] for n in rng.integers(3, 30, 64)]
    check(m.embed_batch_token_ids(queries), ref.run(queries, 8))
    for i in (0, 5, 63):                                       # the <= 32-token query path, one text at a time
        check(m.embed_token_ids(queries[i])[None, :], ref.run([queries[i]], 1))
    docs = [[101] + rng.integers(1000, 3000, int(n) - 2).tolist() + [102] for n in (512, 512, 300, 129, 64, 33)]
    check(m.embed_batch_token_ids(docs), ref.run(docs, 8))
    # the numpy oracle and its C restatement agree on these weights too (the C one is what the long documents are held against)
    assert np.max(np.abs(ref.run(queries[:6], 2) - bert_oracle.embed_forward(w, queries[:6], 6))) < 1e-5


def test_one_launch_path_short_texts(fa):
    """Every text <= 32 tokens and more than 32 tokens in all: the whole forward is ONE launch (bert_docs_w.hip: a 32-row block
    owns whole texts).  Block packing edge cases — texts of exactly 32 tokens, 1-token texts (32 texts in one block), empty texts
    at the start / middle / end / a whole block of them, a text that does not fit the rest of its block — against the f32 oracle,
    and against the batch path (the same texts with a 40-token text appended, which routes the call to the multi-launch kernels)."""
    from oracle import bert_oracle
    rng = np.random.default_rng(31)
    w = bert_oracle.random_weights(27, 3000, 384, 6, 1536)
    m = fa.NativeEmbedder(w)
    ref = bert_oracle.CForward(w, 6)

    def text(n):
        if n == 0:
            return []
        if n == 1:
            return [101]
        return [101] + rng.integers(1000, 3000, n - 2).tolist() + [102]

    cases = [
        [32, 32, 32, 1],
        [1] * 70,
        [0, 0, 5, 0, 27, 6, 0, 0, 31, 2, 0],
        [0] * 40 + [20, 20] + [0] * 3,
        [17, 16, 15, 18, 32, 1, 31, 2, 30, 3, 9, 9, 9, 9, 9],
        [int(x) for x in rng.integers(0, 33, 300)],
    ]
    for lens in cases:
        batch = [text(n) for n in lens]
        assert sum(lens) > 32 and max(lens) <= 32
        got = m.embed_batch_token_ids(batch)
        check(got, ref.run(batch, 8))
        # the multi-launch batch path on the same texts (+ one 40-token text that disqualifies the call from the one-launch path)
        other = m.embed_batch_token_ids(batch + [text(40)])[:-1]
        assert np.max(np.abs(got - other)) <= 1e-3, np.max(np.abs(got - other))
        # a text's embedding does not depend on what shares its block or its call (up to the summation order of the attention's
        # matrix-core reductions, which follows the text's row offset inside its block)
        alone = m.embed_batch_token_ids([batch[-1], text(32), text(32)])[0] if lens[-1] else np.zeros(384, np.float32)
        assert np.max(np.abs(got[-1] - alone)) <= 2e-4


def test_one_launch_path_outlier_weights_and_large_calls(fa):
    """The one-launch path on the heavy-tailed weight set (f16 Q/K/V, context and intermediate tiles in LDS), and a call too large
    for the pinned staging block (4,000 texts: pageable H2D / D2H) equal to its parts."""
    from oracle import bert_oracle
    rng = np.random.default_rng(37)
    w = bert_oracle.heavy_tailed_weights(43, 3000, 384, 6, 1536)
    m = fa.NativeEmbedder(w)
    ref = bert_oracle.CForward(w, 6)
    queries = [[101] + rng.integers(1000, 3000, int(n)).tolist() + [102] for n in rng.integers(3, 31, 96)]
    check(m.embed_batch_token_ids(queries), ref.run(queries, 8))
    many = [[101] + rng.integers(1000, 3000, int(n)).tolist() + [102] for n in rng.integers(1, 31, 4000)]
    whole = m.embed_batch_token_ids(many)
    parts = np.concatenate([m.embed_batch_token_ids(many[i:i + 250]) for i in range(0, 4000, 250)])
    # (a text's row offset inside its block differs between the two packings, and with it the summation order of the attention's
    # matrix-core reductions: equal up to that)
    assert np.max(np.abs(whole - parts)) <= 2e-4
    again = m.embed_batch_token_ids(many)
    assert np.array_equal(whole.view(np.uint32), again.view(np.uint32))
    check(whole[:16], ref.run(many[:16], 8))


def test_model_file_blob_gives_the_same_embedder_as_the_tensor_struct(fa, tmp_path):
    """fsgpu_bert_create_safetensors (NativeEmbedder::load -> parse_weights, native.rs:1359-1602): a safetensors file in the bare
    sentence-transformers key layout and one in the `bert.`-prefixed cross-encoder layout (with position_ids, pooler tensors and
    metadata in them) give bit for bit the embeddings of fsgpu_bert_create over the same tensors."""
    from safetensors.numpy import save_file
    from oracle import bert_oracle

    w = bert_oracle.random_weights(11, 700, 128, 2, 512)
    ref = fa.NativeEmbedder(w)
    texts = [[101] + list(range(10, 40)) + [102], [101, 7, 8, 102], list(range(100, 160)), []]
    want = ref.embed_batch_token_ids(texts)
    for prefix in ("", "bert."):
        tensors = {}
        for k, v in w.items():
            bare = k[len("bert."):] if k.startswith("bert.") else k
            tensors[prefix + bare] = np.ascontiguousarray(v, dtype=np.float32)
        tensors[prefix + "embeddings.position_ids"] = np.arange(512, dtype=np.int64)[None, :]
        tensors[prefix + "pooler.dense.weight"] = np.zeros((128, 128), np.float32)
        path = str(tmp_path / f"model_{len(prefix)}.safetensors")
        save_file(tensors, path, metadata={"format": "pt"})
        enc = fa.NativeEmbedder.from_safetensors(path)
        assert enc.dimension() == 128
        got = enc.embed_batch_token_ids(texts)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), prefix
        enc.close()
        # the same file with a header whose length is not a multiple of 4 (older writers do not pad it): the tensor bytes sit at odd
        # addresses; parse_weights decodes with f32::from_le_bytes at any alignment, the library stages such tensors
        blob = open(path, "rb").read()
        hlen = int.from_bytes(blob[:8], "little")
        header = blob[8:8 + hlen].rstrip(b" ")
        header = header[:-1] + b" " * ((1 - len(header)) % 4) + b"}"   # length = 1 (mod 4), still one JSON object
        assert len(header) % 4 == 1
        odd = len(header).to_bytes(8, "little") + header + blob[8 + hlen:]
        enc = fa.NativeEmbedder.from_safetensors_bytes(odd)
        got = enc.embed_batch_token_ids(texts)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), prefix + " (unaligned tensor data)"
        enc.close()
    ref.close()


def test_index_build_batches_of_more_than_8k_tokens_take_the_64_row_blocks(fa):
    """Above 8,192 tokens (more 32-row blocks than one round of the chip's CUs) the post-attention block runs as 64-row blocks
    (bert_ffn_w64_kernel: half the weight stream per token).  Same tolerance against the f32 C oracle for every hidden size the
    kernel is built for, on benign and heavy-tailed weights; ragged lengths, a row count that is not a multiple of 64."""
    from oracle import bert_oracle
    rng = np.random.default_rng(57)
    cases = [(bert_oracle.random_weights(9, 3000, 384, 6, 1536), 6), (bert_oracle.heavy_tailed_weights(47, 3000, 384, 3, 1536), 3),
             (bert_oracle.random_weights(10, 3000, 256, 2, 1024), 2), (bert_oracle.random_weights(12, 3000, 128, 2, 512), 2)]
    for w, layers in cases:
        m = fa.NativeEmbedder(w)
        ref = bert_oracle.CForward(w, layers)
        lens = [512] * 14 + [int(n) for n in rng.integers(40, 400, 12)] + [33, 77]   # 7,168 + ~2,600 + 110 tokens
        docs = [[101] + rng.integers(1000, 3000, n - 2).tolist() + [102] for n in lens]
        assert sum(lens) > 8192 and sum(lens) % 64 != 0
        got = m.embed_batch_token_ids(docs)
        check(got, ref.run(docs, 16))
        m.close()


def test_index_build_batches_of_thousands_of_tokens(fa):
    """Calls of >= 6,144 tokens (documents of an index build, index_builder.rs:191,416) take the large-M form of the batch path: the
    weight-stationary QKV / FFN-up GEMMs (bert_gemm_wp_kernel) and the post-attention block as three launches.  Same tolerance
    against the f32 C oracle, on benign and heavy-tailed weights; ragged lengths and a row count that is not a multiple of the 64-row
    tile; a smaller call of the same documents (the fused 32-row-block kernels) agrees within the tolerance too."""
    from oracle import bert_oracle
    rng = np.random.default_rng(31)
    for w in (bert_oracle.random_weights(7, 3000, 384, 6, 1536), bert_oracle.heavy_tailed_weights(43, 3000, 384, 6, 1536)):
        m = fa.NativeEmbedder(w)
        ref = bert_oracle.CForward(w, 6)
        lens = [512] * 10 + [int(n) for n in rng.integers(40, 400, 14)] + [33, 77]   # 5,120 + ~3,000 + 110 tokens
        docs = [[101] + rng.integers(1000, 3000, n - 2).tolist() + [102] for n in lens]
        assert sum(lens) >= 6144 and sum(lens) % 64 != 0
        got = m.embed_batch_token_ids(docs)
        check(got, ref.run(docs, 16))
        small = np.concatenate([m.embed_batch_token_ids(docs[i:i + 4]) for i in range(0, len(docs), 4)])
        assert np.max(np.abs(got - small)) <= ABS_MAX and np.all(np.sum(got * small, axis=1) >= COS_MIN)
        m.close()


def test_real_weights_conformance_script_on_a_model_directory_of_the_real_layout():
    """scripts/conformance_minilm.py is what a maintainer points at the real all-MiniLM-L6-v2 files (model_manifest.rs:65-70,308-314:
    the reference pins the model by a certificate over MODEL_CONFORMANCE_TEXTS_V1).  No weights exist here, so its --selftest builds a
    directory of the same layout — model.safetensors, tokenizer.json, config.json — with seeded random weights of the architecture and runs
    every step: tokenizers -> fsgpu_bert_create_safetensors + embed (alone and as a batch) against transformers f32 on the same file."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "scripts", "conformance_minilm.py"), "--selftest"], capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "PASSED" in res.stdout and res.stdout.count(" ok") == 8, res.stdout[-2000:]
