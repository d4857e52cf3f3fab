"""Pins the numpy BERT oracle on golden vectors produced by transformers.BertModel (tests/golden/make_bert_golden.py)
and on the structural invariants the reference tests carry (native_embedder.rs:308-333: unit norm, single == batch)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bert_golden.npz")


def load_batch(g):
    lens, ids = g["batch_lens"], g["batch_ids"]
    out, o = [], 0
    for n in lens:
        out.append(ids[o:o + n].tolist())
        o += n
    return out


@pytest.mark.parametrize("name", ["tiny", "minilm_shape"])
def test_oracle_matches_transformers_golden(name):
    from oracle import bert_oracle
    g = np.load(GOLD)
    seed, vocab, hidden, layers, inter = (int(x) for x in g[f"{name}_config"])
    w = bert_oracle.random_weights(seed, vocab, hidden, layers, inter)
    got = bert_oracle.embed_forward(w, load_batch(g), layers)
    want = g[f"{name}_expected"]
    assert got.shape == want.shape
    # f32 vs f32 (different op order, A-S erf vs libm erf): tight
    assert np.max(np.abs(got - want)) < 2e-5
    assert np.all(np.sum(got * want, axis=1) > 0.999999)


def test_oracle_invariants():
    from oracle import bert_oracle
    w = bert_oracle.random_weights(3, 300, 128, 2, 512)
    batch = [[101, 5, 6, 102], [], [101, 9, 102], [101] + list(range(10, 200)) + [102]]
    out = bert_oracle.embed_forward(w, batch, 2)
    assert np.all(out[1] == 0)                                   # empty text -> zeros
    assert np.allclose(np.linalg.norm(out[[0, 2, 3]], axis=1), 1.0, atol=1e-5)   # unit norm
    single = bert_oracle.embed_forward(w, [batch[3]], 2)[0]
    assert np.sum(single * out[3]) > 0.999999                    # single == batch (native_embedder.rs:308-333)
    # bare and bert.-prefixed keys are the same model (native.rs:1466-1476)
    w2 = {("bert." + k): v for k, v in w.items()}
    assert np.array_equal(bert_oracle.embed_forward(w2, batch, 2), out)


def test_product_synthetic_generator_matches_oracle_generator():
    """frankensearch_amd.synthetic (used by bench.py) and the oracle's generator are the same function of the seed."""
    from frankensearch_amd.synthetic import random_bert_weights
    from oracle import bert_oracle
    a = random_bert_weights(9, 50, 128, 1, 256)
    b = bert_oracle.random_weights(9, 50, 128, 1, 256)
    assert a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("name", ["tiny", "minilm_shape"])
def test_c_restatement_matches_numpy_oracle_and_golden(name):
    """oracle/bert_oracle_c.c (the multi-threaded encoder CPU baseline of bench.py) is the same f32 forward: equal to the numpy
    oracle within f32 reassociation, equal to the transformers goldens within the oracle's own tolerance, whatever the thread count."""
    from oracle import bert_oracle
    g = np.load(GOLD)
    seed, vocab, hidden, layers, inter = (int(x) for x in g[f"{name}_config"])
    w = bert_oracle.random_weights(seed, vocab, hidden, layers, inter)
    batch = load_batch(g) + [[], [101, 102]]
    want = bert_oracle.embed_forward(w, batch, layers)
    c = bert_oracle.CForward(w, layers)
    for threads in (1, 3, 8):
        got = c.run(batch, threads)
        assert got.shape == want.shape
        assert np.max(np.abs(got - want)) < 5e-6, threads
        assert np.all(got[len(batch) - 2] == 0)
    gold = g[f"{name}_expected"]
    assert np.max(np.abs(c.run(load_batch(g), 2) - gold)) < 2e-5
