#!/usr/bin/env python3
"""Randomised batches through the GPU encoders against the oracle (run on the GPU box): Model2Vec bit-exact, the
MiniLM-class BERT within the tolerance of tests/test_gpu_bert.py.  Batch shapes are drawn to hit every GEMM shape, the
short- and long-document attention kernels and the fused / unfused residual+LayerNorm paths."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frankensearch_amd as fa
from oracle import oracle, bert_oracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
t_end = time.time() + budget
COS_MIN, ABS_MAX = 0.999, 2e-3
bad = cases = 0
# Model2Vec
table = rng.standard_normal((5000, 256)).astype(np.float32)
m2v = fa.Model2VecEmbedder(table)
for _ in range(200):
    n = int(rng.integers(1, 40))
    texts = [rng.integers(0, 5400, int(rng.integers(0, 60))).tolist() for _ in range(n)]  # some ids out of vocabulary
    got = m2v.embed_batch_token_ids(texts)
    want = np.stack([oracle.m2v_embed(table, t) for t in texts])
    cases += 1
    if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
        bad += 1
        print("m2v mismatch", n, flush=True)
# BERT (MiniLM shape, small vocabulary)
w = bert_oracle.random_weights(31 + seed, 1500, 384, 6, 1536)
bert = fa.NativeEmbedder(w)
while time.time() < t_end:
    kind = int(rng.integers(0, 4))
    if kind == 0:
        lens = rng.integers(1, 33, int(rng.integers(1, 40)))          # query-like
    elif kind == 1:
        lens = rng.integers(60, 513, int(rng.integers(1, 4)))         # long documents
    elif kind == 2:
        lens = np.concatenate([rng.integers(1, 20, 6), rng.integers(200, 400, 1)])  # ragged mix
    else:
        lens = rng.integers(0, 3, int(rng.integers(1, 6)))            # empty / one-token texts
    batch = [rng.integers(1, 1500, int(n)).tolist() for n in lens]
    got = bert.embed_batch_token_ids(batch)
    want = bert_oracle.embed_forward(w, batch, 6)
    cases += 1
    nz = np.linalg.norm(want, axis=1) > 0
    ok = got.shape == want.shape and np.max(np.abs(got - want)) <= ABS_MAX and \
        np.all(np.sum(got[nz] * want[nz], axis=1) >= COS_MIN) and np.all(got[~nz] == 0)
    if not ok:
        bad += 1
        print("bert mismatch", kind, lens.tolist(), float(np.max(np.abs(got - want))), flush=True)
print(f"seed={seed}: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
