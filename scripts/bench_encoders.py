"""Encoder micro-benchmarks (tuning aid): Model2Vec pool and the MiniLM-class BERT forward, batch of 256 queries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights

rng = np.random.default_rng(0)
B = int(os.environ.get("B", "256"))
# potion-multilingual-128M shape: 500,353 x 256 f32 (model_manifest.rs:1407-1421)
table = rng.standard_normal((500_353, 256)).astype(np.float32)
m2v = fa.Model2VecEmbedder(table)
qs = [rng.integers(0, 500_353, int(rng.integers(4, 24))).tolist() for _ in range(B)]
for _ in range(3): m2v.embed_batch_token_ids(qs)
t0 = time.perf_counter(); n = 20
for _ in range(n): m2v.embed_batch_token_ids(qs)
dt = (time.perf_counter() - t0) / n
print(f"m2v batch {B}: {dt*1e3:.3f} ms/batch  ({B/dt:.0f} texts/s); single:", end=" ")
t0 = time.perf_counter()
for _ in range(50): m2v.embed_token_ids(qs[0])
print(f"{(time.perf_counter()-t0)/50*1e3:.3f} ms")

w = random_bert_weights(1, 30522, 384, 6, 1536)
bert = fa.NativeEmbedder(w)
batch = [[101] + rng.integers(1000, 30000, int(rng.integers(6, 31))).tolist() + [102] for _ in range(B)]
tokens = sum(len(b) for b in batch)
for _ in range(3): bert.embed_batch_token_ids(batch)
t0 = time.perf_counter(); n = 20
for _ in range(n): bert.embed_batch_token_ids(batch)
dt = (time.perf_counter() - t0) / n
flops = tokens * 21.23e6 + sum(6 * 4 * len(b) * len(b) * 384 for b in batch)
print(f"bert batch {B} ({tokens} tokens): {dt*1e3:.3f} ms/batch ({B/dt:.0f} texts/s, {flops/dt/1e12:.2f} TFLOP/s); single:", end=" ")
t0 = time.perf_counter()
for _ in range(50): bert.embed_token_ids(batch[0])
print(f"{(time.perf_counter()-t0)/50*1e3:.3f} ms")
long = [[101] + rng.integers(1000, 30000, 510).tolist() + [102] for _ in range(32)]
for _ in range(2): bert.embed_batch_token_ids(long)
t0 = time.perf_counter()
for _ in range(5): bert.embed_batch_token_ids(long)
dt = (time.perf_counter() - t0) / 5
tok = 32 * 512
flops = tok * 21.23e6 + 32 * 6 * 4 * 512 * 512 * 384
print(f"bert 32 docs x 512 tokens: {dt*1e3:.3f} ms ({flops/dt/1e12:.2f} TFLOP/s)")
