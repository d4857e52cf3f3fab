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
def flatten(batch, dtype):
    offs = np.zeros(len(batch) + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(b) for b in batch])
    return np.concatenate([np.asarray(b, dtype=dtype) for b in batch]), offs
# timed at the C ABI's own argument shape (flat ids + offsets, what a host holds after tokenising); the list-of-lists
# convenience wrapper spends ~0.3 ms per 256 texts in Python before the call
mf, mo = flatten(qs, np.uint32)
mout = np.empty((B, 256), dtype=np.float32)
for _ in range(3): m2v.embed_flat(mf, mo, mout)
t0 = time.perf_counter(); n = 50
for _ in range(n): m2v.embed_flat(mf, mo, mout)
dt = (time.perf_counter() - t0) / n
print(f"m2v batch {B}: {dt*1e3:.3f} ms/batch  ({B/dt:.0f} texts/s); single:", end=" ")
t0 = time.perf_counter()
for _ in range(50): m2v.embed_token_ids(qs[0])
print(f"{(time.perf_counter()-t0)/50*1e3:.3f} ms")

w = random_bert_weights(1, 30522, 384, 6, 1536)
bert = fa.NativeEmbedder(w)
batch = [[101] + rng.integers(1000, 30000, int(rng.integers(6, 31))).tolist() + [102] for _ in range(B)]
tokens = sum(len(b) for b in batch)
bf, bo = flatten(batch, np.int32)
bout = np.empty((B, 384), dtype=np.float32)
for _ in range(3): bert.embed_flat(bf, bo, bout)
t0 = time.perf_counter(); n = 50
for _ in range(n): bert.embed_flat(bf, bo, bout)
dt = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(10): bert.embed_batch_token_ids(batch)
dt_lists = (time.perf_counter() - t0) / 10
flops = tokens * 21.23e6 + sum(6 * 4 * len(b) * len(b) * 384 for b in batch)
print(f"bert batch {B} ({tokens} tokens): {dt*1e3:.3f} ms/batch ({B/dt:.0f} texts/s, {flops/dt/1e12:.2f} TFLOP/s; {dt_lists*1e3:.3f} ms through the list-of-lists wrapper); single:", end=" ")
sf, so = flatten(batch[:1], np.int32)
sout = np.empty((1, 384), dtype=np.float32)
for _ in range(5): bert.embed_flat(sf, so, sout)
t0 = time.perf_counter()
for _ in range(200): bert.embed_flat(sf, so, sout)
print(f"{(time.perf_counter()-t0)/200*1e3:.3f} ms")
long = [[101] + rng.integers(1000, 30000, 510).tolist() + [102] for _ in range(32)]
lf, lo = flatten(long, np.int32)
lout = np.empty((32, 384), dtype=np.float32)
for _ in range(30): bert.embed_flat(lf, lo, lout)   # (an index build is a long run of such calls: steady clocks)
t0 = time.perf_counter()
for _ in range(100): bert.embed_flat(lf, lo, lout)
dt = (time.perf_counter() - t0) / 100
tok = 32 * 512
flops = tok * 21.23e6 + 32 * 6 * 4 * 512 * 512 * 384
print(f"bert 32 docs x 512 tokens: {dt*1e3:.3f} ms ({flops/dt/1e12:.2f} TFLOP/s)")
