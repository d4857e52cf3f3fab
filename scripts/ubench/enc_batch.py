"""Tuning aid: the 256-query MiniLM batch alone (for rocprofv3 --kernel-trace --stats runs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights
rng = np.random.default_rng(0)
B = int(os.environ.get("B", "256"))
bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536))
batch = [[101] + rng.integers(1000, 30000, int(rng.integers(6, 31))).tolist() + [102] for _ in range(B)]
offs = np.zeros(B + 1, dtype=np.uint32); offs[1:] = np.cumsum([len(b) for b in batch])
flat = np.concatenate([np.asarray(b, dtype=np.int32) for b in batch])
out = np.empty((B, 384), dtype=np.float32)
for _ in range(3): bert.embed_flat(flat, offs, out)
t0 = time.perf_counter(); n = 30
for _ in range(n): bert.embed_flat(flat, offs, out)
print(f"bert batch {B} ({flat.size} tokens): {(time.perf_counter()-t0)/n*1e3:.3f} ms")
