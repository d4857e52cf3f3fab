"""Tuning aid: 32 documents x 512 tokens through the MiniLM-class encoder (for rocprofv3 --kernel-trace runs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights
rng = np.random.default_rng(0)
D, L = int(os.environ.get("D", "32")), int(os.environ.get("L", "512"))
bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536))
flat = np.concatenate([np.asarray([101] + rng.integers(1000, 30000, L - 2).tolist() + [102], dtype=np.int32) for _ in range(D)])
offs = (np.arange(D + 1) * L).astype(np.uint32)
out = np.empty((D, 384), dtype=np.float32)
for _ in range(3): bert.embed_flat(flat, offs, out)
t0 = time.perf_counter(); n = 20
for _ in range(n): bert.embed_flat(flat, offs, out)
print(f"bert {D} docs x {L} tokens: {(time.perf_counter()-t0)/n*1e3:.3f} ms")
