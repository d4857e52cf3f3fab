"""32 documents x 512 tokens through the MiniLM encoder, nothing else (the large-M kernels under a tracer / counter pass)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights

rng = np.random.default_rng(0)
bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536))
long = [[101] + rng.integers(1000, 30000, 510).tolist() + [102] for _ in range(32)]
offs = np.zeros(33, dtype=np.uint32)
offs[1:] = np.cumsum([len(b) for b in long])
ids = np.concatenate([np.asarray(b, dtype=np.int32) for b in long])
out = np.empty((32, 384), dtype=np.float32)
n = int(os.environ.get("N", "10"))
for _ in range(int(os.environ.get("WARM", "2"))):
    bert.embed_flat(ids, offs, out)
t0 = time.perf_counter()
for _ in range(n):
    bert.embed_flat(ids, offs, out)
dt = (time.perf_counter() - t0) / n
flops = 32 * 512 * 21.23e6 + 32 * 6 * 4 * 512 * 512 * 384
print(f"bert 32 docs x 512 tokens: {dt*1e3:.3f} ms ({flops/dt/1e12:.2f} TFLOP/s)")
