"""One-launch MiniLM forward (bert_docs_w.hip): time per call against the number of texts (= row blocks).  If the time is flat the
kernel is bound per CU (a block's own weight stream); if it grows with the block count below 256 blocks, by what all blocks share
(the L2s' aggregate bandwidth)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights

rng = np.random.default_rng(0)
bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536))
def flatten(batch):
    offs = np.zeros(len(batch) + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(b) for b in batch])
    return np.concatenate([np.asarray(b, dtype=np.int32) for b in batch]), offs
def blocks(batch):
    n, rows = 1, 0
    for b in batch:
        if rows + len(b) > 32:
            n, rows = n + 1, 0
        rows += len(b)
    return n
for B in (8, 16, 32, 64, 128, 192, 256, 320, 384, 512, 768, 1024):
    batch = [[101] + rng.integers(1000, 30000, int(rng.integers(6, 31))).tolist() + [102] for _ in range(B)]
    f, o = flatten(batch)
    out = np.empty((B, 384), dtype=np.float32)
    for _ in range(5): bert.embed_flat(f, o, out)
    t0 = time.perf_counter(); n = 100
    for _ in range(n): bert.embed_flat(f, o, out)
    dt = (time.perf_counter() - t0) / n
    print(f"texts {B:5d} tokens {f.size:6d} blocks {blocks(batch):4d}: {dt*1e3:.3f} ms", flush=True)
# full blocks: texts of exactly 32 tokens
for B in (64, 128, 256):
    batch = [[101] + rng.integers(1000, 30000, 30).tolist() + [102] for _ in range(B)]
    f, o = flatten(batch)
    out = np.empty((B, 384), dtype=np.float32)
    for _ in range(5): bert.embed_flat(f, o, out)
    t0 = time.perf_counter(); n = 100
    for _ in range(n): bert.embed_flat(f, o, out)
    dt = (time.perf_counter() - t0) / n
    print(f"texts {B:5d} x 32 tokens, blocks {B:4d}: {dt*1e3:.3f} ms", flush=True)
