import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights
rng = np.random.default_rng(0)
B = int(os.environ.get("B", "256"))
w = random_bert_weights(1, 30522, 384, 6, 1536)
bert = fa.NativeEmbedder(w)
batch = [[101] + rng.integers(1000, 30000, int(rng.integers(6, 31))).tolist() + [102] for _ in range(B)]
for _ in range(5): bert.embed_batch_token_ids(batch)
t0 = time.perf_counter(); n = 50
for _ in range(n): bert.embed_batch_token_ids(batch)
print(f"batch {B} ({sum(len(b) for b in batch)} tokens): {(time.perf_counter()-t0)/n*1e3:.3f} ms")
