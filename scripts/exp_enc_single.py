import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights
rng = np.random.default_rng(0)
w = random_bert_weights(1, 30522, 384, 6, 1536)
bert = fa.NativeEmbedder(w)
q = [101] + rng.integers(1000, 30000, 14).tolist() + [102]
for _ in range(5): bert.embed_token_ids(q)
t0 = time.perf_counter()
for _ in range(200): bert.embed_token_ids(q)
print("single ms", (time.perf_counter() - t0) / 200 * 1e3)
