import os, sys
sys.path.insert(0, "/root/repo")
os.environ["FSGPU_DEBUG_BATCHED"] = "1"
import numpy as np
import frankensearch_amd as fa
rng = np.random.default_rng(211)
n, dim = 150_001, 384
cent = rng.standard_normal((24, dim)).astype(np.float32); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
rows = cent[rng.integers(0, 24, n)] + 0.3 * rng.uniform(-1, 1, (n, dim)).astype(np.float32)
rows /= np.linalg.norm(rows, axis=1, keepdims=True)
slab = rows.astype(np.float16).view(np.uint16)
idx = fa.VectorIndex.from_slab(slab)
q = cent[rng.integers(0, 24, 300)] + 0.3 * rng.uniform(-1, 1, (300, dim)).astype(np.float32)
for bits, fn in ((8, idx.search_int8_two_pass_batched), (4, idx.search_4bit_two_pass_batched)):
    for k, mult in ((10, 5), (10, 1)):
        r = fn(q, k, mult)
        print("bits", bits, "k", k, "mult", mult, "fallbacks", r[3], flush=True)
