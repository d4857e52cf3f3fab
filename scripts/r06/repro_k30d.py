import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import frankensearch_amd as fa
S = fa.NativeShardedIndex
rng = np.random.default_rng(10405)
dim, n, nq = 384, 42981, 600
x = rng.standard_normal((n, dim)).astype(np.float32)
x[:, rng.integers(0, dim, 3)] *= 12.0
x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-9
slab = x.astype(np.float16).view(np.uint16)
q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.15).astype(np.float32)
w0 = fa.VectorIndex.from_slab(slab)
for k in (30, 10):
    ref = [np.concatenate(z) for z in zip(*[w0.search_batch(q[s0:s0 + 64], k, exact=True) for s0 in range(0, nq, 64)])]
    for m in (256, 300, 512, 520, 600):
        w = fa.VectorIndex.from_slab(slab)
        r, s, c, fb = w.search_batched(q[:m], k)
        bad = [i for i in range(m) if not (np.array_equal(r[i], ref[0][i]) and np.array_equal(s[i].view(np.uint32), ref[1][i].view(np.uint32)))]
        miss_hi = sum(1 for i in bad for rr in set(ref[0][i].tolist()) - set(r[i].tolist()) if rr >= 32768)
        miss_lo = sum(1 for i in bad for rr in set(ref[0][i].tolist()) - set(r[i].tolist()) if rr < 32768)
        print(f"unsharded fresh index k {k} nq {m}: {len(bad)} bad (first {bad[:6]}), missing rows >= 32768: {miss_hi}, below: {miss_lo}, fb {fb}")
        w.close()
