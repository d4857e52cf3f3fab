import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import frankensearch_amd as fa
S = fa.NativeShardedIndex
def bits(a): return np.ascontiguousarray(a).view(np.uint32)
for seed, n, dim, noise, ncl, nq, k in [(1, 60000, 384, 0.05, 48, 300, 30), (2, 60000, 384, 0.02, 48, 300, 33), (3, 120000, 256, 0.05, 16, 520, 64), (4, 60000, 128, 0.03, 8, 300, 30),
                                        (5, 200000, 384, 0.04, 24, 520, 33), (6, 90000, 256, 0.02, 4, 300, 10), (7, 150000, 384, 0.08, 48, 520, 64)]:
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((ncl, dim)).astype(np.float32)
    x = cent[rng.integers(0, ncl, n)] + (rng.standard_normal((n, dim)) * noise).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-9
    slab = x.astype(np.float16).view(np.uint16)
    q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.15).astype(np.float32)
    whole = fa.VectorIndex.from_slab(slab)
    ref = [np.concatenate(z) for z in zip(*[whole.search_batch(q[s0:s0 + 64], k, exact=True) for s0 in range(0, nq, 64)])]
    st0 = whole.batched_filter_stats()
    r, s, c, fb = whole.search_batched(q, k)
    st1 = whole.batched_filter_stats()
    ok_u = np.array_equal(r, ref[0]) and np.array_equal(bits(s), bits(ref[1]))
    out = []
    for g, sh in ((1, 1), (1, 2), (2, 2)):
        idx = S.from_slab(slab, [0] * (g * sh), exchange=S.EXCHANGE_PEER_COPY, query_groups=g)
        r2, s2, c2, fb2 = idx.search(q, k, S.BATCHED)
        out.append((g, sh, bool(np.array_equal(r2, ref[0]) and np.array_equal(bits(s2), bits(ref[1]))), fb2))
        idx.close()
    print(f"seed {seed} n {n} dim {dim} noise {noise} clusters {ncl} nq {nq} k {k}: unsharded ok {ok_u} fb {fb} refiltered {st1['refiltered_f16'] - st0['refiltered_f16']} | sharded {out}")
    whole.close()
