"""Latency of a selective filter: gathered rows vs the masked full scan (variant 6 forces the scan)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frankensearch_amd as fa
import bench

N, DIM = 10_000_000, 384
dev = torch.device("cuda", 0)
slab = bench.gen_corpus(0, N, DIM, dev)
q = bench.gen_queries(8, DIM, dev).cpu().numpy()
idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), N, DIM, device=0, keepalive=slab)
rng = np.random.default_rng(1)
for allowed in (100, 1000, 8000, 50_000, 199_000):
    allow = np.zeros(N, bool)
    allow[rng.choice(N, allowed, replace=False)] = True
    words = fa.index.pack_bitmap(allow)
    out = {}
    for variant in (0, 6):
        idx.set_variant(variant)
        ref = idx.search_batch(q[0], 10, allow=words)
        t = []
        for i in range(30):
            t0 = time.perf_counter()
            r = idx.search_batch(q[i % 8], 10, allow=words)
            t.append(time.perf_counter() - t0)
        out[variant] = (np.median(t[5:]) * 1e3, r)
    same = all(np.array_equal(a, b) for a, b in zip(out[0][1], out[6][1]))
    print(f"allowed={allowed:7d} gather {out[0][0]:.3f} ms  scan {out[6][0]:.3f} ms  identical={same}", flush=True)
print(idx.filter_stats())
