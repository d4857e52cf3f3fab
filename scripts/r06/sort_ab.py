"""Collect-all / large-k searches (search.rs:449-473) on the library's own radix sort against the rocPRIM call of rounds 1-5
(run once per library: scripts/r06/sort_ab.sh swaps libfsgpu.so)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
import frankensearch_amd as fa

rng = np.random.default_rng(3)
for n in (1_000_000, 10_000_000):
    dim = 384
    x = rng.standard_normal((n, dim), dtype=np.float32).astype(np.float16)
    idx = fa.VectorIndex.from_slab(x.view(np.uint16))
    q = rng.standard_normal(dim).astype(np.float32)
    for k in (1000, 10_000):
        idx.search_batch(q, k)
        t = []
        for _ in range(10):
            t0 = time.perf_counter(); r = idx.search_batch(q, k); t.append(time.perf_counter() - t0)
        print(f"n={n} k={k}: p50 {1e3 * sorted(t)[5]:.3f} ms  first rows {r[0][0][:3].tolist()}", flush=True)
    idx.close()
    del x
