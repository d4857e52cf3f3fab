#!/usr/bin/env python3
"""Thread-count sweep of the oracle's AVX2 scan on the GPU box's host cores (picks bench.py's cpu_baseline setting)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402

oracle.build()
rng = np.random.default_rng(0)
n, dim = 2_500_000, 384
staged = (rng.standard_normal((n, dim), dtype=np.float32) / np.sqrt(dim)).astype(np.float16).view(np.uint16)
if len(sys.argv) > 1 and sys.argv[1] == "spread":
    from concurrent.futures import ThreadPoolExecutor
    slab = np.empty_like(staged)
    step = 8192
    with ThreadPoolExecutor(max_workers=64) as ex:
        list(ex.map(lambda lo: slab.__setitem__(slice(lo, lo + step), staged[lo:lo + step]), range(0, n, step)))
else:
    slab = staged
qs = rng.standard_normal((8, dim)).astype(np.float32)
for nt in (16, 32, 64, 96, 128, 192, 256):
    oracle.search_top_k(slab[:100_000], qs[0], 10, nthreads=nt)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for q in qs:
            oracle.search_top_k(slab, q, 10, nthreads=nt)
        best = min(best, (time.perf_counter() - t0) / len(qs))
    print(f"threads={nt:4d}  {best * 1e3:8.2f} ms/query  {n * dim * 2 / best / 1e9:7.1f} GB/s", flush=True)
