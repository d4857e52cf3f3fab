#!/usr/bin/env python3
"""MRL-128 and exact single-query latency against the scan grid (FSGPU_GRID_BLOCKS)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, frankensearch_amd as fa
dev = torch.device("cuda", 0)
slab = bench.gen_corpus(0, 10_000_000, 384, dev)
idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), 10_000_000, 384, device=0, keepalive=slab)
q = bench.gen_queries(16, 384, dev).cpu().numpy()
def p50(f):
    for i in range(5): f(i)
    t = []
    for i in range(40):
        t0 = time.perf_counter(); f(i); t.append(time.perf_counter() - t0)
    return sorted(t)[20] * 1e3
print(f"grid={os.environ.get('FSGPU_GRID_BLOCKS')}: exact k=10 {p50(lambda i: idx.search_batch(q[i % 16], 10)):.3f} ms, "
      f"exact k=30 {p50(lambda i: idx.search_batch(q[i % 16], 30)):.3f} ms, "
      f"mrl128 {p50(lambda i: idx.mrl_search(q[i % 16], 10, search_dims=128)):.3f} ms, "
      f"mrl64 {p50(lambda i: idx.mrl_search(q[i % 16], 10, search_dims=64)):.3f} ms, "
      f"mrl256 {p50(lambda i: idx.mrl_search(q[i % 16], 10, search_dims=256)):.3f} ms, "
      f"mrl32 {p50(lambda i: idx.mrl_search(q[i % 16], 10, search_dims=32)):.3f} ms")
