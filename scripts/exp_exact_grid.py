#!/usr/bin/env python3
"""Exact single-query latency against the scan grid (FSGPU_GRID_BLOCKS) for a slab of DIM dimensions."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, frankensearch_amd as fa
dev = torch.device("cuda", 0)
DIM = int(os.environ.get("DIM", "256"))
slab = bench.gen_corpus(0, 10_000_000, DIM, dev)
idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), 10_000_000, DIM, device=0, keepalive=slab)
q = bench.gen_queries(16, DIM, dev).cpu().numpy()
def p50(f):
    for i in range(5): f(i)
    t = []
    for i in range(40):
        t0 = time.perf_counter(); f(i); t.append(time.perf_counter() - t0)
    return sorted(t)[20] * 1e3
floor = 10_000_000 * DIM * 2 / 8e12 * 1e3
print(f"dim={DIM} grid={os.environ.get('FSGPU_GRID_BLOCKS')}: k=10 {p50(lambda i: idx.search_batch(q[i % 16], 10)):.3f} ms, "
      f"k=30 {p50(lambda i: idx.search_batch(q[i % 16], 30)):.3f} ms, 2 queries {p50(lambda i: idx.search_batch(q[:2], 10)):.3f} ms (floor {floor:.3f} ms at 8 TB/s)")
