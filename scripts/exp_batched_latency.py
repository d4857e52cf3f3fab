#!/usr/bin/env python3
"""Latency of one host-pointer batched call (what a coalescer leader executes) at 10M rows."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import frankensearch_amd as fa  # noqa: E402

rows = int(os.environ.get("ROWS", 10_000_000))
dev = torch.device("cuda", 0)
for dim in (384, 256):
    slab = bench.gen_corpus(0, rows, dim, dev)
    idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, dim, device=0, keepalive=slab)
    q = bench.gen_queries(256, dim, dev).cpu().numpy()
    for nq in (1, 16, 80, 128, 256):
        for k in (10, 30):
            idx.search_batched(q[:nq], k)
            t = []
            fb = 0
            for i in range(10):
                t0 = time.perf_counter()
                r = idx.search_batched(q[:nq], k)
                t.append((time.perf_counter() - t0) * 1e3)
                fb += r[3]
            print(f"dim={dim} nq={nq:4d} k={k:3d}  p50={sorted(t)[5]:8.3f} ms  min={min(t):8.3f}  fallbacks/call={fb / 10:.1f}", flush=True)
    idx.close()
    del slab
