#!/usr/bin/env python3
"""Batched int8 two-pass: kernel shape sweep (FSGPU_MFMA_SHAPE_I8) at 10M x 384."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, frankensearch_amd as fa
dev = torch.device("cuda", 0)
rows, dim = 10_000_000, int(os.environ.get("DIM", 384))
slab = bench.gen_corpus(0, rows, dim, dev)
idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, dim, device=0, keepalive=slab)
q = bench.gen_queries(1024, dim, dev).cpu().numpy()
idx.search_int8_two_pass_batched(q, 10, 3)
idx.scan_stats(reset=True); idx.set_profiling(True)
t0 = time.perf_counter()
for _ in range(8): fb = idx.search_int8_two_pass_batched(q, 10, 3)[3]
dt = time.perf_counter() - t0
ms, n, r = idx.scan_stats(reset=True)
print(f"shape={os.environ.get('FSGPU_MFMA_SHAPE_I8')} dim={dim} qps={8*1024/dt:.0f} pass1={ms/n:.4f} ms  {r/n*dim/(ms/n*1e-3)/1e12:.3f} TB/s fallbacks={fb}")
