#!/usr/bin/env python3
"""Batched scan at the row counts one rank holds when 10M rows are sharded 1/2/4/8 ways (single GPU, one shard)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, frankensearch_amd as fa
dev = torch.device("cuda", 0)
dim = 384
q = bench.gen_queries(1024, dim, dev)
for shards in (8, 4, 2, 1):
    rows = 10_000_000 // shards
    slab = bench.gen_corpus(0, rows, dim, dev)
    idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, dim, device=0, keepalive=slab)
    qh = q.cpu().numpy()
    r = idx.search_batched(qh, 10)
    e = idx.search_batch(qh[:8], 10)
    same = np.array_equal(r[0][:8], e[0]) and np.array_equal(r[1][:8].view(np.uint32), e[1].view(np.uint32))
    t0 = time.perf_counter()
    for _ in range(10): fb = idx.search_batched(qh, 10)[3]
    dt = (time.perf_counter() - t0) / 10
    print(f"shards={shards} rows={rows} {dt*1e3:.3f} ms per 1024 queries -> {1024/dt*shards/1e3:.1f}k q/s aggregate if perfectly parallel; fallbacks={fb} exact={same}", flush=True)
    idx.close(); del slab
