#!/usr/bin/env python3
"""One rank's work at N-way sharding (device-resident queries, batched path): wall time per 1024-query step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, frankensearch_amd as fa
from frankensearch_amd.sharded import GpuShardBackend
dev = torch.device("cuda", 0)
dim = 384
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = 10_000_000 // shards
q = bench.gen_queries(1024, dim, dev)
slab = bench.gen_corpus(0, rows, dim, dev)
idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, dim, device=0, keepalive=slab)
be = GpuShardBackend(idx, dev, batched=True)
for _ in range(3):
    be.search_packed(q, 10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    be.search_packed(q, 10)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"shards={shards} rows={rows}: {dt*1e3:.3f} ms per 1024 queries, fallbacks={be.last_fallbacks}; "
      f"floor {rows*dim*2*8/8e12*1e3:.3f} ms (8 passes at 8 TB/s)")
# the merge every rank runs on the all-gathered lists ([W, B, k] packed hits)
local = be.search_packed(q, 10)
gathered = torch.stack([local] * shards).contiguous()
for _ in range(3):
    be.merge(gathered, 10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    be.merge(gathered, 10)
torch.cuda.synchronize()
print(f"merge of {shards} x 1024 x 10 packed hits: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
