#!/usr/bin/env python3
"""Per-rank step time of `bench.py --gpus N` rehearsed on ONE GPU: a shard-sized slab (default 1.25M rows = one of eight),
1,024 queries per step, the launcher's loop (scan of step i, then the RCCL all-gather + merge of step i on a side stream
underneath the scan of step i + 1) over a one-rank RCCL group — against the same loop without the exchange."""
import os, socket, sys, time
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import frankensearch_amd as fa
from frankensearch_amd.sharded import GpuShardBackend, ShardedVectorIndex

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
device = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
rows, B, k, steps = int(os.environ.get("ROWS", 1_250_000)), 1024, 10, 100
slab = bench.gen_corpus(0, rows, 384, device)
queries = bench.gen_queries(2 * B, 384, device)
index = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, 384, device=0, keepalive=slab)
for label, force in (("no exchange", 0), ("all-gather + merge enqueued between two scans", 1),
                     ("all-gather + merge enqueued from inside the next scan call (bench.py)", 2)):
    sh = ShardedVectorIndex(GpuShardBackend(index, device, batched=True), overlap=bool(force), force_collective=bool(force))
    batch_of = lambda i: queries[(i % 2) * B:(i % 2) * B + B]
    def run(n):
        if force == 2:
            sh.search_steps(batch_of, 0, n, k); return
        pending = None
        for i in range(n):
            qb = batch_of(i)
            if not force:
                sh.search(qb, k); continue
            local = sh.search_begin(qb, k)
            if pending is not None: pending[3].synchronize()
            pending = sh.search_end(local, k)
        if pending is not None: pending[3].synchronize()
    run(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{label}: {dt*1e3:.3f} ms per step = {B/dt:,.0f} queries/s per rank-step", flush=True)
dist.destroy_process_group()
