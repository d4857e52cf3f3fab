import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, frankensearch_amd as fa
dev = torch.device("cuda", 0)
rows, dim = 10_000_000, int(os.environ.get("DIM", 256))
slab = bench.gen_corpus(0, rows, dim, dev)
idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, dim, device=0, keepalive=slab)
q = bench.gen_queries(1024, dim, dev).cpu().numpy()
for nq in (128, 256, 384, 1024):
    for k in (10, 30):
        idx.search_int8_two_pass_batched(q[:nq], k, 3)
        t0 = time.perf_counter()
        for _ in range(4): fb = idx.search_int8_two_pass_batched(q[:nq], k, 3)[3]
        dt = (time.perf_counter() - t0) / 4
        print(f"dim={dim} nq={nq} k={k}: {dt*1e3:.3f} ms  {nq/dt:.0f} q/s fallbacks={fb}", flush=True)
