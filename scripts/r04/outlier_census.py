"""The anisotropic / Zipf-cluster corpus of bench.py at full size under the int8 filter: per batch the fallback census of the batched
path (FSGPU_DEBUG_BATCHED lines on stderr), filter stats, throughput.   python scripts/r04/outlier_census.py [steps]"""
import os, sys, time
os.environ["FSGPU_DEBUG_BATCHED"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import frankensearch_amd as fa
from frankensearch_amd.sharded import GpuShardBackend
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
rows, dim, k, B = 10_000_000, 384, 10, 1024
slab = bench.gen_outlier_corpus(0, rows, dim, dev)
q = bench.adversarial_queries("outlier", slab, 2 * B, dev)
idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, dim, device=0, keepalive=slab)
be = GpuShardBackend(idx, dev, batched=True)
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    be.search_batched(q[(i % 2) * B:(i % 2) * B + B], k)
    torch.cuda.synchronize()
    st = idx.batched_filter_stats()
    print(f"step {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms  fallbacks {be.last_fallbacks}  filter {st}", flush=True)
