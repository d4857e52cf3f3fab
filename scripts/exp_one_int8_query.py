import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench, frankensearch_amd as fa
dev = torch.device("cuda", 0)
DIM = int(os.environ.get("DIM", "256")); K = int(os.environ.get("K", "30")); BITS = int(os.environ.get("BITS", "8"))
slab = bench.gen_corpus(0, 10_000_000, DIM, dev)
idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), 10_000_000, DIM, device=0, keepalive=slab)
q = bench.gen_queries(16, DIM, dev).cpu().numpy()
for i in range(5): (idx.search_top_k_int8_two_pass(q[i], K, 3) if BITS == 8 else idx.search_top_k_4bit_two_pass(q[i], K, 5))
t=[]
for i in range(40):
    t0=time.perf_counter(); (idx.search_top_k_int8_two_pass(q[i%16], K, 3) if BITS == 8 else idx.search_top_k_4bit_two_pass(q[i%16], K, 5)); t.append(time.perf_counter()-t0)
print(f"bits={BITS} dim={DIM} k={K} p50 ms", sorted(t)[20]*1e3)
