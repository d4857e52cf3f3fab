"""Batched exact search at 10M x 384 with an allow bitmap through the host-pointer ABI: the per-call bitmap (1.25 MB uploaded on every
search) against the resident one (fsgpu_allow_bitmap: uploaded once) and the unfiltered rate.  r03 verdict item 7: >= 300 k queries/s
at 50 % allowed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import frankensearch_amd as fa
dev = torch.device("cuda", 0)
rows, dim, B, k = 10_000_000, 384, 1024, 10
slab = bench.gen_corpus(0, rows, dim, dev)
index = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, dim, device=0, keepalive=slab)
q = bench.gen_queries(2 * B, dim, dev).cpu().numpy()
rng = np.random.default_rng(1)
def run(tag, allow=None, steps=20):
    for i in range(4): out = index.search_batched(q[(i % 2) * B:(i % 2) * B + B], k, allow)
    t0 = time.perf_counter()
    for i in range(steps): out = index.search_batched(q[(i % 2) * B:(i % 2) * B + B], k, allow)
    dt = (time.perf_counter() - t0) / steps
    print(f"{tag}: {B / dt:.0f} queries/s, {dt * 1e3:.3f} ms per step (host-pointer ABI), fallbacks {out[3]}", flush=True)
    return out
base = run("no filter")
allow = rng.random(rows) < 0.5
percall = run("allow bitmap, 50 % of the rows, per-call upload", allow)
f = index.resident_filter(allow)
resident = run("allow bitmap, 50 % of the rows, resident (fsgpu_allow_bitmap)", f)
print("resident == per-call hits:", bool(np.array_equal(percall[0], resident[0]) and np.array_equal(percall[1].view(np.uint32), resident[1].view(np.uint32))))
r = resident[0]
print("filtered hits are allowed:", bool(np.all(allow[r[r != 0xFFFFFFFF]])))
