"""Batched exact search at 10M x 384 with tombstones / an allow bitmap against the unfiltered rate (the wide main pass consults the
bitmaps only for rows that pass their threshold)."""
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
fa_out = run("allow bitmap, 50 % of the rows", allow)
live = rng.random(rows) >= 0.01
index.set_live(live)
tl = run("tombstones, 1 % of the rows")
both = run("tombstones + allow bitmap", allow)
# spot check: filtered answers only contain allowed, live rows
r = both[0]
ok = bool(np.all(allow[r[r != 0xFFFFFFFF]]) and np.all(live[r[r != 0xFFFFFFFF]]))
print("filtered hits are allowed and live:", ok)
