"""Single-query p50 through the host-pointer ABI on small slabs: the exact f16 kernels (default) against the int8 latency mode
(fsgpu_index_set_int8_latency: certified single pass over the int8 copy + exact re-score, staged path behind it).
python scripts/r04/small_n_latency.py [rows ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import frankensearch_amd as fa
import bench

dev = torch.device("cuda", 0)
for rows in [int(a) for a in sys.argv[1:]] or [1_000_000, 1_250_000]:
    slab = bench.gen_corpus(0, rows, 384, dev)
    q = bench.gen_queries(256, 384, dev).cpu().numpy()
    idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, 384, device=0, keepalive=slab)
    ref = [idx.search_batch(q[i:i + 1], 10) for i in range(64)]
    for mode in ("exact f16 kernels", "int8 latency mode"):
        idx.set_int8_latency(mode != "exact f16 kernels")
        for i in range(40):
            idx.search_batch(q[i % 64:i % 64 + 1], 10)
        lat, same = [], True
        for i in range(400):
            j = i % 64
            t0 = time.perf_counter()
            r = idx.search_batch(q[j:j + 1], 10)
            lat.append((time.perf_counter() - t0) * 1e3)
            same &= bool(np.array_equal(r[0], ref[j][0]) and np.array_equal(r[1].view(np.uint32), ref[j][1].view(np.uint32)))
        lat.sort()
        print(f"{rows} x 384, {mode}: p50 {lat[200]:.4f} ms  p10 {lat[40]:.4f}  p90 {lat[360]:.4f}  hits identical to the exact kernels: {same}", flush=True)
    idx.close()
    del slab
