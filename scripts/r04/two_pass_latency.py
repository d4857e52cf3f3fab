"""Single-query p50 of the two-pass searches through the host-pointer ABI (int8 / 4-bit, multiplier 3; k 10 and 30), hits against
the oracle-checked batched form's.   python scripts/r04/two_pass_latency.py [rows dim ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import frankensearch_amd as fa
import bench

dev = torch.device("cuda", 0)
args = [int(a) for a in sys.argv[1:]] or [10_000_000, 256, 10_000_000, 384, 1_000_000, 384]
for rows, dim in zip(args[0::2], args[1::2]):
    slab = bench.gen_corpus(0, rows, dim, dev)
    q = bench.gen_queries(64, dim, dev).cpu().numpy()
    idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, dim, device=0, keepalive=slab)
    for name, fn in (("int8", idx.search_top_k_int8_two_pass), ("4-bit", idx.search_top_k_4bit_two_pass)):
        for k in (10, 30):
            for i in range(20):
                fn(q[i % 64], k, 3)
            lat = []
            for i in range(200):
                t0 = time.perf_counter()
                fn(q[i % 64], k, 3)
                lat.append((time.perf_counter() - t0) * 1e3)
            lat.sort()
            print(f"{rows} x {dim}, {name} two-pass, k {k} (x3 candidates): p50 {lat[100]:.4f} ms  p10 {lat[20]:.4f}  p90 {lat[180]:.4f}", flush=True)
    idx.close()
    del slab
