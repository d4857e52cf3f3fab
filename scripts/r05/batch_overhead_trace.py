#!/usr/bin/env python3
"""One coalesced batch (200 queries, top-30, int8 two-pass, 10M x 384) through the unsharded index (MODE=whole) or the one-shard handle
(MODE=sharded), 40 calls: run under rocprofv3 --kernel-trace --stats to compare the kernels and copies each path issues."""
import os, sys, time
import numpy as np
import torch  # noqa: F401
import frankensearch_amd as fa

rows, dim = 10_000_000, 384
gen = torch.Generator(device="cuda").manual_seed(3)
slab = torch.randn((rows, dim), device="cuda", generator=gen, dtype=torch.float16)
slab = torch.nn.functional.normalize(slab.float(), dim=1).half().contiguous()
rng = np.random.default_rng(5)
q = rng.standard_normal((200, dim)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
mode = os.environ.get("MODE", "whole")
if mode == "whole":
    idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, dim, device=0)
    f = lambda: idx.search_int8_two_pass_batched(q, 30, 3)
else:
    S = fa.NativeShardedIndex
    idx = S.from_device_slabs([0], dim, [rows], [slab.data_ptr()])
    f = lambda: idx.search(q, 30, S.INT8_TWO_PASS, 3)
for _ in range(5):
    f()
t = []
for _ in range(40):
    t0 = time.perf_counter(); f(); t.append((time.perf_counter() - t0) * 1e3)
print(mode, "p50 %.3f ms" % float(np.median(t)))
