#!/usr/bin/env python3
"""Host-side phases of a search through the one-shard handle (lab build -DFSGPU_SHARDED_TIMING copied over libfsgpu.so): 200-query batches,
clustered 10M x 384 corpus, batched exact and int8 two-pass modes, against the unsharded index's blocking calls."""
import sys, time
import numpy as np
import torch
import frankensearch_amd as fa

rows, dim = 10_000_000, 384
gen = torch.Generator(device="cuda").manual_seed(3)
cent = torch.nn.functional.normalize(torch.randn((2000, dim), device="cuda", generator=gen), dim=1)
which = torch.randint(0, 2000, (rows,), device="cuda", generator=gen)
slab = torch.empty((rows, dim), device="cuda", dtype=torch.float16)
for lo in range(0, rows, 1_000_000):
    hi = min(rows, lo + 1_000_000)
    x = cent[which[lo:hi]] + 0.08 * torch.randn((hi - lo, dim), device="cuda", generator=gen)
    slab[lo:hi] = torch.nn.functional.normalize(x, dim=1).half()
q = (cent[torch.randint(0, 2000, (256,), device="cuda", generator=gen)] + 0.08 * torch.randn((256, dim), device="cuda", generator=gen))
q = torch.nn.functional.normalize(q, dim=1).cpu().numpy().astype(np.float32)
whole = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, dim, device=0)
S = fa.NativeShardedIndex
sh = S.from_device_slabs([0], dim, [rows], [slab.data_ptr()])


def p50(f, n=300):
    for _ in range(10):
        f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(t))


for nq in (192, 64):
    print(f"nq={nq} int8 two-pass: unsharded {p50(lambda: whole.search_int8_two_pass_batched(q[:nq], 30, 3)):.3f} ms", flush=True)
    print(f"nq={nq} int8 two-pass: sharded(1) {p50(lambda: sh.search(q[:nq], 30, S.INT8_TWO_PASS, 3)):.3f} ms", flush=True)
    print(f"nq={nq} exact batched: unsharded {p50(lambda: whole.search_batched(q[:nq], 30)):.3f} ms", flush=True)
    print(f"nq={nq} exact batched: sharded(1) {p50(lambda: sh.search(q[:nq], 30, S.BATCHED)):.3f} ms", flush=True)
