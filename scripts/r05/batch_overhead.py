#!/usr/bin/env python3
"""What one coalesced batch costs through the sharded handle against the unsharded index (10M x 384, host-pointer C ABI): ~200 queries,
top-30 — the int8 two-pass (the fast tier's batch) and the exact batched search (the quality tier's) — p50 of 60 calls each."""
import sys, time
import numpy as np
import torch  # noqa: F401
import frankensearch_amd as fa

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = 384
gen = torch.Generator(device="cuda").manual_seed(3)
slab = torch.randn((rows, dim), device="cuda", generator=gen, dtype=torch.float16)
slab = torch.nn.functional.normalize(slab.float(), dim=1).half().contiguous()
rng = np.random.default_rng(5)
q = rng.standard_normal((200, dim)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
whole = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, dim, device=0)
S = fa.NativeShardedIndex
sh = S.from_device_slabs([0], dim, [rows], [slab.data_ptr()])


def p50(f, n=60):
    for _ in range(5):
        f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        t.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(t))


for nq in (200, 64):
    a = p50(lambda: whole.search_int8_two_pass_batched(q[:nq], 30, 3))
    b = p50(lambda: sh.search(q[:nq], 30, S.INT8_TWO_PASS, 3))
    c = p50(lambda: whole.search_batch(q[:nq], 30))
    d = p50(lambda: sh.search(q[:nq], 30, S.BATCHED))
    print(f"nq={nq}: int8 two-pass  unsharded {a:.3f} ms  sharded(1) {b:.3f} ms | exact batched  unsharded {c:.3f} ms  sharded(1) {d:.3f} ms")
