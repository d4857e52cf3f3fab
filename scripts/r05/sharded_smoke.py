#!/usr/bin/env python3
"""Round 5: a quick pass over every form of the threadless sharded handle on ONE GPU (virtual shards by peer copies): layouts 1x1, 1x4,
2x2, 4x1 (query groups x row shards) against the unsharded index, bit for bit — lone exact / lone two-pass / batched / two-pass
batches / begin-end pipelining / allow bitmaps — then the two-tier flow's latencies over one shard at the given size.
usage: sharded_smoke.py [rows] [--latency]"""
import faulthandler
import sys
import time

import numpy as np

faulthandler.enable()
import torch  # noqa: E402

import frankensearch_amd as fa  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 400_000
dim, k = 384, 10
rng = np.random.default_rng(7)
slab = rng.standard_normal((rows, dim)).astype(np.float32)
slab /= np.linalg.norm(slab, axis=1, keepdims=True)
slab16 = slab.astype(np.float16)
q = rng.standard_normal((300, dim)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
whole = fa.VectorIndex.from_slab(slab16.view(np.uint16), device=0)
ref = whole.search_batch(q, k)
_i8 = whole.search_int8_two_pass_batched(q[:6], k, 3)
ref_i8 = [(_i8[0][i][:_i8[2][i]], _i8[1][i][:_i8[2][i]]) for i in range(6)]
S = fa.NativeShardedIndex
bad = 0


def same(a, b, what):
    global bad
    ok = np.array_equal(a[0], b[0]) and np.array_equal(np.asarray(a[1]).view(np.uint32), np.asarray(b[1]).view(np.uint32))
    if not ok:
        bad += 1
        print("MISMATCH", what, flush=True)
    return ok


for groups, shards in ((1, 1), (1, 4), (2, 2), (4, 1), (2, 4)):
    w = groups * shards
    idx = S.from_slab(slab16.view(np.uint16), [0] * w, exchange=S.EXCHANGE_PEER_COPY, query_groups=groups)
    assert idx.query_groups() == groups and idx.row_shards() == shards
    for lat in (False, True):
        idx.set_int8_latency(lat)
        for i in range(6):   # lone exact (certified int8 pass when lat) and lone two-pass
            r = idx.search(q[i], k, S.EXACT)
            same((r[0][0], r[1][0]), (ref[0][i], ref[1][i]), f"{groups}x{shards} lone exact lat={lat} q{i}")
            r = idx.search(q[i], k, S.INT8_TWO_PASS, 3)
            same((r[0][0][:r[2][0]], r[1][0][:r[2][0]]), (ref_i8[i][0], ref_i8[i][1]), f"{groups}x{shards} lone int8 two-pass q{i}")
    for nq in (2, 3, 7, 64, 257, 300):   # batches: exact kernels, matrix-core path, ragged query groups
        for mode in (S.EXACT, S.BATCHED):
            if mode == S.EXACT and nq > 8:
                continue
            r = idx.search(q[:nq], k, mode)
            same((r[0], r[1]), (ref[0][:nq], ref[1][:nq]), f"{groups}x{shards} nq={nq} mode={mode}")
    r = idx.search(q[:5], k, S.INT8_TWO_PASS, 3)
    for i in range(5):
        same((r[0][i][:r[2][i]], r[1][i][:r[2][i]]), (ref_i8[i][0], ref_i8[i][1]), f"{groups}x{shards} int8 two-pass batch q{i}")
    # two searches in flight
    t1 = idx.search_begin(q[:130], k, S.BATCHED)
    t2 = idx.search_begin(q[130:300], k, S.BATCHED)
    r1 = idx.search_end(t1)
    r2 = idx.search_end(t2)
    same((r1[0], r1[1]), (ref[0][:130], ref[1][:130]), f"{groups}x{shards} begin/end 1")
    same((r2[0], r2[1]), (ref[0][130:300], ref[1][130:300]), f"{groups}x{shards} begin/end 2")
    # allow bitmap
    allow = rng.random(rows) < 0.5
    ra = whole.search_batch(q[:9], k, allow=allow)
    r = idx.search(q[:9], k, S.BATCHED, allow=allow)
    same((r[0], r[1]), (ra[0], ra[1]), f"{groups}x{shards} allow batched")
    idx.close()
    print(f"layout {groups} x {shards}: done, mismatches so far {bad}", flush=True)
print("mismatches:", bad, flush=True)
if "--latency" in sys.argv:
    idx = S.from_slab(slab16.view(np.uint16), [0], exchange=S.EXCHANGE_AUTO)
    for lat in (False, True):
        idx.set_int8_latency(lat)
        whole.set_int8_latency(lat)
        for name, obj, f in (("sharded", idx, lambda i: idx.search(q[i % 64], k, S.EXACT)), ("unsharded", whole, lambda i: whole.search_batch(q[i % 64], k))):
            for i in range(20):
                f(i)
            ts = []
            for i in range(100):
                t0 = time.perf_counter()
                f(i)
                ts.append((time.perf_counter() - t0) * 1e3)
            print(f"lone exact, int8 latency {lat}: {name} p50 {sorted(ts)[50]:.4f} ms", flush=True)
    idx.close()
whole.close()
sys.exit(1 if bad else 0)
