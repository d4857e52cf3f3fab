#!/usr/bin/env python3
"""Fuzzer of the one-handle sharded index (fsgpu_sharded_*) on ONE GPU with virtual shards: random corpora (clustered / uniform / with
outlier channels), sizes, dimensions, query-group x row-shard layouts, modes (lone exact with and without the int8 latency path, exact
batches, matrix-core batches, int8 / 4-bit two-pass lone and batched), allow bitmaps, tombstones, ragged batches, two searches in
flight — every answer must equal the UNSHARDED index's rows and f32 score bits.   python scripts/fuzz_sharded.py SEED SECONDS"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frankensearch_amd as fa

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t_end = time.time() + budget
S = fa.NativeShardedIndex
cases = bad = checks = 0


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same(got, want, what):
    global bad, checks
    checks += 1
    gr, gs, gc = got[0], got[1], got[2]
    wr, ws, wc = want[0], want[1], want[2]
    ok = np.array_equal(gc, wc)
    if ok:
        for i in range(len(wc)):
            m = int(wc[i])
            if not (np.array_equal(gr[i][:m], wr[i][:m]) and np.array_equal(bits(gs[i][:m]), bits(ws[i][:m]))):
                ok = False
                break
    if not ok:
        bad += 1
        print("MISMATCH", what, flush=True)


while time.time() < t_end:
    rng = np.random.default_rng(seed)
    seed += 1
    cases += 1
    dim = int(rng.choice([64, 128, 256, 384]))
    n = int(rng.integers(20_000, 260_000))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        cent = rng.standard_normal((48, dim)).astype(np.float32)
        x = cent[rng.integers(0, 48, n)] + (rng.standard_normal((n, dim)) * 0.08).astype(np.float32)
    elif kind == 1:
        x = rng.standard_normal((n, dim)).astype(np.float32)
    else:
        x = rng.standard_normal((n, dim)).astype(np.float32)
        x[:, rng.integers(0, dim, 3)] *= 12.0   # outlier channels: the filter copy rotates
    x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-9
    slab = x.astype(np.float16).view(np.uint16)
    groups, shards = [(1, 2), (2, 2), (3, 1), (2, 4), (1, 5), (4, 2), (1, 1), (2, 1)][int(rng.integers(0, 8))]
    live = (rng.random(n) > 0.15) if rng.random() < 0.4 else None
    whole = fa.VectorIndex.from_slab(slab, live=live)
    idx = S.from_slab(slab, [0] * (groups * shards), live=live, exchange=S.EXCHANGE_PEER_COPY, query_groups=groups)
    lat = bool(rng.integers(0, 2))
    idx.set_int8_latency(lat)
    nq = int(rng.choice([1, 2, 5, 9, 63, 130, 257, 300, 520]))
    k = int(rng.choice([1, 3, 10, 30, 33]))
    q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.15).astype(np.float32)
    tag = f"seed {seed - 1} dim {dim} n {n} kind {kind} layout {groups}x{shards} live {live is not None} lat {lat} nq {nq} k {k}"
    # exact reference answers (the exact kernels of the unsharded index)
    ref = [np.concatenate(z) for z in zip(*[whole.search_batch(q[s0:s0 + 64], k, exact=True) for s0 in range(0, nq, 64)])]
    same(idx.search(q, k, S.BATCHED), ref, tag + " batched")
    if nq <= 9:
        same(idx.search(q, k, S.EXACT), ref, tag + " exact batch")
    for i in range(min(nq, 3)):
        r = idx.search(q[i], k, S.EXACT)
        same(r, [ref[0][i:i + 1], ref[1][i:i + 1], ref[2][i:i + 1]], tag + f" lone exact q{i}")
    if rng.random() < 0.5:
        allow = rng.random(n) < float(rng.choice([0.5, 0.9, 0.05]))
        refa = [np.concatenate(z) for z in zip(*[whole.search_batch(q[s0:s0 + 64], k, allow=allow, exact=True) for s0 in range(0, nq, 64)])]
        same(idx.search(q, k, S.BATCHED, allow=allow), refa, tag + " batched + allow")
    if dim % 8 == 0 and k * 3 <= 256 and live is None:
        mult = int(rng.choice([1, 3, 5]))
        if k * mult * shards <= 1024 and k * mult <= 256:
            r2 = whole.search_int8_two_pass_batched(q, k, mult)
            same(idx.search(q, k, S.INT8_TWO_PASS, mult), r2[:3], tag + f" int8 two-pass x{mult}")
            r1 = idx.search(q[0], k, S.INT8_TWO_PASS, mult)
            same(r1, [r2[0][:1], r2[1][:1], r2[2][:1]], tag + f" lone int8 two-pass x{mult}")
            r4 = whole.search_4bit_two_pass_batched(q, k, mult)
            same(idx.search(q, k, S.FOURBIT_TWO_PASS, mult), r4[:3], tag + f" 4-bit two-pass x{mult}")
    if nq >= 9:   # two searches in flight
        h = nq // 2
        t1 = idx.search_begin(q[:h], k, S.BATCHED)
        t2 = idx.search_begin(q[h:], k, S.BATCHED)
        a = idx.search_end(t1)
        b = idx.search_end(t2)
        same(a, [ref[0][:h], ref[1][:h], ref[2][:h]], tag + " in flight 1")
        same(b, [ref[0][h:], ref[1][h:], ref[2][h:]], tag + " in flight 2")
    if nq >= 2:   # two LONE tickets in flight, and a lone ticket with a batch ticket of any kind, in either order
        one = lambda i: [ref[0][i:i + 1], ref[1][i:i + 1], ref[2][i:i + 1]]
        ta, tb = idx.search_begin(q[0], k, S.EXACT), idx.search_begin(q[1], k, S.EXACT)
        if rng.random() < 0.5:
            rb, ra = idx.search_end(tb), idx.search_end(ta)
        else:
            ra, rb = idx.search_end(ta), idx.search_end(tb)
        same(ra, one(0), tag + " lone + lone, first")
        same(rb, one(1), tag + " lone + lone, second")
        bmode = [S.EXACT, S.BATCHED][int(rng.integers(0, 2))]
        nb = min(nq, 9) if bmode == S.EXACT else nq
        if rng.random() < 0.5:
            tl, tb = idx.search_begin(q[1], k, S.EXACT), idx.search_begin(q[:nb], k, bmode)
            rl, rb = idx.search_end(tl), idx.search_end(tb)
        else:
            tb, tl = idx.search_begin(q[:nb], k, bmode), idx.search_begin(q[1], k, S.EXACT)
            rb, rl = idx.search_end(tb), idx.search_end(tl)
        same(rl, one(1), tag + f" lone beside a batch (mode {bmode})")
        same(rb, [ref[0][:nb], ref[1][:nb], ref[2][:nb]], tag + f" batch beside a lone query (mode {bmode})")
    idx.close()
    whole.close()
print(f"seed..{seed - 1}: {cases} cases, {checks} checks, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
