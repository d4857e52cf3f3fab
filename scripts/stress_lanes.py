#!/usr/bin/env python3
"""Concurrent UNCOALESCED callers on one index handle (run on the GPU box): filtered single-query searches from 1..64
threads run on the handle's lanes (own stream + workspaces each); every answer must equal the lone call's, also while another
thread keeps changing tombstones; prints the throughput per thread count."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frankensearch_amd as fa

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n, dim, k = 1_000_000, 384, 10
x = rng.standard_normal((n, dim)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
idx = fa.VectorIndex.from_slab(x.astype(np.float16).view(np.uint16))
NQ = 64
q = x[rng.integers(0, n, NQ)] + (rng.standard_normal((NQ, dim)) * 0.2).astype(np.float32)
selective = [fa.pack_bitmap(np.isin(np.arange(n), rng.choice(n, 2000, replace=False))) for _ in range(8)]   # gather path
broad = [fa.pack_bitmap(rng.random(n) > 0.5) for _ in range(4)]                                              # masked scan
def call(i):
    f = selective[i % 8] if i % 3 else broad[i % 4]
    return idx.search_batch(q[i % NQ], k, allow=f)
want = [call(i) for i in range(96)]
bad = []
for nthreads in (1, 4, 16, 64):
    per = 192 // nthreads if nthreads <= 16 else 6
    def work(t):
        for j in range(per):
            i = (t * 7 + j * 13) % 96
            got = call(i)
            if not all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got, want[i])):
                bad.append((nthreads, t, i))
    ts = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print(f"{nthreads:3d} threads: {nthreads * per / dt:8.0f} filtered searches/s", flush=True)
# searches against a moving tombstone set: a result must equal the lone call under the bitmap before OR after the change
live_a = np.ones(n, bool)
live_b = live_a.copy(); live_b[::3] = False
idx.set_live(live_a); wa = [call(i) for i in range(24)]
idx.set_live(live_b); wb = [call(i) for i in range(24)]
stop = False
def mutate():
    flip = 0
    while not stop:
        idx.set_live(live_a if flip else live_b); flip ^= 1
        time.sleep(0.002)
def reader(t):
    for j in range(30):
        i = (t + j) % 24
        got = call(i)
        same = lambda w: all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got, w))
        if not (same(wa[i]) or same(wb[i])): bad.append(("mut", t, i))
m = threading.Thread(target=mutate); m.start()
ts = [threading.Thread(target=reader, args=(t,)) for t in range(16)]
for t in ts: t.start()
for t in ts: t.join()
stop = True; m.join()
print(f"{len(bad)} mismatches", bad[:5])
sys.exit(1 if bad else 0)
