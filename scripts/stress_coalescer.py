#!/usr/bin/env python3
"""Concurrency stress of the coalesced entry points: many threads mix exact (k varies), int8 two-pass and Model2Vec /
BERT single-item calls on shared handles while the coalescing window changes; every answer must equal the one the same
call gives alone (run on the GPU box)."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
n, dim = 200_000, 384
x = rng.standard_normal((n, dim)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
idx = fa.VectorIndex.from_slab(x.astype(np.float16).view(np.uint16))
table = rng.standard_normal((5000, 256)).astype(np.float32)
m2v = fa.Model2VecEmbedder(table)
bert = fa.NativeEmbedder(random_bert_weights(3, 2000, 384, 6, 1536))
NQ = 400
q = x[rng.integers(0, n, NQ)] + (rng.standard_normal((NQ, dim)) * 0.2).astype(np.float32)
ks = rng.choice([1, 7, 10, 30, 64], NQ)
texts = [rng.integers(0, 5000, int(rng.integers(1, 30))).tolist() for _ in range(NQ)]
toks = [[101] + rng.integers(1000, 2000, int(rng.integers(2, 30))).tolist() + [102] for _ in range(NQ)]
shared = [fa.pack_bitmap(rng.random(n) > 0.5) for _ in range(3)]   # filters several callers share (same uint64 array = same pointer)
want_exact = [idx.search_batch(q[i], int(ks[i])) for i in range(NQ)]
want_filt = [idx.search_batch(q[i], int(ks[i]), allow=shared[i % 3]) for i in range(NQ)]
want_i8 = [idx.search_top_k_int8_two_pass(q[i], int(ks[i]), 3) for i in range(NQ)]
want_m2v = [m2v.embed_token_ids(texts[i]) for i in range(NQ)]
want_bert = [bert.embed_token_ids(toks[i]) for i in range(NQ)]
bad, rounds = [], 0
t_end = time.time() + budget
while time.time() < t_end:
    mb = int(rng.choice([2, 8, 64, 128, 256]))
    wait = int(rng.choice([50, 500, 5000]))
    idx.set_coalescing(mb, wait); m2v.set_coalescing(mb, wait); bert.set_coalescing(mb, wait)
    nthreads = int(rng.choice([3, 16, 64, 200]))
    picks = rng.integers(0, NQ, (nthreads, 6))
    kinds = rng.integers(0, 5, (nthreads, 6))
    def work(t):
        for i, kind in zip(picks[t], kinds[t]):
            try:
                if kind == 0:
                    got = idx.search_batch(q[i], int(ks[i]))
                    ok = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got, want_exact[i]))
                elif kind == 1:
                    got = idx.search_top_k_int8_two_pass(q[i], int(ks[i]), 3)
                    ok = [(h.index, np.float32(h.score).view(np.uint32)) for h in got] == \
                         [(h.index, np.float32(h.score).view(np.uint32)) for h in want_i8[i]]
                elif kind == 4:
                    got = idx.search_batch(q[i], int(ks[i]), allow=shared[i % 3])
                    ok = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got, want_filt[i]))
                elif kind == 2:
                    ok = np.array_equal(m2v.embed_token_ids(texts[i]).view(np.uint32), want_m2v[i].view(np.uint32))
                else:
                    ok = float(np.dot(bert.embed_token_ids(toks[i]), want_bert[i])) > 0.99999
                if not ok:
                    bad.append((int(kind), int(i), mb, wait, nthreads))
            except Exception as e:  # noqa: BLE001
                bad.append(("exc", repr(e)))
    ts = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for t in ts: t.start()
    for t in ts: t.join()
    rounds += 1
print(f"seed={seed}: {rounds} rounds, {len(bad)} mismatches", bad[:5])
sys.exit(1 if bad else 0)
