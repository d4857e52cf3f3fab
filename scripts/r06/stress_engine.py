#!/usr/bin/env python3
"""Soak of libfshost's many-queries engine: per-query callers with dynamic batching from 1..96 threads while search_many calls of random
sizes run beside them and the batching window is reconfigured; every Initial list must equal the unbatched per-query call's, every
Refined list the per-query call's up to the encoder's batch-shape tolerance (>= 97 % identical doc-id lists per round).
   python scripts/r06/stress_engine.py SECONDS [seed]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import frankensearch_amd as fa
from frankensearch_amd.host import NativeTwoTierSearcher
from frankensearch_amd.synthetic import random_bert_weights

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(seed)
n = 60_000
fast_slab = rng.standard_normal((n, 256)).astype(np.float16).view(np.uint16)
qual_slab = rng.standard_normal((n, 384)).astype(np.float16).view(np.uint16)
fast, qual = fa.VectorIndex.from_slab(fast_slab), fa.VectorIndex.from_slab(qual_slab)
m2v = fa.Model2VecEmbedder(rng.standard_normal((5000, 256)).astype(np.float32))
bert = fa.NativeEmbedder(random_bert_weights(5, 3000, 384, 6, 1536))
doc = lambda r: f"doc-{int(r):08d}"
NQ = 800
fq = [rng.integers(0, 5000, int(rng.integers(1, 24))).tolist() for _ in range(NQ)]
qq = [[101] + rng.integers(1000, 3000, int(rng.integers(2, 30))).tolist() + [102] for _ in range(NQ)]
lex = [[(doc(r), float(30 - i)) for i, r in enumerate(rng.choice(n, 30, replace=False))] for _ in range(NQ)]
t_end = time.time() + budget
rounds = bad_initial = low_final = 0
for pool in (0, 1):
    s = NativeTwoTierSearcher(fast, qual, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3, quality_pool=pool)
    want = [s.search(fq[i], qq[i], 10, lex[i]) for i in range(NQ)]
    while time.time() < t_end - (budget / 2 if pool == 0 else 0):
        rounds += 1
        nthreads = int(rng.choice([1, 3, 16, 48, 96]))
        s.set_batching(int(rng.choice([8, 64, 256])), int(rng.choice([100, 1000, 3000])))
        got = [None] * NQ
        errs = []

        def caller(tid):
            try:
                for i in range(tid, NQ, nthreads):
                    got[i] = s.search(fq[i], qq[i], 10, lex[i])
            except Exception as e:   # noqa: BLE001
                errs.append(e)

        many_out = {}

        def many_caller():
            try:
                a = int(rng.integers(0, NQ - 300))
                m = int(rng.integers(1, 300))
                many_out["range"] = (a, a + m)
                many_out["res"] = s.search_many(fq[a:a + m], qq[a:a + m], 10, lex[a:a + m], chunk=int(rng.choice([0, 64, 200])))
            except Exception as e:   # noqa: BLE001
                errs.append(e)

        threads = [threading.Thread(target=caller, args=(t,)) for t in range(nthreads)] + [threading.Thread(target=many_caller)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errs:
            print("ERROR", errs[:2], flush=True)
            bad_initial += 1
            continue
        same = 0
        for i in range(NQ):
            if got[i][0] != want[i][0]:
                bad_initial += 1
                print(f"INITIAL MISMATCH pool {pool} round {rounds} query {i} threads {nthreads}", flush=True)
            same += int([h.doc_id for h in got[i][1]] == [h.doc_id for h in want[i][1]])
        a, b = many_out["range"]
        ini, fin = many_out["res"][0], many_out["res"][1]
        for j, i in enumerate(range(a, b)):
            if ini[j] != want[i][0]:
                bad_initial += 1
                print(f"INITIAL MISMATCH (search_many) pool {pool} round {rounds} query {i}", flush=True)
        if same < 0.97 * NQ:
            low_final += 1
            print(f"refined lists: only {same} of {NQ} identical (pool {pool}, round {rounds}, {nthreads} threads)", flush=True)
    s.set_batching(0)
    s.close()
print(f"stress_engine: {rounds} rounds, {bad_initial} initial-list mismatches / errors, {low_final} rounds below 97 % identical refined lists")
sys.exit(1 if bad_initial or low_final else 0)
