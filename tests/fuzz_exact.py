#!/usr/bin/env python3
"""Randomised parity of the exact search entry points against the CPU oracle (run on the GPU box; the oracle is the
checker here, exactly as in tests/)."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frankensearch_amd as fa
from oracle import oracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = bad = 0
tmp = tempfile.mkdtemp()
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
while time.time() < t_end:
    n = int(rng.choice([1, 2, 15, 16, 17, 100, 1000, 5000, 20000]))
    dim = int(rng.choice([1, 3, 7, 8, 12, 32, 40, 100, 128, 256, 384, 390]))
    f32 = rng.random() < 0.3
    x = rng.standard_normal((n, dim)).astype(np.float32)
    if n > 20 and rng.random() < 0.4:
        x[rng.integers(0, n, n // 4)] = x[0]          # ties
    if n > 5 and rng.random() < 0.2:
        x[rng.integers(0, n)] *= np.float32(1e4)       # an outlier
    nq = int(rng.integers(1, 10))
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    if rng.random() < 0.1:   # non-finite query elements: search_top_k does not reject them (only _classified does)
        q[0, int(rng.integers(0, dim))] = rng.choice([np.nan, np.inf, -np.inf])
    if rng.random() < 0.05:
        q[-1] = 0.0
    k = int(rng.choice([1, 2, 10, 64, 65, 256, 257, n, n + 3]))
    live = None if rng.random() < 0.5 else (rng.random(n) > 0.3)
    allow = None if rng.random() < 0.5 else (rng.random(n) > float(rng.choice([0.5, 0.99])))
    eff = None
    if live is not None or allow is not None:
        eff = np.ones(n, bool)
        if live is not None: eff &= live
        if allow is not None: eff &= allow
    ok = True
    if n >= 15 and dim >= 3 and rng.random() < 0.25:
        # a file index with duplicate doc ids, soft deletes and resident WAL entries (superseding and new ids):
        # search_top_k + resolve_hits (dedup, shadowing, tombstones; search.rs:1449-1596) against the oracle's
        if not np.all(np.linalg.norm(x, axis=1) > 1e-3):
            continue
        p = os.path.join(tmp, "w.fsvi")
        ids = [f"d{i % max(1, n - n // 10):06}" for i in range(n)]
        fa.write_fsvi(p, [(ids[i], x[i].tolist()) for i in range(n)], "e", "r", quantization=0 if f32 else 1)
        o = oracle.Fsvi(p)
        g = fa.VectorIndex.open(p)
        for _ in range(int(rng.integers(0, 6))):
            did = ids[int(rng.integers(0, n))]
            assert o.soft_delete(did) == g.soft_delete(did)
        for j in range(int(rng.integers(0, 8))):
            did = ids[int(rng.integers(0, n))] if j % 2 else f"new-{j}"
            v = rng.standard_normal(dim).astype(np.float32)
            assert o.append(did, v) == 0
            g.append(did, v)
        for qi in range(min(nq, 3)):
            kk = int(min(k, 300))
            oh, os_ = o.search_top_k(q[qi], kk)
            gh = g.search_top_k(q[qi], kk)
            if [(h.index, h.doc_id) for h in gh] != [(h[0], h[2]) for h in oh] or \
               not np.array_equal(bits([h.score for h in gh]), bits(os_)):
                ok = False
        cases += 1
        if not ok:
            bad += 1
            print(f"MISMATCH(file) seed={seed} case={cases} n={n} dim={dim} f32={f32} k={k}", flush=True)
        g.close()
        o.close()
        continue
    if f32:
        if live is not None:
            continue  # FSVI files start with every row live
        p = os.path.join(tmp, "f.fsvi")
        if not np.all(np.linalg.norm(x, axis=1) > 1e-3):
            continue
        fa.write_fsvi(p, [(f"d{i:06}", x[i].tolist()) for i in range(n)], "e", "r", quantization=0)
        o = oracle.Fsvi(p)
        slab = np.frombuffer(open(p, "rb").read()[o.vectors_offset:], dtype="<f4").reshape(n, dim)
        idx = fa.VectorIndex.open(p)
        rows, scores, counts = idx.search_batch(q, k, allow=allow)
        for qi in range(nq):
            er, es = oracle.search_top_k_f32(slab, q[qi], k, live=eff)
            m = int(counts[qi])
            if m != len(er) or not np.array_equal(rows[qi, :m], er) or not np.array_equal(bits(scores[qi, :m]), bits(es)):
                ok = False
    else:
        slab = x.astype(np.float16).view(np.uint16)
        idx = fa.VectorIndex.from_slab(slab, live=live)
        if allow is None and rng.random() < 0.35:
            # the quantised two-pass searches and the MRL search of the same index against the oracle's
            kk = int(min(k, 200))
            mult = int(rng.choice([1, 3, 5]))
            sd = int(rng.integers(1, dim + 2))
            rd = int(rng.choice([0, sd, dim]))
            rk = int(rng.choice([0, kk, 3 * kk]))
            for qi in range(min(nq, 2)):
                checks = [(idx.search_top_k_int8_two_pass(q[qi], kk, mult), oracle.search_int8_two_pass(slab, q[qi], kk, mult, live=live)),
                          (idx.search_top_k_4bit_two_pass(q[qi], kk, mult), oracle.search_4bit_two_pass(slab, q[qi], kk, mult, live=live)),
                          (idx.mrl_search(q[qi], kk, search_dims=sd, rescore_dims=rd, rescore_top_k=rk),
                           oracle.mrl_search(slab, q[qi], kk, sd, rd, rk, live=live))]
                for name, (hits, (er, es)) in zip(("int8", "4bit", "mrl"), checks):
                    if [h.index for h in hits] != er.tolist() or not np.array_equal(bits([h.score for h in hits]), bits(es)):
                        ok = False
                        print(f"  {name} differs: n={n} dim={dim} k={kk} mult={mult} sd={sd} rd={rd} rk={rk}", flush=True)
        rows, scores, counts = idx.search_batch(q, k, allow=allow)
        for qi in range(nq):
            er, es = oracle.search_top_k(slab, q[qi], k, live=eff)
            m = int(counts[qi])
            if m != len(er) or not np.array_equal(rows[qi, :m], er) or not np.array_equal(bits(scores[qi, :m]), bits(es)):
                ok = False
    cases += 1
    if not ok:
        bad += 1
        print(f"MISMATCH seed={seed} case={cases} n={n} dim={dim} f32={f32} nq={nq} k={k} live={live is not None} allow={allow is not None}", flush=True)
    idx.close()
print(f"seed={seed}: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
