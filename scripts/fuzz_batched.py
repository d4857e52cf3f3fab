#!/usr/bin/env python3
"""Randomised cross-check of the batched matrix-core paths against the per-query kernels (run on the GPU box).

Every case draws a corpus shape, a data kind (gaussian / clustered with exact duplicates / few distinct rows / topical runs),
tombstones, an optional allow bitmap, a batch size that exercises multi-group rounds with ragged tails, and k; the batched
answer under BOTH filters (int8 slab, f16 slab) must equal the exact kernels' bit for bit for every query, and the batched
int8 two-pass must equal the per-query int8 two-pass.  Some corpora get outlier dimensions or a non-finite value.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frankensearch_amd as fa

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = bad = 0
tot_q = tot_r = 0
while time.time() < t_end:
    dim = int(rng.choice([128, 256, 384]))
    n = int(rng.integers(33_000, 300_000))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        x = rng.standard_normal((n, dim)).astype(np.float32)
    elif kind == 1:   # clusters with exact duplicates: ties and crowded thresholds
        cent = rng.standard_normal((64, dim)).astype(np.float32)
        x = cent[rng.integers(0, 64, n)] + (rng.standard_normal((n, dim)) * 0.05).astype(np.float32)
        dup = rng.integers(0, n, n // 20)
        x[dup] = x[(dup * 7 + 1) % n]
    elif kind == 2:   # very few distinct rows: pool overflow -> exact fallback
        base = rng.standard_normal((int(rng.integers(3, 40)), dim)).astype(np.float32)
        x = base[rng.integers(0, base.shape[0], n)]
    else:             # topical runs: neighbouring rows alike
        cent = rng.standard_normal((n // 500 + 1, dim)).astype(np.float32)
        x = cent[np.arange(n) // 500] + (rng.standard_normal((n, dim)) * 0.1).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-9
    slab = x.astype(np.float16).view(np.uint16)
    live = None if rng.random() < 0.5 else (rng.random(n) > 0.2)
    allow = None if rng.random() < 0.6 else (rng.random(n) > float(rng.choice([0.3, 0.9])))
    nq = int(rng.choice([1, 5, 63, 64, 65, 127, 128, 129, 200, 256, 257, 300, 511, 520, 640, 700, 1030]))
    k = int(rng.choice([1, 2, 10, 30, 33, 64]))
    q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.2).astype(np.float32)
    if nq > 3:
        q[1] = 0.0                      # zero query
        q[2] *= 37.5                    # non-unit query
        if rng.random() < 0.3:
            q[3, int(rng.integers(0, dim))] = rng.choice([np.nan, np.inf])  # non-finite: answered by the exact kernels
        if nq > 6:
            q[4, int(rng.integers(0, dim))] = float(rng.choice([65520.0, -7e4, 3e5]))   # finite, but +-inf as f16
            q[5] = (rng.standard_normal(dim) * float(rng.choice([3e-6, 1e-9]))).astype(np.float32)  # f16 subnormals / zeros
            q[6, :: int(rng.integers(2, 9))] = 1.5e-6
    outl = rng.random()
    if outl < 0.25:    # outlier dimensions (as real embedding models have): they stretch the corpus-wide int8 scale
        cols = rng.integers(0, dim, int(rng.integers(1, 4)))
        x[:, cols] *= float(rng.choice([4.0, 12.0]))
        x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-9
        slab = x.astype(np.float16).view(np.uint16)
    elif outl < 0.30:  # a non-finite value in the slab: no int8 bound exists, the f16 filter / exact kernels answer
        slab = slab.copy()
        slab[int(rng.integers(0, n)), int(rng.integers(0, dim))] = int(rng.choice([0x7c00, 0xfc00, 0x7e00]))
    idx = fa.VectorIndex.from_slab(slab, live=live)
    ok = True
    exact = [idx.search_batch(q[s0:s0 + 64], k, allow=allow) for s0 in range(0, nq, 64)]
    fb = []
    for filt in (2, 1):   # int8 filter, f16 filter: both must give the exact kernels' rows and score bits
        idx.set_batched_filter(filt)
        br, bs, bc, f = idx.search_batched(q, k, allow=allow)
        fb.append(f)
        for j, s0 in enumerate(range(0, nq, 64)):
            er, es, ec = exact[j]
            sl = slice(s0, min(nq, s0 + 64))
            if not (np.array_equal(bc[sl], ec) and np.array_equal(br[sl], er) and
                    np.array_equal(bs[sl].view(np.uint32), es.view(np.uint32))):
                ok = False
                for qi in range(sl.start, sl.stop):
                    j0 = qi - sl.start
                    if not (bc[qi] == ec[j0] and np.array_equal(br[qi], er[j0]) and np.array_equal(bs[qi].view(np.uint32), es[j0].view(np.uint32))):
                        bad_at = np.nonzero(br[qi] != er[j0])[0]
                        print(f"  filter={filt} query={qi} counts {bc[qi]} vs {ec[j0]} first differing rank {bad_at[:3]} got {br[qi][bad_at[:3]]} "
                              f"{bs[qi][bad_at[:3]]} want {er[j0][bad_at[:3]]} {es[j0][bad_at[:3]]}", flush=True)
                        break
    st = idx.batched_filter_stats()
    tot_q += st["int8_queries"]
    tot_r += st["refiltered_f16"]
    if allow is None:
        mult = int(rng.choice([1, 3, 5]))
        r8, s8, c8, fb8 = idx.search_int8_two_pass_batched(q, k, mult)
        for qi in rng.choice(nq, min(nq, 6), replace=False):
            hits = idx.search_top_k_int8_two_pass(q[qi], k, mult)
            if [h.index for h in hits] != r8[qi, :c8[qi]].tolist() or \
               not np.array_equal(np.array([h.score for h in hits], np.float32).view(np.uint32), s8[qi, :c8[qi]].view(np.uint32)):
                ok = False
        if rng.random() < 0.5:
            mult4 = int(rng.choice([1, 5]))
            r4, s4, c4, fb4 = idx.search_4bit_two_pass_batched(q, k, mult4)
            for qi in rng.choice(nq, min(nq, 4), replace=False):
                hits = idx.search_top_k_4bit_two_pass(q[qi], k, mult4)
                if [h.index for h in hits] != r4[qi, :c4[qi]].tolist() or \
                   not np.array_equal(np.array([h.score for h in hits], np.float32).view(np.uint32), s4[qi, :c4[qi]].view(np.uint32)):
                    ok = False
    cases += 1
    if not ok:
        bad += 1
        if os.environ.get("FUZZ_DUMP_DIR"):
            np.savez_compressed(os.path.join(os.environ["FUZZ_DUMP_DIR"], f"fuzz_fail_{seed}_{cases}.npz"), slab=slab, q=q, k=k,
                                live=live if live is not None else np.zeros(0, bool), allow=allow if allow is not None else np.zeros(0, bool))
        print(f"MISMATCH seed={seed} case={cases} dim={dim} n={n} kind={kind} nq={nq} k={k} live={live is not None} allow={allow is not None} fb={fb}", flush=True)
    idx.close()
print(f"seed={seed}: {cases} cases, {bad} mismatches; int8 filter took {tot_q} queries, handed {tot_r} on to the f16 filter")
sys.exit(1 if bad else 0)
