"""The same batches again and again: every repetition of a batched search must return the first repetition's rows and score bits
(and the exact kernels').  A race in the selection / append / merge paths shows up here as a run that differs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import frankensearch_amd as fa
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
only_i8 = len(sys.argv) > 3 and sys.argv[3] == "i8"   # the int8 filter alone (the path every differing repetition so far was on)
rng = np.random.default_rng(seed)
bad = 0
for case in range(6):
    dim = int(rng.choice([128, 256, 384]))
    n = int(rng.integers(60_000, 300_000))
    kind = case % 3
    if kind == 0:
        cent = rng.standard_normal((64, dim)).astype(np.float32)
        x = cent[rng.integers(0, 64, n)] + (rng.standard_normal((n, dim)) * 0.05).astype(np.float32)
        dup = rng.integers(0, n, n // 20)
        x[dup] = x[(dup * 7 + 1) % n]
    elif kind == 1:
        cent = rng.standard_normal((n // 500 + 1, dim)).astype(np.float32)
        x = cent[np.arange(n) // 500] + (rng.standard_normal((n, dim)) * 0.1).astype(np.float32)
    else:
        x = rng.standard_normal((n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-9
    slab = x.astype(np.float16).view(np.uint16)
    live = rng.random(n) > 0.2 if case & 1 else None
    allow = rng.random(n) > 0.3 if case & 2 else None
    nq = int(rng.choice([520, 640, 1030]))
    k = int(rng.choice([10, 30, 33]))
    q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.2).astype(np.float32)
    idx = fa.VectorIndex.from_slab(slab, live=live)
    exact = [idx.search_batch(q[s0:s0 + 64], k, allow=allow) for s0 in range(0, nq, 64)]
    er = np.concatenate([e[0] for e in exact]); es = np.concatenate([e[1] for e in exact])
    for filt in ((2,) if only_i8 else (2, 1)):
        idx.set_batched_filter(filt)
        for r in range(reps):
            br, bs, bc, f = idx.search_batched(q, k, allow=allow)
            if not (np.array_equal(br, er) and np.array_equal(bs.view(np.uint32), es.view(np.uint32))):
                bad += 1
                w = np.nonzero(np.any(br != er, axis=1) | np.any(bs.view(np.uint32) != es.view(np.uint32), axis=1))[0]
                qi = int(w[0])
                pos = np.nonzero((br[qi] != er[qi]) | (bs[qi].view(np.uint32) != es[qi].view(np.uint32)))[0]
                print(f"case {case} (dim {dim} n {n} kind {kind} nq {nq} k {k} live {live is not None} allow {allow is not None}) waves {sorted(set((w % 512 // 64).tolist()))}", flush=True)
                print(f"case {case} filter {filt} rep {r}: {w.size} queries differ, first {w[:4]}; query {qi} count {bc[qi]} ranks {pos[:8]} got rows {br[qi][pos[:4]]} "
                      f"scores {bs[qi][pos[:4]]} want {er[qi][pos[:4]]} {es[qi][pos[:4]]} fallbacks {f}", flush=True)
    # the exact kernels themselves, repeated
    for r in range(0 if only_i8 else reps // 3):
        ex2 = [idx.search_batch(q[s0:s0 + 64], k, allow=allow) for s0 in range(0, nq, 64)]
        r2 = np.concatenate([e[0] for e in ex2]); s2 = np.concatenate([e[1] for e in ex2])
        if not (np.array_equal(r2, er) and np.array_equal(s2.view(np.uint32), es.view(np.uint32))):
            bad += 1
            print(f"case {case} EXACT kernels rep {r} differ from their first run", flush=True)
    print(f"case {case}: dim {dim} n {n} kind {kind} nq {nq} k {k} live {live is not None} allow {allow is not None}: done", flush=True)
    idx.close()
print(f"seed={seed}: {bad} differing repetitions")
