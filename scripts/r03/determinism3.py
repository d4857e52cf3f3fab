"""Repro hunt for the rare differing repetition: clustered corpus with duplicates, dim 256, 1,030 queries, k = 33, int8 filter;
modes: no bitmap / tombstones / tombstones + allow bitmap; many repetitions each."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import frankensearch_amd as fa
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["none", "live", "both"]
rng = np.random.default_rng(24)
dim, n, nq, k = 256, 112_990, 1030, 33
cent = rng.standard_normal((64, dim)).astype(np.float32)
x = cent[rng.integers(0, 64, n)] + (rng.standard_normal((n, dim)) * 0.05).astype(np.float32)
dup = rng.integers(0, n, n // 20)
x[dup] = x[(dup * 7 + 1) % n]
x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-9
slab = x.astype(np.float16).view(np.uint16)
live_m = rng.random(n) > 0.2
allow_m = rng.random(n) > 0.3
q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.2).astype(np.float32)
for mode in modes:
    live = live_m if mode in ("live", "both") else None
    allow = allow_m if mode == "both" else None
    idx = fa.VectorIndex.from_slab(slab, live=live)
    exact = [idx.search_batch(q[s0:s0 + 64], k, allow=allow) for s0 in range(0, nq, 64)]
    er = np.concatenate([e[0] for e in exact]); es = np.concatenate([e[1] for e in exact])
    idx.set_batched_filter(2)
    bad = 0
    for r in range(reps):
        br, bs, bc, f = idx.search_batched(q, k, allow=allow)
        if not (np.array_equal(br, er) and np.array_equal(bs.view(np.uint32), es.view(np.uint32))):
            bad += 1
            w = np.nonzero(np.any(br != er, axis=1) | np.any(bs.view(np.uint32) != es.view(np.uint32), axis=1))[0]
            missing = [sorted(set(er[qi][er[qi] != 0xFFFFFFFF].tolist()) - set(br[qi].tolist())) for qi in w[:6]]
            print(f"mode {mode} rep {r}: queries {w[:8].tolist()} (waves {sorted(set((w % 512 // 64).tolist()))}) missing rows {missing} fb {f}", flush=True)
    print(f"mode {mode}: {bad} of {reps} repetitions differ", flush=True)
    idx.close()
