"""Repro of the rare differing repetition: dim 384, gaussian rows, k = 30, tombstones, int8 filter, 520 queries — many repetitions."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import frankensearch_amd as fa
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
use_live = (sys.argv[2] != "nolive") if len(sys.argv) > 2 else True
k = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rng = np.random.default_rng(7)
dim, n, nq = 384, 118597, 520
x = rng.standard_normal((n, dim)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-9
slab = x.astype(np.float16).view(np.uint16)
live = rng.random(n) > 0.2 if use_live else None
q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.2).astype(np.float32)
idx = fa.VectorIndex.from_slab(slab, live=live)
exact = [idx.search_batch(q[s0:s0 + 64], k) for s0 in range(0, nq, 64)]
er = np.concatenate([e[0] for e in exact]); es = np.concatenate([e[1] for e in exact])
idx.set_batched_filter(2)
bad = 0
for r in range(reps):
    br, bs, bc, f = idx.search_batched(q, k)
    if not (np.array_equal(br, er) and np.array_equal(bs.view(np.uint32), es.view(np.uint32))):
        bad += 1
        w = np.nonzero(np.any(br != er, axis=1) | np.any(bs.view(np.uint32) != es.view(np.uint32), axis=1))[0]
        qi = int(w[0])
        pos = np.nonzero((br[qi] != er[qi]) | (bs[qi].view(np.uint32) != es[qi].view(np.uint32)))[0]
        print(f"rep {r}: {w.size} queries differ {w[:5]}; query {qi} count {bc[qi]} ranks {pos[:6]} got rows {br[qi][pos[:4]]} scores {bs[qi][pos[:4]]} want {er[qi][pos[:4]]} {es[qi][pos[:4]]} fb {f}", flush=True)
print(f"live={use_live} k={k}: {bad} of {reps} repetitions differ")
