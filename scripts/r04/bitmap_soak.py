"""Time-budgeted soak of FILTERED batched searches (tombstones and / or an allow bitmap, int8 filter): every repetition must return the
exact kernels' rows and score bits.  The r03 failure (one repetition in ~30,000 with a wrong bitmap word in the wide kernel's append
path) showed up here; this form counts repetitions so that library variants can be compared at equal effort
(scripts/r04/soak_ab.sh).   python scripts/r04/bitmap_soak.py SECONDS [first_seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import frankensearch_amd as fa

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t_end = time.time() + budget
reps_total = bad = cases = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    seed += 1
    for case in (1, 2, 3):
        if time.time() >= t_end:
            break
        dim = int(rng.choice([128, 256, 384]))
        n = int(rng.integers(60_000, 300_000))
        if case == 3:
            x = rng.standard_normal((n, dim)).astype(np.float32)
        else:
            cent = rng.standard_normal((64, dim)).astype(np.float32)
            x = cent[rng.integers(0, 64, n)] + (rng.standard_normal((n, dim)) * 0.05).astype(np.float32)
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
        idx.set_batched_filter(2)
        cases += 1
        for r in range(400):
            if time.time() >= t_end:
                break
            br, bs, bc, f = idx.search_batched(q, k, allow=allow)
            reps_total += 1
            if not (np.array_equal(br, er) and np.array_equal(bs.view(np.uint32), es.view(np.uint32))):
                bad += 1
                w = np.nonzero(np.any(br != er, axis=1) | np.any(bs.view(np.uint32) != es.view(np.uint32), axis=1))[0]
                extra = [sorted(set(br[qi].tolist()) - set(er[qi].tolist())) for qi in w[:4]]
                missing = [sorted(set(er[qi].tolist()) - set(br[qi].tolist())) for qi in w[:4]]
                print(f"seed {seed - 1} case {case} (dim {dim} n {n} nq {nq} k {k} live {live is not None} allow {allow is not None}) rep {r}: "
                      f"queries {w[:8].tolist()} waves {sorted(set((w % 512 // 64).tolist()))} extra rows {extra} missing rows {missing}", flush=True)
        idx.close()
# a lab build with -DFSGPU_LAB_BITMAP_CHECK records every bitmap word whose scalar-path value differed from its vector-path value
import ctypes
L = fa._lib.lib()
if hasattr(L, "fsgpu_lab_bitmap_debug"):
    buf = (ctypes.c_uint64 * (64 * 8))()
    n = L.fsgpu_lab_bitmap_debug(buf, 64)
    print(f"bitmap words whose scalar load disagreed with the vector load: {n}")
    for i in range(max(0, min(n, 64))):
        r = buf[i * 8:(i + 1) * 8]
        print(f"  block {r[0]:3d} wave {r[1]} word {r[2]:6d} bitmap {'allow' if r[3] & 0xff else 'live '} addr 0x{r[3] >> 8:x}: scalar 0x{r[4]:016x} "
              f"vector 0x{r[5]:016x} | again: scalar 0x{r[6]:016x} vector 0x{r[7]:016x} | differing bits {bin(r[4] ^ r[5]).count('1')}")
print(f"bitmap_soak: {reps_total} repetitions over {cases} indexes in {budget:.0f} s, {bad} differing", flush=True)
