import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import frankensearch_amd as fa
S = fa.NativeShardedIndex
for seed in (int(x) for x in (sys.argv[1:] or ["10405", "9937", "10482"])):
    rng = np.random.default_rng(seed)
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
        x[:, rng.integers(0, dim, 3)] *= 12.0
    x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-9
    slab = x.astype(np.float16).view(np.uint16)
    groups, shards = [(1, 2), (2, 2), (3, 1), (2, 4), (1, 5), (4, 2), (1, 1), (2, 1)][int(rng.integers(0, 8))]
    live = (rng.random(n) > 0.15) if rng.random() < 0.4 else None
    whole = fa.VectorIndex.from_slab(slab, live=live)
    lat = bool(rng.integers(0, 2))
    nq = int(rng.choice([1, 2, 5, 9, 63, 130, 257, 300, 520]))
    k = int(rng.choice([1, 3, 10, 30, 33]))
    q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.15).astype(np.float32)
    print(f"seed {seed} dim {dim} n {n} kind {kind} layout {groups}x{shards} live {live is not None} lat {lat} nq {nq} k {k}")
    ref = [np.concatenate(z) for z in zip(*[whole.search_batch(q[s0:s0 + 64], k, exact=True) for s0 in range(0, nq, 64)])]
    for name, fn in (("unsharded search_batched", lambda: whole.search_batched(q, k)),):
        for rep in range(3):
            r, s, c, fb = fn()
            bad = [i for i in range(nq) if not (c[i] == ref[2][i] and np.array_equal(r[i], ref[0][i]) and np.array_equal(s[i].view(np.uint32), ref[1][i].view(np.uint32)))]
            print(f"  {name} rep {rep}: {len(bad)} bad queries {bad[:10]} fallbacks {fb} rotated {whole.filter_rotated()}")
            for i in bad[:2]:
                miss = sorted(set(ref[0][i].tolist()) - set(r[i].tolist()))
                extra = sorted(set(r[i].tolist()) - set(ref[0][i].tolist()))
                print(f"    q{i}: count {c[i]} vs {ref[2][i]} missing {miss[:5]} extra {extra[:5]} ref scores tail {ref[1][i][-3:]} got tail {s[i][-3:]}")
    idx = S.from_slab(slab, [0] * (groups * shards), live=live, exchange=S.EXCHANGE_PEER_COPY, query_groups=groups)
    idx.set_int8_latency(lat)
    r, s, c, fb = idx.search(q, k, S.BATCHED)
    bad = [i for i in range(nq) if not (c[i] == ref[2][i] and np.array_equal(r[i][:c[i]], ref[0][i][:c[i]]) and np.array_equal(s[i][:c[i]].view(np.uint32), ref[1][i][:c[i]].view(np.uint32)))]
    print(f"  sharded batched: {len(bad)} bad queries {bad[:10]} fallbacks {fb}")
    for i in bad[:3]:
        miss = sorted(set(ref[0][i].tolist()) - set(r[i].tolist()))
        extra = sorted(set(r[i].tolist()) - set(ref[0][i].tolist()))
        print(f"    q{i}: count {c[i]} vs {ref[2][i]} missing {miss[:5]} extra {extra[:5]}; pos of first diff {int(np.argmax(r[i] != ref[0][i]))}")
    idx.close(); whole.close()
