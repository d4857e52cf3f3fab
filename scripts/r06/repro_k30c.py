import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import frankensearch_amd as fa
from frankensearch_amd.sharded import GpuShardBackend
S = fa.NativeShardedIndex
rng = np.random.default_rng(10405)
dim, n, nq = 384, 42981, 300
x = rng.standard_normal((n, dim)).astype(np.float32)
x[:, rng.integers(0, dim, 3)] *= 12.0
x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-9
slab = x.astype(np.float16).view(np.uint16)
q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.15).astype(np.float32)
dev = torch.device("cuda", 0)
qd = torch.from_numpy(q).to(dev)
w0 = fa.VectorIndex.from_slab(slab)
refs = {k: [np.concatenate(z) for z in zip(*[w0.search_batch(q[s0:s0 + 64], k, exact=True) for s0 in range(0, nq, 64)])] for k in (10, 30, 33)}
def nbad(r, s, k):
    ref = refs[k]
    return sum(1 for i in range(nq) if not (np.array_equal(np.asarray(r[i]).view(np.uint32), ref[0][i].view(np.uint32)) and np.array_equal(np.asarray(s[i]).view(np.uint32), ref[1][i].view(np.uint32))))
# A: fresh unsharded index, first call = begin/end packed k=30
w = fa.VectorIndex.from_slab(slab)
be = GpuShardBackend(w, dev, batched=True)
out, t = be.scan_begin(qd, 30, packed=True); be.scan_end(t); torch.cuda.synchronize()
p = out.cpu().numpy().view(np.uint64)
print("A fresh unsharded, first call begin/end packed k=30:", nbad((p & np.uint64(0xFFFFFFFF)).astype(np.uint32), (p >> np.uint64(32)).astype(np.uint32), 30))
w.close()
# B: sharded 1x1, sequences
for seq in ([30], [10, 30], [33, 30], [30, 30, 10]):
    idx = S.from_slab(slab, [0], exchange=S.EXCHANGE_PEER_COPY)
    res = []
    for k in seq:
        r, s, c, fb = idx.search(q, k, S.BATCHED)
        res.append((k, nbad(r, s, k)))
    print("B sharded 1x1 sequence", seq, "->", res)
    idx.close()
# C: sharded 1x1 with host queries of nq=256 only / 44 only
for sub in (slice(0, 256), slice(256, 300), slice(0, 128)):
    idx = S.from_slab(slab, [0], exchange=S.EXCHANGE_PEER_COPY)
    r, s, c, fb = idx.search(q[sub], 30, S.BATCHED)
    ref = refs[30]
    bad = sum(1 for j, i in enumerate(range(*sub.indices(nq))) if not (np.array_equal(r[j], ref[0][i]) and np.array_equal(s[j].view(np.uint32), ref[1][i].view(np.uint32))))
    print("C sharded 1x1 queries", sub, "bad", bad)
    idx.close()
