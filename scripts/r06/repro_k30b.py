import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import frankensearch_amd as fa
from frankensearch_amd.sharded import GpuShardBackend
rng = np.random.default_rng(10405)
dim, n, nq = 384, 42981, 300
x = rng.standard_normal((n, dim)).astype(np.float32)
x[:, rng.integers(0, dim, 3)] *= 12.0
x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-9
slab = x.astype(np.float16).view(np.uint16)
q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.15).astype(np.float32)
dev = torch.device("cuda", 0)
qd = torch.from_numpy(q).to(dev)
for k in (10, 24, 25, 30, 32, 33):
    whole = fa.VectorIndex.from_slab(slab)
    ref = [np.concatenate(z) for z in zip(*[whole.search_batch(q[s0:s0 + 64], k, exact=True) for s0 in range(0, nq, 64)])]
    be = GpuShardBackend(whole, dev, batched=True)
    for packed in (False, True):
        out, t = be.scan_begin(qd, k, packed=packed)
        fb = be.scan_end(t)
        torch.cuda.synchronize()
        if packed:
            p = out.cpu().numpy().view(np.uint64)
            r = (p & np.uint64(0xFFFFFFFF)).astype(np.uint32); s = (p >> np.uint64(32)).astype(np.uint32)
        else:
            r = out[0].cpu().numpy().view(np.uint32); s = out[1].cpu().numpy().view(np.uint32)
        bad = [i for i in range(nq) if not (np.array_equal(r[i], ref[0][i].view(np.uint32)) and np.array_equal(s[i], ref[1][i].view(np.uint32)))]
        print(f"k {k} begin/end packed={packed}: {len(bad)} bad {bad[:8]} fb {fb}")
    r, s, c, fb = whole.search_batched(q, k)
    bad = [i for i in range(nq) if not (np.array_equal(r[i], ref[0][i]) and np.array_equal(s[i].view(np.uint32), ref[1][i].view(np.uint32)))]
    print(f"k {k} blocking host call: {len(bad)} bad")
    whole.close()
