import sys, os, json, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights
dev = torch.device("cuda", 0)
rows = 10_000_000
slab = bench.gen_corpus(0, rows, 384, dev)
idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, 384, device=0, keepalive=slab)
bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536), device=0)
rng = np.random.default_rng(1)
batch = []
for i in range(512):
    n = int(rng.integers(6, 24))
    batch.append([101] + [int(x) for x in rng.integers(1000, 30000, n)] + [102])
q = bert.embed_batch_token_ids(batch)
print("query norms", np.linalg.norm(q, axis=1)[:4], "max|q| stats", np.abs(q).max(axis=1).mean(), np.abs(q).max(axis=1).max(), "rms", np.sqrt((q**2).mean()))
delta, qscale, sscale, qi8, _ = idx.int8_filter_bound(q)
unit = sscale * qscale
print("delta (score units) mean/max", (delta/unit).mean(), (delta/unit).max(), "slab scale", sscale)
cq = bench.gen_queries(64, 384, dev).cpu().numpy()
d2, qs2, ss2, _, _ = idx.int8_filter_bound(cq)
print("clustered queries: delta (score units) mean/max", (d2/(ss2*qs2)).mean(), (d2/(ss2*qs2)).max(), "max|q|", np.abs(cq).max(axis=1).mean())
for k in (10, 30):
    for nq in (128, 512):
        idx.set_batched_filter(2)
        r, s, c, fb = idx.search_batched(q[:nq], k)
        st = idx.batched_filter_stats()
        print("k", k, "nq", nq, "fallbacks", fb, st)
