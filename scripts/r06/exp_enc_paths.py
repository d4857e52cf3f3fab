#!/usr/bin/env python3
"""One-launch forward of short texts (bert_docs_w) against the three-launches-per-layer batch path, by batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights
rng = np.random.default_rng(0)
w = random_bert_weights(1, 30522, 384, 6, 1536)
bert = fa.NativeEmbedder(w)
for B in (int(x) for x in os.environ.get("BS", "128,256,384,512,768,1024,2048").split(",")):
    texts = [np.concatenate([[101], rng.integers(1000, 30000, int(rng.integers(6, 31))), [102]]).astype(np.int32) for _ in range(B)]
    offs = np.zeros(B + 1, np.uint32); offs[1:] = np.cumsum([len(t) for t in texts])
    ids = np.concatenate(texts)
    out = np.empty((B, 384), np.float32)
    for _ in range(8): bert.embed_flat(ids, offs, out)
    lat = []
    for _ in range(40):
        t0 = time.perf_counter(); bert.embed_flat(ids, offs, out); lat.append((time.perf_counter() - t0) * 1e3)
    print(f"batch {B:5d} ({ids.size:6d} tokens): median {np.median(lat):.3f} ms  min {np.min(lat):.3f}", flush=True)
