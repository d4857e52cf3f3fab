"""The lone query's MiniLM forward: ONE launch with grid-wide barriers (default) against the replayed graph of 25 launches
(FSGPU_BERT_NO_ONE_LAUNCH=1) — latency through the C ABI; with SAVE=path the outputs are stored for comparison between the two runs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
import frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights

rng = np.random.default_rng(0)
bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536))
outs = []
for ntok in (3, 8, 12, 20, 32):
    ids = np.array([101] + rng.integers(1000, 30000, ntok - 2).tolist() + [102], np.int32)
    offs = np.array([0, ntok], np.uint32)
    out = np.empty((1, 384), np.float32)
    for _ in range(5): bert.embed_flat(ids, offs, out)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); bert.embed_flat(ids, offs, out); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{ntok:2d} tokens: p50 {1e3 * ts[150]:.4f} ms  p90 {1e3 * ts[270]:.4f} ms  min {1e3 * ts[0]:.4f} ms", flush=True)
    outs.append(out.copy())
# three short texts in one call (<= 32 tokens in all)
ids = np.array([101, 2000, 2001, 102, 101, 3000, 102, 101, 4000, 4001, 4002, 102], np.int32)
offs = np.array([0, 4, 7, 12], np.uint32)
out = np.empty((3, 384), np.float32)
bert.embed_flat(ids, offs, out)
outs.append(out.copy())
if os.environ.get("SAVE"):
    np.save(os.environ["SAVE"], np.concatenate(outs))
