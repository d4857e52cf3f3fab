#!/usr/bin/env python3
"""Index-build throughput of the MiniLM encoder with ONE and with TWO encoder handles on the same GPU (two host threads, each with its
own handle and stream, 32 documents x 512 tokens per call): kernels of one forward that leave the chip partly idle — the attention is
bound by the vector ALUs, the QKV projection by latency, the post-attention block by the L2s' weight stream — run beside the other's."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights

w = random_bert_weights(1, 30522, 384, 6, 1536)
rng = np.random.default_rng(0)
long = [[101] + rng.integers(1000, 30000, 510).tolist() + [102] for _ in range(32)]
offs = np.zeros(33, dtype=np.uint32)
offs[1:] = np.cumsum([len(b) for b in long])
ids = np.concatenate([np.asarray(b, dtype=np.int32) for b in long])
N = int(os.environ.get("N", "300"))


def worker(enc, out, n):
    for _ in range(n):
        enc.embed_flat(ids, offs, out)


for handles in (1, 2, 3):
    encs = [fa.NativeEmbedder(w) for _ in range(handles)]
    outs = [np.empty((32, 384), dtype=np.float32) for _ in range(handles)]
    for e, o in zip(encs, outs):
        worker(e, o, 30)
    th = [threading.Thread(target=worker, args=(e, o, N)) for e, o in zip(encs, outs)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    same = all(np.array_equal(outs[0].view(np.uint32), o.view(np.uint32)) for o in outs)
    print(f"{handles} handle(s): {handles * N * 32 / dt:.0f} documents/s ({handles * N * 32 * 512 / dt / 1e6:.2f} M tokens/s), {dt / N * 1e3:.3f} ms per call per handle, outputs identical: {same}")
    for e in encs:
        e.close()
