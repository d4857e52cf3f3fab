#!/usr/bin/env python3
"""The many-form's three GPU stages one after another, each alone on the GPU (for rocprofv3 --kernel-trace --stats): fast tier int8
two-pass batched (10M x 256), quality tier batched exact (10M x 384), MiniLM batch — 1,024 queries per call."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import frankensearch_amd as fa  # noqa: E402
from frankensearch_amd.synthetic import random_bert_weights  # noqa: E402

rows = int(os.environ.get("ROWS", 10_000_000))
B = int(os.environ.get("B", 1024))
reps = int(os.environ.get("REPS", 10))
dev = torch.device("cuda", 0)
qslab = bench.gen_corpus(0, rows, 384, dev)
fslab = bench.gen_corpus(0, rows, 256, dev)
quality = fa.VectorIndex.from_device_slab(qslab.data_ptr(), rows, 384, device=0, keepalive=qslab)
fast = fa.VectorIndex.from_device_slab(fslab.data_ptr(), rows, 256, device=0, keepalive=fslab)
rng = np.random.default_rng(0)
table = rng.standard_normal((500_353, 256)).astype(np.float32)
m2v = fa.Model2VecEmbedder(table, device=0)
bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536), device=0)
fq = [rng.integers(0, 500_353, int(rng.integers(4, 24))).astype(np.uint32) for _ in range(B)]
foffs = np.zeros(B + 1, np.uint32); foffs[1:] = np.cumsum([len(x) for x in fq])
fvec = np.empty((B, 256), np.float32)
m2v.embed_flat(np.concatenate(fq), foffs, fvec)
texts = [np.concatenate([[101], rng.integers(1000, 30000, int(rng.integers(6, 31))), [102]]).astype(np.int32) for _ in range(B)]
offs = np.zeros(B + 1, np.uint32); offs[1:] = np.cumsum([len(t) for t in texts])
ids = np.concatenate(texts)
qvec = np.empty((B, 384), np.float32)
bert.embed_flat(ids, offs, qvec)


def timed(name, fn):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    print(f"{name}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per {B} queries", flush=True)


STAGE = os.environ.get("STAGE")
if STAGE == "fast":
    timed("fast tier int8 two-pass batched (fetch 30, x3)", lambda: fast.search_int8_two_pass_batched(fvec, 30, 3))
    sys.exit(0)
if STAGE == "quality":
    timed("quality tier batched exact (fetch 30)", lambda: quality.search_batched(qvec, 30))
    sys.exit(0)
if STAGE == "quality10":
    timed("quality tier batched exact (k 10)", lambda: quality.search_batched(qvec, 10))
    sys.exit(0)
timed("fast tier int8 two-pass batched (fetch 30, x3)", lambda: fast.search_int8_two_pass_batched(fvec, 30, 3))
timed("fast tier batched exact (fetch 30)", lambda: fast.search_batched(fvec, 30))
timed("quality tier batched exact (fetch 30)", lambda: quality.search_batched(qvec, 30))
timed("quality tier batched exact (k 10)", lambda: quality.search_batched(qvec, 10))
timed("MiniLM batch", lambda: bert.embed_flat(ids, offs, qvec))
timed("Model2Vec batch", lambda: m2v.embed_flat(np.concatenate(fq), foffs, fvec))
