#!/usr/bin/env python3
"""Thread-count / batching-window sweep of the native two-tier load generator on the bench corpus (config 3)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import frankensearch_amd as fa  # noqa: E402
from frankensearch_amd.host import NativeTwoTierSearcher  # noqa: E402
from frankensearch_amd.synthetic import random_bert_weights  # noqa: E402

rows = int(os.environ.get("ROWS", 10_000_000))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
qslab = bench.gen_corpus(0, rows, 384, dev)
fslab = bench.gen_corpus(0, rows, 256, dev)
quality = fa.VectorIndex.from_device_slab(qslab.data_ptr(), rows, 384, device=0, keepalive=qslab)
fast = fa.VectorIndex.from_device_slab(fslab.data_ptr(), rows, 256, device=0, keepalive=fslab)
rng = np.random.default_rng(0)
table = rng.standard_normal((500_353, 256)).astype(np.float32)
m2v = fa.Model2VecEmbedder(table, device=0)
bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536), device=0)
native = NativeTwoTierSearcher(fast, quality, m2v, bert, doc_id_mode=1,
                               fast_tier_int8_multiplier=int(os.environ.get("FAST_INT8", "3")))
SWEEP = os.environ.get("SWEEP")
cases = ((1, 0, 0), (1, 128, 1000), (8, 128, 1000), (64, 128, 1000), (256, 128, 1000), (1024, 128, 1000))
if SWEEP:  # "threads:batch:wait[:quality_batch],..."
    cases = tuple(tuple(int(x) for x in c.split(":")) for c in SWEEP.split(","))
for case in cases:
    threads, mb, wait = case[:3]
    qmb = case[3] if len(case) > 3 else mb
    fast.set_coalescing(mb, wait)
    quality.set_coalescing(qmb, wait)
    m2v.set_coalescing(2 * mb, wait // 2)
    bert.set_coalescing(2 * mb, wait)
    nq = 200 if threads == 1 else (4000 if threads < 100 else 40_000)
    r = native.run_load(threads=threads, queries=nq, warmup_queries=max(threads * 2, 64), k=10, fast_vocab=500_353,
                        corpus_rows=rows)
    print(f"threads={threads:5d} batch={mb:4d}/{qmb:4d} wait={wait:5d}us  qps={r.queries_per_sec:9.1f}  p0 p50={r.phase0_p50_ms:7.3f} "
          f"p1 p50={r.phase1_p50_ms:7.3f} p95={r.phase1_p95_ms:7.3f}  means: fe={r.mean_fast_embed_ms:.3f} fs={r.mean_fast_search_ms:.3f} "
          f"qe={r.mean_quality_embed_ms:.3f} qs={r.mean_quality_search_ms:.3f} fuse={r.mean_fusion_ms:.3f} failed={r.failed} {r.first_error}",
          flush=True)
    print("   coalescing (batches, requests): fast", fast.coalescing_stats(), "quality", quality.coalescing_stats(), flush=True)
