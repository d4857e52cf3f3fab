#!/usr/bin/env python3
"""Closed-loop per-query callers (fshost_run_load) with the dynamic batching of fshost_two_tier_set_batching: threads x max_chunk x wait."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import frankensearch_amd as fa  # noqa: E402
from frankensearch_amd.host import NativeTwoTierSearcher  # noqa: E402
from frankensearch_amd.synthetic import random_bert_weights  # noqa: E402

rows = int(os.environ.get("ROWS", 10_000_000))
dev = torch.device("cuda", 0)
qslab = bench.gen_corpus(0, rows, 384, dev)
fslab = bench.gen_corpus(0, rows, 256, dev)
quality = fa.VectorIndex.from_device_slab(qslab.data_ptr(), rows, 384, device=0, keepalive=qslab)
fast = fa.VectorIndex.from_device_slab(fslab.data_ptr(), rows, 256, device=0, keepalive=fslab)
table = np.random.default_rng(0).standard_normal((500_353, 256)).astype(np.float32)
m2v = fa.Model2VecEmbedder(table, device=0)
bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536), device=0)
s = NativeTwoTierSearcher(fast, quality, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3)
s.run_load_many(queries=8192, warmup_queries=2048, k=10, fast_vocab=500_353, corpus_rows=rows)   # builds the int8 copies, warms the engine
# default: the setting bench.py ships (max_chunk 256, max_wait 3,000 us) over the caller counts, then two contrasts (a short wait starves
# the chunks of a small population: 64 callers in chunks of 16; a larger cap for 2,048 callers)
cases = os.environ.get("CASES", "1:256:3000,8:256:3000,64:256:3000,128:256:3000,200:256:3000,256:256:3000,512:256:3000,1024:256:3000,2048:256:3000,"
                                "64:256:200,2048:1024:3000")
for case in cases.split(","):
    threads, chunk, wait = (int(x) for x in case.split(":"))
    s.set_batching(chunk, wait)
    c0, r0 = s.batching_stats()
    nq = 300 if threads == 1 else (6000 if threads < 100 else 60_000)
    r = s.run_load(threads=threads, queries=nq, warmup_queries=max(threads * 2, 64), k=10, fast_vocab=500_353, corpus_rows=rows)
    c1, r1 = s.batching_stats()
    print(f"threads={threads:5d} max_chunk={chunk:5d} wait={wait:4d}us  qps={r.queries_per_sec:9.1f}  p0 p50={r.phase0_p50_ms:7.3f} p1 p50={r.phase1_p50_ms:7.3f} "
          f"p95={r.phase1_p95_ms:7.3f} p99={r.phase1_p99_ms:7.3f}  mean chunk {(r1 - r0) / max(c1 - c0, 1):7.1f}  failed={r.failed} {r.first_error}", flush=True)
s.set_batching(0, 0)
