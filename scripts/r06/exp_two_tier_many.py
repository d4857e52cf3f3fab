#!/usr/bin/env python3
"""fshost_two_tier_search_many on the config-3 corpora (10M x 256 + 10M x 384): chunk size / fusion threads / pool sweep."""
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
torch.cuda.set_device(0)
qslab = bench.gen_corpus(0, rows, 384, dev)
fslab = bench.gen_corpus(0, rows, 256, dev)
quality = fa.VectorIndex.from_device_slab(qslab.data_ptr(), rows, 384, device=0, keepalive=qslab)
fast = fa.VectorIndex.from_device_slab(fslab.data_ptr(), rows, 256, device=0, keepalive=fslab)
table = np.random.default_rng(0).standard_normal((500_353, 256)).astype(np.float32)
m2v = fa.Model2VecEmbedder(table, device=0)
bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536), device=0)
cases = os.environ.get("CASES", "0:1024:0,0:1024:4,0:1024:8,0:1024:12,0:512:0,0:2048:0,1:1024:0")
nq = int(os.environ.get("NQ", 32768))
for case in cases.split(","):
    pool, chunk, threads = (int(x) for x in case.split(":"))
    s = NativeTwoTierSearcher(fast, quality, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3, quality_pool=pool)
    for rep in range(2):
        r = s.run_load_many(queries=nq, warmup_queries=4096, k=10, fast_vocab=500_353, corpus_rows=rows, chunk=chunk, fusion_threads=threads)
    print(f"pool={pool} chunk={chunk:5d} fusion_threads={r['fusion_threads']:2d}  qps={r['queries_per_sec']:9.1f}  per chunk ms: fe={r['mean_fast_embed_ms']:.3f} "
          f"fs={r['mean_fast_search_ms']:.3f} qe={r['mean_quality_embed_ms']:.3f} qs={r['mean_quality_search_ms']:.3f} fusion_busy={r['fusion_busy_ms_per_chunk']:.2f} "
          f"first initial {r['first_chunk_initial_ms']:.2f} refined {r['first_chunk_refined_ms']:.2f}  full={r['queries_with_k_initial_and_refined_hits']}/{r['queries']} "
          f"fb={r['fast_fallbacks']}/{r['quality_fallbacks']} dev={r['device_resident_handoff']} {r['error_detail']}", flush=True)
    s.close()
