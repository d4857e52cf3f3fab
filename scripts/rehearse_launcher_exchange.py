#!/usr/bin/env python3
"""Single-GPU rehearsal of the per-rank path of `bench.py --gpus N` with a ONE-rank RCCL process group: shard-local batched
scan -> all_gather_into_tensor (RCCL) -> merge of the gathered [W, B, k] lists on a side stream underneath the next step's scan.
Every step's hits must equal the unsharded index's, bit for bit.  Run by tests/test_gpu_sharded.py in a process of its own."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frankensearch_amd as fa
from frankensearch_amd.sharded import GpuShardBackend, ShardedVectorIndex
from oracle import oracle

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
    sock.close()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
n, dim, k, nq = 300_000, 384, 10, 600
oracle.build()
slab = oracle.clustered_corpus_f16(0, n, dim)
q = np.stack([oracle.clustered_query(i, dim) for i in range(nq)])
whole = fa.VectorIndex.from_slab(slab)
want = [whole.search_batch(q[s:s + 64], k) for s in range(0, nq, 64)]
wr = np.concatenate([w[0] for w in want])
ws = np.concatenate([w[1] for w in want])
index = fa.VectorIndex.from_slab(slab)
sharded = ShardedVectorIndex(GpuShardBackend(index, device, batched=True), overlap=True, force_collective=True)
tq = torch.from_numpy(q).to(device)
# the bench's loop (ShardedVectorIndex.search_steps): the exchange of step i - 1 enqueued from inside the scan call of step i
outs = sharded.search_steps(lambda i: tq, 0, 4, k, keep_all=True)
# (a backend whose scan comes in two halves is pipelined through scan_begin / scan_end; the others through the after-enqueue hook)
assert len(outs) == 4 and (sharded.backend.hook_fired or getattr(sharded.backend, "supports_pipelined_scans", False)), \
    "neither the pipelined scans nor the after-enqueue hook ran"
# ... and the begin / end form by hand
pending = None
for step in range(3):
    local = sharded.search_begin(tq, k)
    if pending is not None:
        pending[3].synchronize()
        outs.append(pending[:3])
    pending = sharded.search_end(local, k)
pending[3].synchronize()
outs.append(pending[:3])
outs.append(sharded.search(tq, k))            # the blocking form goes through the collective too
for rows, scores, counts in outs:
    assert np.array_equal(rows.cpu().numpy().astype(np.uint32), wr), "rows differ"
    assert np.array_equal(bits(scores.cpu().numpy()), bits(ws)), "score bits differ"
    assert int(counts.min().item()) == k
# The fallback tail.  A corpus of a handful of distinct rows overflows every candidate pool: the batched call synchronises its
# stream, decides on the fallbacks and returns with the exact kernels + the scatter of their hits only ENQUEUED.  The exchange of
# step i - 1, enqueued from inside the scan call of step i, must wait for that tail (search_begin's event) — every step has its own
# queries, so a merge of a half-written list cannot pass for the right answer.
rng = np.random.default_rng(5)
n2, nq2, steps2 = 200_000, 300, 5
base = rng.standard_normal((7, dim)).astype(np.float32)
x = base[rng.integers(0, 7, n2)]
x /= np.linalg.norm(x, axis=1, keepdims=True)
slab2 = x.astype(np.float16).view(np.uint16)
whole2 = fa.VectorIndex.from_slab(slab2)
index2 = fa.VectorIndex.from_slab(slab2)
sharded2 = ShardedVectorIndex(GpuShardBackend(index2, device, batched=True), overlap=True, force_collective=True)
qs2 = [(x[rng.integers(0, n2, nq2)] + 0.2 * rng.standard_normal((nq2, dim))).astype(np.float32) for _ in range(steps2)]
tqs2 = [torch.from_numpy(a).to(device) for a in qs2]
fallbacks = []
outs2 = sharded2.search_steps(lambda i: tqs2[i], 0, steps2, k, keep_all=True,
                              after_scan=lambda: fallbacks.append(sharded2.backend.last_fallbacks))
assert len(outs2) == steps2 and min(fallbacks) > 0, fallbacks
for i, (rows, scores, counts) in enumerate(outs2):
    want2 = [whole2.search_batch(qs2[i][s0:s0 + 64], k) for s0 in range(0, nq2, 64)]
    assert np.array_equal(rows.cpu().numpy().astype(np.uint32), np.concatenate([w[0] for w in want2])), f"fallback step {i}: rows differ"
    assert np.array_equal(bits(scores.cpu().numpy()), bits(np.concatenate([w[1] for w in want2]))), f"fallback step {i}: score bits differ"
# The RE-FILTER tail (round 6).  Tight clusters: the int8 filter's lists overflow for most queries and the end half hands them to the f16
# filter, which certifies them — no exact fallback (last_fallbacks == 0), yet their hits are written by work enqueued in the end half.
# Through round 5 the exchange was re-ordered only behind FALLBACKS: a re-filtered query's corrected list never travelled
# (scripts/fuzz_sharded.py found it).  scan_end now reports late answers of both kinds.
rng = np.random.default_rng(6)
n3, nq3, steps3, k3 = 90_000, 300, 4, 10
cent = rng.standard_normal((4, 256)).astype(np.float32)
x3 = cent[rng.integers(0, 4, n3)] + (rng.standard_normal((n3, 256)) * 0.02).astype(np.float32)
x3 /= np.linalg.norm(x3, axis=1, keepdims=True)
slab3 = x3.astype(np.float16).view(np.uint16)
whole3 = fa.VectorIndex.from_slab(slab3)
index3 = fa.VectorIndex.from_slab(slab3)
sharded3 = ShardedVectorIndex(GpuShardBackend(index3, device, batched=True), overlap=True, force_collective=True)
qs3 = [(x3[rng.integers(0, n3, nq3)] + 0.15 * rng.standard_normal((nq3, 256))).astype(np.float32) for _ in range(steps3)]
tqs3 = [torch.from_numpy(a).to(device) for a in qs3]
late3, fb3 = [], []
outs3 = sharded3.search_steps(lambda i: tqs3[i], 0, steps3, k3, keep_all=True,
                              after_scan=lambda: (late3.append(sharded3.backend.last_late_answers), fb3.append(sharded3.backend.last_fallbacks)))
assert len(outs3) == steps3 and min(late3) > 0, (late3, fb3)
for i, (rows, scores, counts) in enumerate(outs3):
    want3 = [whole3.search_batch(qs3[i][s0:s0 + 64], k3, exact=True) for s0 in range(0, nq3, 64)]
    assert np.array_equal(rows.cpu().numpy().astype(np.uint32), np.concatenate([w[0] for w in want3])), f"re-filter step {i}: rows differ"
    assert np.array_equal(bits(scores.cpu().numpy()), bits(np.concatenate([w[1] for w in want3]))), f"re-filter step {i}: score bits differ"
dist.barrier()
dist.destroy_process_group()
print("exchange path OK: %d steps over a 1-rank RCCL group equal the unsharded index (+ %d steps with %d..%d fallbacks each, + %d steps with %d..%d "
      "late answers of which %d..%d exact fallbacks)" % (len(outs), steps2, min(fallbacks), max(fallbacks), steps3, min(late3), max(late3), min(fb3), max(fb3)),
      flush=True)
