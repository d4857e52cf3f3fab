"""N > 1 path on CPU: world_size-2 gloo processes exercise the shard partitioning, the all-gather of
packed per-shard top-k and the [W, B, k] merge layout of frankensearch_amd.sharded, with the oracle as
the per-shard searcher (tests may use the oracle; the product backend is GpuShardBackend)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleShardBackend:
    """CPU stand-in with the ShardBackend protocol: oracle scan per shard, numpy merge under the reference order."""

    def __init__(self, slab, row_base):
        from oracle import oracle
        self.o = oracle
        self.slab = slab
        self.row_base = row_base

    def search_packed(self, queries, k):
        q = queries.numpy()
        out = np.full((q.shape[0], k), -1, dtype=np.int64)
        for i in range(q.shape[0]):
            if self.slab.shape[0] == 0:
                continue
            rows, scores = self.o.search_top_k(self.slab, q[i], k)
            packed = (scores.view(np.uint32).astype(np.uint64) << np.uint64(32)) | (rows.astype(np.uint64) + np.uint64(self.row_base))
            out[i, :len(rows)] = packed.view(np.int64)
        return torch.from_numpy(out)

    def merge(self, gathered, k):
        g = gathered.numpy().view(np.uint64)  # [W, B, k]
        w, b, kk = g.shape
        rows = np.full((b, k), 0xFFFFFFFF, dtype=np.uint32)
        scores = np.full((b, k), np.nan, dtype=np.float32)
        counts = np.zeros(b, dtype=np.int32)
        for qi in range(b):
            cand = [int(x) for x in g[:, qi, :].reshape(-1) if x != np.uint64(0xFFFFFFFFFFFFFFFF)]
            def key(p):
                s = np.uint32(p >> 32).view(np.float32)
                s = -np.inf if np.isnan(s) else float(s)
                return (-s, p & 0xFFFFFFFF)
            cand.sort(key=key)
            cand = cand[:k]
            counts[qi] = len(cand)
            for j, p in enumerate(cand):
                rows[qi, j] = p & 0xFFFFFFFF
                scores[qi, j] = np.uint32(p >> 32).view(np.float32)
        return torch.from_numpy(rows.view(np.int32)), torch.from_numpy(scores), torch.from_numpy(counts)


class PipelinedOracleBackend(OracleShardBackend):
    """... with the scan in two halves (GpuShardBackend.scan_begin / scan_end): ShardedVectorIndex.search_steps then enqueues step
    i + 1 before it ends step i — here the 'scan' simply happens in scan_begin, and every third one reports a fallback."""
    supports_pipelined_scans = True

    def __init__(self, slab, row_base):
        super().__init__(slab, row_base)
        self.begun, self.ended, self.last_fallbacks = 0, [], 0

    def scan_begin(self, queries, k, packed):
        assert packed
        self.begun += 1
        assert self.begun - len(self.ended) <= 2, "more than two scans outstanding"
        return self.search_packed(queries, k), (self.begun, queries)

    def scan_end(self, ticket):
        assert ticket[0] == len(self.ended) + 1, "scans are ended in the order they were begun"
        self.ended.append(ticket[0])
        self.last_fallbacks = 1 if ticket[0] % 3 == 0 else 0
        return self.last_fallbacks


def _worker(rank, world, port, n, dim, k, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from frankensearch_amd.sharded import ShardedVectorIndex, shard_range
    rng = np.random.default_rng(5)
    slab = rng.standard_normal((n, dim)).astype(np.float16).view(np.uint16)
    if n > 4:
        slab[3] = slab[n - 2]  # a cross-shard tie
    queries = rng.standard_normal((3, dim)).astype(np.float32)
    lo, hi = shard_range(n, rank, world)
    idx = ShardedVectorIndex(OracleShardBackend(slab[lo:hi], lo))
    rows, scores, counts = idx.search(torch.from_numpy(queries), k)
    # the two halves bench.py pipelines (scan of step i+1 over the exchange of step i) give the same answer
    r2, s2, c2 = idx.search_end(idx.search_begin(torch.from_numpy(queries), k), k)
    assert torch.equal(r2, rows) and torch.equal(c2, counts) and torch.equal(s2.view(torch.int32), scores.view(torch.int32))
    # ... and the launcher's step loop (a backend without the after-enqueue window: the loop does the exchange itself)
    for r3, s3, c3 in idx.search_steps(lambda i: torch.from_numpy(queries), 0, 3, k, keep_all=True):
        assert torch.equal(r3, rows) and torch.equal(c3, counts) and torch.equal(s3.view(torch.int32), scores.view(torch.int32))
    # ... and the same loop over a backend whose scan comes in two halves (what the GPU ranks run)
    pidx = ShardedVectorIndex(PipelinedOracleBackend(slab[lo:hi], lo))
    seen = []
    outs = pidx.search_steps(lambda i: torch.from_numpy(queries), 0, 5, k, after_scan=lambda: seen.append(pidx.backend.last_fallbacks), keep_all=True)
    assert len(outs) == 5 and pidx.backend.ended == [1, 2, 3, 4, 5] and seen == [0, 0, 1, 0, 0]
    for r4, s4, c4 in outs:
        assert torch.equal(r4, rows) and torch.equal(c4, counts) and torch.equal(s4.view(torch.int32), scores.view(torch.int32))
    last = pidx.search_steps(lambda i: torch.from_numpy(queries), 5, 1, k)
    assert torch.equal(last[0], rows) and pidx.backend.ended[-1] == 6
    if rank == 0:
        ret["rows"] = rows.numpy().view(np.uint32).copy()
        ret["scores"] = scores.numpy().copy()
        ret["counts"] = counts.numpy().copy()
        ret["slab"] = slab
        ret["queries"] = queries
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,k,world", [(1001, 10, 2), (7, 10, 2), (64, 64, 2), (1001, 10, 3), (3, 5, 4)])
def test_world2_allgather_merge_equals_unsharded_oracle(oracle, n, k, world):
    # world 3: uneven shards; (3 rows, 4 ranks): one rank owns no row at all
    dim = 40
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, dim, k, ret), nprocs=world, join=True)
    slab, queries = ret["slab"], ret["queries"]
    for qi in range(queries.shape[0]):
        er, es = oracle.search_top_k(slab, queries[qi], k)
        c = int(ret["counts"][qi])
        assert c == len(er)
        assert np.array_equal(ret["rows"][qi, :c], er)
        assert np.array_equal(ret["scores"][qi, :c].view(np.uint32), es.view(np.uint32))


def _hybrid_worker(rank, world, port, groups, n, nq, dim, k, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from frankensearch_amd.sharded import ShardedVectorIndex, shard_range
    rng = np.random.default_rng(11)
    slab = rng.standard_normal((n, dim)).astype(np.float16).view(np.uint16)
    slab[3] = slab[n - 2]  # a tie across row shards
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    shards = world // groups
    lo, hi = shard_range(n, rank % shards, shards)   # rank r holds row shard r % S
    idx = ShardedVectorIndex(OracleShardBackend(slab[lo:hi], lo), query_groups=groups)
    rows, scores, counts = idx.search(torch.from_numpy(queries), k)
    assert rows.shape == (nq, k)
    # the two halves, and the pipelined step loop over a two-half backend (ragged batches: the last groups hold fewer queries, or none)
    r2, s2, c2 = idx.search_end(idx.search_begin(torch.from_numpy(queries), k), k)
    assert torch.equal(r2, rows) and torch.equal(c2, counts) and torch.equal(s2.view(torch.int32), scores.view(torch.int32))
    pidx = ShardedVectorIndex(PipelinedOracleBackend(slab[lo:hi], lo), query_groups=groups)
    for r4, s4, c4 in pidx.search_steps(lambda i: torch.from_numpy(queries), 0, 4, k, keep_all=True):
        assert torch.equal(r4, rows) and torch.equal(c4, counts) and torch.equal(s4.view(torch.int32), scores.view(torch.int32))
    for r3, s3, c3 in idx.search_steps(lambda i: torch.from_numpy(queries), 0, 3, k, keep_all=True):
        assert torch.equal(r3, rows) and torch.equal(c3, counts) and torch.equal(s3.view(torch.int32), scores.view(torch.int32))
    if rank == 0:
        ret["rows"] = rows.numpy().view(np.uint32).copy()
        ret["scores"] = scores.numpy().copy()
        ret["counts"] = counts.numpy().copy()
        ret["slab"] = slab
        ret["queries"] = queries
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,groups,n,nq,k", [(4, 2, 1001, 7, 10), (4, 4, 500, 5, 10), (2, 2, 64, 1, 5), (4, 2, 37, 2, 64)])
def test_query_groups_x_row_shards_equal_unsharded_oracle(oracle, world, groups, n, nq, k):
    # the hybrid layout of round 5: world = query groups x row shards; (world 4, 2 groups) = 2 x 2, (4, 4) = four replicas each taking
    # a quarter of the batch, nq 1 with 2 groups / nq 5 with 4: groups without a query take part in the all-gather with empty lists
    dim = 40
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_hybrid_worker, args=(world, _free_port(), groups, n, nq, dim, k, ret), nprocs=world, join=True)
    slab, queries = ret["slab"], ret["queries"]
    for qi in range(queries.shape[0]):
        er, es = oracle.search_top_k(slab, queries[qi], k)
        c = int(ret["counts"][qi])
        assert c == len(er)
        assert np.array_equal(ret["rows"][qi, :c], er)
        assert np.array_equal(ret["scores"][qi, :c].view(np.uint32), es.view(np.uint32))


def test_shard_range_is_a_contiguous_partition():
    from frankensearch_amd.sharded import shard_range
    for n in (0, 1, 7, 8, 9, 10_000_000, 50_000_001):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(hi >= lo for lo, hi in spans)
