"""The row-sharded index behind ONE C-ABI handle (fsgpu_sharded_*) against the oracle and against the
unsharded index: bit-exact row ids and f32 score bits.

Reference shape: scan_parallel's contiguous chunks + merge_partial_heaps
(crates/frankensearch-index/src/search.rs:1013-1036,1704-1720) and its test parallel_matches_sequential (:2084-2113).
On a one-GPU box the N-way logic is rehearsed with several shards on device 0 (peer-copy exchange; RCCL wants one rank per
device) and RCCL itself with a 1-rank communicator; with >= 2 GPUs visible the real all-gather runs over
min(visible, 8) devices.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import frankensearch_amd as fa_mod
    from frankensearch_amd.build import build

    build()
    assert fa_mod._lib.lib().fsgpu_device_count() >= 1, "no GPU visible"
    return fa_mod


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rand_slab(rng, n, dim):
    return rng.standard_normal((n, dim)).astype(np.float16).view(np.uint16)


def check_against_oracle(oracle, slab, live, queries, k, rows, scores, counts, hreduce=0):
    for qi in range(queries.shape[0]):
        er, es = oracle.search_top_k(slab, queries[qi], k, live=live, hreduce=hreduce)
        assert int(counts[qi]) == len(er), (qi, counts[qi], len(er))
        assert np.array_equal(rows[qi, :len(er)], er), qi
        assert np.array_equal(bits(scores[qi, :len(es)]), bits(es)), qi


@pytest.mark.parametrize("shards", [1, 2, 3, 8])
def test_virtual_shards_on_one_device_match_oracle(fa, oracle, shards):
    rng = np.random.default_rng(100 + shards)
    n, dim = 20_011, 64   # not a multiple of the shard count or of 64: ragged last shard, bitmap words split across shards
    slab = rand_slab(rng, n, dim)
    slab[7] = slab[n - 1]            # a tie across the first and the last shard: lower row wins
    live = rng.random(n) > 0.1
    q = rng.standard_normal((5, dim)).astype(np.float32)
    idx = fa.NativeShardedIndex.from_slab(slab, [0] * shards, live=live, exchange=fa.NativeShardedIndex.EXCHANGE_PEER_COPY)
    assert idx.shard_count() == shards and idx.record_count() == n and idx.dimension() == dim
    per = (n + shards - 1) // shards
    assert [idx.shard_range(r) for r in range(shards)] == [(min(n, r * per), min(n, (r + 1) * per)) for r in range(shards)]
    for k in (1, 10, 100, 256):
        rows, scores, counts = idx.search_batch(q, k)
        check_against_oracle(oracle, slab, live, q, k, rows, scores, counts)
    with pytest.raises(fa.DimensionMismatch):
        idx.search_batch(q[:, :32], 3)
    with pytest.raises(fa.InvalidConfig):
        idx.search_batch(q, 300)     # the exchange carries the fused tiers' packed lists
    idx.set_hreduce(2)
    rows, scores, counts = idx.search_batch(q, 10)
    check_against_oracle(oracle, slab, live, q, 10, rows, scores, counts, hreduce=2)
    idx.close()


def test_more_shards_than_rows_and_empty_index(fa, oracle):
    rng = np.random.default_rng(5)
    slab = rand_slab(rng, 3, 16)
    q = rng.standard_normal((2, 16)).astype(np.float32)
    idx = fa.NativeShardedIndex.from_slab(slab, [0] * 5, exchange=2)
    rows, scores, counts = idx.search_batch(q, 10)      # k > N: collect-all semantics, 3 hits
    check_against_oracle(oracle, slab, None, q, 10, rows, scores, counts)
    assert counts.tolist() == [3, 3]
    empty = fa.NativeShardedIndex.from_slab(np.zeros((0, 16), np.uint16), [0, 0], exchange=2)
    assert empty.search_batch(q, 4)[2].tolist() == [0, 0]


def test_batched_path_through_the_sharded_handle(fa, oracle):
    # every shard large enough for the matrix-core batched path (>= 4 x 8192 rows); same answers as the exact kernels
    rng = np.random.default_rng(9)
    n, dim, k = 140_000, 128, 10
    slab = oracle.clustered_corpus_f16(0, n, dim)
    q = np.stack([oracle.clustered_query(i, dim) for i in range(150)])
    whole = fa.VectorIndex.from_slab(slab)
    wr, ws, wc = whole.search_batch(q, k)
    for shards in (1, 2, 4):
        idx = fa.NativeShardedIndex.from_slab(slab, [0] * shards, exchange=2)
        rows, scores, counts, fb = idx.search_batch(q, k, batched=True)
        assert np.array_equal(rows, wr) and np.array_equal(bits(scores), bits(ws)) and np.array_equal(counts, wc)
        er, es, ec = idx.search_batch(q[:9], k)
        assert np.array_equal(er, wr[:9]) and np.array_equal(bits(es), bits(ws[:9]))
        idx.close()
    for qi in (0, 77, 149):
        orow, osc = oracle.search_top_k(slab, q[qi], k)
        assert np.array_equal(wr[qi], orow) and np.array_equal(bits(ws[qi]), bits(osc))


def test_rccl_all_gather_over_the_visible_devices(fa, oracle):
    """ncclCommInitAll + ncclAllGather inside libfsgpu.so: world = min(visible GPUs, 8) (1 on a one-GPU box: a 1-rank
    communicator still goes through RCCL's init, the collective launch and the gather layout)."""
    world = min(fa._lib.lib().fsgpu_device_count(), 8)
    rng = np.random.default_rng(12)
    n, dim, k = 50_000, 384, 10
    slab = rand_slab(rng, n, dim)
    q = rng.standard_normal((12, dim)).astype(np.float32)
    idx = fa.NativeShardedIndex.from_slab(slab, list(range(world)), exchange=fa.NativeShardedIndex.EXCHANGE_RCCL)
    assert idx.exchange_mode() == fa.NativeShardedIndex.EXCHANGE_RCCL and idx.shard_count() == world
    rows, scores, counts = idx.search_batch(q, k)
    check_against_oracle(oracle, slab, None, q, k, rows, scores, counts)
    brows, bscores, bcounts, _ = idx.search_batch(q, k, batched=True)
    assert np.array_equal(brows, rows) and np.array_equal(bits(bscores), bits(scores))
    # many calls back to back: the workers, the staging buffer and the communicator are reused
    for _ in range(20):
        r2, s2, _ = idx.search_batch(q, k)
        assert np.array_equal(r2, rows) and np.array_equal(bits(s2), bits(scores))
    idx.close()
    with pytest.raises(fa.InvalidConfig):
        fa.NativeShardedIndex.from_slab(slab, [0, 0], exchange=fa.NativeShardedIndex.EXCHANGE_RCCL)  # one rank per device


def test_torch_rccl_group_of_one_rank_runs_the_launcher_exchange_path(fa):
    """What `bench.py --gpus N` does on every rank, rehearsed with a ONE-rank RCCL process group (all this box has): the
    shard-local batched scan, `all_gather_into_tensor` over RCCL and the merge of the gathered [W, B, k] layout on a side stream
    underneath the next step's scan (search_begin / search_end), results equal to the unsharded index bit for bit.  In a process
    of its own (scripts/rehearse_launcher_exchange.py): torch's process group and the library's own communicators are not mixed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    res = subprocess.run([sys.executable, os.path.join(root, "scripts", "rehearse_launcher_exchange.py")], capture_output=True,
                         text=True, timeout=600, env=env)
    assert res.returncode == 0 and "exchange path OK" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def test_multi_gpu_matches_unsharded_bit_for_bit(fa, oracle):
    world = min(fa._lib.lib().fsgpu_device_count(), 8)
    if world < 2:
        pytest.skip("needs >= 2 GPUs (the driver's multi-GPU box); the one-GPU rehearsal is test_virtual_shards_*")
    n, dim, k = 400_000, 384, 10
    slab = oracle.clustered_corpus_f16(0, n, dim)
    q = np.stack([oracle.clustered_query(i, dim) for i in range(200)])
    whole = fa.VectorIndex.from_slab(slab)
    wr, ws, wc = whole.search_batch(q, k)
    idx = fa.NativeShardedIndex.from_slab(slab, list(range(world)))   # EXCHANGE_AUTO must pick RCCL here
    assert idx.exchange_mode() == fa.NativeShardedIndex.EXCHANGE_RCCL
    rows, scores, counts = idx.search_batch(q, k)
    assert np.array_equal(rows, wr) and np.array_equal(bits(scores), bits(ws)) and np.array_equal(counts, wc)
    rows, scores, counts, _ = idx.search_batch(q, k, batched=True)
    assert np.array_equal(rows, wr) and np.array_equal(bits(scores), bits(ws))
    peer = fa.NativeShardedIndex.from_slab(slab, list(range(world)), exchange=2)
    rows, scores, counts = peer.search_batch(q, k)
    assert np.array_equal(rows, wr) and np.array_equal(bits(scores), bits(ws))


@pytest.mark.parametrize("groups,shards", [(1, 1), (1, 2), (1, 3), (1, 8), (2, 2), (2, 4), (4, 2), (3, 1)])
def test_every_entry_point_of_the_sharded_handle_equals_the_unsharded_index(fa, oracle, groups, shards):
    """search_top_k(query, limit, filter) (search.rs:192-206, filter.rs:19-56), tombstone updates, dot_query_at routing and the
    int8 / 4-bit two-pass searches (search.rs:514-661, 876-946; ONE corpus-wide scale, simd.rs:1865-1886) on the sharded handle:
    row ids and f32 score bits of the unsharded index, whatever the number of shards — ragged last shard, bitmap words split
    across shards, two-pass candidates included — and whatever the layout: query groups x row shards (round 5: device r holds row
    shard r % shards and scans it for 1 / groups of every batch; 2 x 4 is the 8-GPU layout `bench.py --gpus 8` runs)."""
    S = fa.NativeShardedIndex
    rng = np.random.default_rng(300 + shards + 16 * groups)
    n, dim, k = 150_011, 128, 10
    slab = oracle.clustered_corpus_f16(0, n, dim)
    # one shard holds the corpus-wide max-abs: without the cross-shard reduction the others would quantise with finer scales
    slab[n - 5, 7] = np.float16(1.75).view(np.uint16)
    q = np.stack([oracle.clustered_query(i, dim) for i in range(70)])
    live = rng.random(n) > 0.15
    allow = rng.random(n) > 0.5
    whole = fa.VectorIndex.from_slab(slab, live=live)
    idx = S.from_slab(slab, [0] * (groups * shards), live=live, exchange=S.EXCHANGE_PEER_COPY, query_groups=groups)
    assert idx.shard_count() == groups * shards and idx.query_groups() == groups and idx.row_shards() == shards
    # filtered searches, exact kernels and matrix-core batched path
    wr, ws, wc = whole.search_batch(q, k, allow=allow)
    for mode in (S.EXACT, S.BATCHED):
        rows, scores, counts, _ = idx.search(q, k, mode, allow=allow)
        assert np.array_equal(rows, wr) and np.array_equal(bits(scores), bits(ws)) and np.array_equal(counts, wc), mode
    # a selective filter (the unsharded index gathers; the shards mask or gather as they see fit)
    few = np.zeros(n, bool)
    few[rng.choice(n, 900, replace=False)] = True
    fr, fs, fc = whole.search_batch(q[:6], k, allow=few)
    rows, scores, counts, _ = idx.search(q[:6], k, S.EXACT, allow=few)
    assert np.array_equal(rows, fr) and np.array_equal(bits(scores), bits(fs)) and np.array_equal(counts, fc)
    # tombstone update through the handle
    live2 = live.copy()
    live2[wr[:, 0]] = False          # every query loses its best hit
    whole.set_live(live2)
    idx.set_live(live2)
    wr2, ws2, wc2 = whole.search_batch(q, k)
    rows, scores, counts, _ = idx.search(q, k, S.BATCHED)
    assert np.array_equal(rows, wr2) and np.array_equal(bits(scores), bits(ws2)) and not np.array_equal(wr2, wr)
    idx.set_live(None)
    whole.set_live(None)
    # dot_query_at over rows of every shard (quality_scores_for_hits on a sharded quality tier)
    pick = rng.choice(n, 200, replace=False).astype(np.uint32)
    pick[:3] = (0, n - 1, n // 2)
    assert np.array_equal(bits(idx.gather_dot(q[3], pick)), bits(whole.gather_dot(q[3], pick)))
    # two-pass searches: the unsharded candidates, hence the unsharded hits
    for mode, mult, per_query in ((S.INT8_TWO_PASS, 3, whole.search_top_k_int8_two_pass), (S.FOURBIT_TWO_PASS, 5, whole.search_top_k_4bit_two_pass),
                                  (S.INT8_TWO_PASS, 1, whole.search_top_k_int8_two_pass)):
        rows, scores, counts, _ = idx.search(q, k, mode, candidate_multiplier=mult)
        for qi in range(q.shape[0]):
            hits = per_query(q[qi], k, mult)
            assert [h.index for h in hits] == rows[qi, :counts[qi]].tolist(), (mode, mult, qi)
            assert np.array_equal(bits([h.score for h in hits]), bits(scores[qi, :counts[qi]])), (mode, mult, qi)
        er, es = (oracle.search_int8_two_pass if mode == S.INT8_TWO_PASS else oracle.search_4bit_two_pass)(slab, q[0], k, mult)
        assert np.array_equal(rows[0, :counts[0]], er) and np.array_equal(bits(scores[0, :counts[0]]), bits(es))
    assert np.float32(idx.quant_scale_max()) == np.float32(np.abs(slab.view(np.float16).astype(np.float32)).max())
    # begin / end: two searches in flight, ended in order; a third begin is refused until one has ended
    t0 = idx.search_begin(q[:40], k, S.BATCHED)
    t1 = idx.search_begin(q[40:], 7, S.EXACT, allow=allow)
    with pytest.raises(fa.InvalidConfig):
        idx.search_begin(q[:2], k)
    r0 = idx.search_end(t0)
    r1 = idx.search_end(t1)
    w0, w1 = whole.search_batch(q[:40], k), whole.search_batch(q[40:], 7, allow=allow)
    assert np.array_equal(r0[0], w0[0]) and np.array_equal(bits(r0[1]), bits(w0[1]))
    assert np.array_equal(r1[0], w1[0]) and np.array_equal(bits(r1[1]), bits(w1[1])) and np.array_equal(r1[2], w1[2])
    with pytest.raises(fa.InvalidConfig):
        idx.search(q, 100, S.INT8_TWO_PASS, candidate_multiplier=3)     # k * multiplier > 256
    # LONE queries (nq = 1): every shard of one group answers through its own latency lane — the exact kernels with the query in the
    # argument block, the certified int8 pass once fsgpu_sharded_set_int8_latency is on, the two-pass lane — and the calling thread
    # merges the short lists; groups take lone queries in turn.  Same rows and score bits as the unsharded index, tombstones included.
    whole.set_live(live)
    idx.set_live(live)
    for lat in (False, True):
        idx.set_int8_latency(lat)
        for qi in range(2 * groups + 3):
            wr1, ws1, wc1 = whole.search_batch(q[qi], k)
            rows, scores, counts, _ = idx.search(q[qi], k, S.EXACT)
            assert np.array_equal(rows, wr1) and np.array_equal(bits(scores), bits(ws1)) and np.array_equal(counts, wc1), (lat, qi)
            for mode, mult, per_query in ((S.INT8_TWO_PASS, 3, whole.search_top_k_int8_two_pass), (S.FOURBIT_TWO_PASS, 5, whole.search_top_k_4bit_two_pass)):
                rows, scores, counts, _ = idx.search(q[qi], k, mode, candidate_multiplier=mult)
                hits = per_query(q[qi], k, mult)
                assert [h.index for h in hits] == rows[0, :counts[0]].tolist(), (lat, mode, qi)
                assert np.array_equal(bits([h.score for h in hits]), bits(scores[0, :counts[0]])), (lat, mode, qi)
    # a lone query begun, a batch begun behind it, both ended in order
    t0 = idx.search_begin(q[5], k, S.EXACT)
    t1 = idx.search_begin(q[:33], k, S.BATCHED)
    r0, r1 = idx.search_end(t0), idx.search_end(t1)
    w0, w1 = whole.search_batch(q[5], k), whole.search_batch(q[:33], k)
    assert np.array_equal(r0[0], w0[0]) and np.array_equal(bits(r0[1]), bits(w0[1]))
    assert np.array_equal(r1[0], w1[0]) and np.array_equal(bits(r1[1]), bits(w1[1]))
    # TWO lone tickets in flight (the API's two tickets): with one query group both queries land on the same shards, whose index
    # holds one lone query at a time — the second goes down the collective path; with several groups it takes the next group.
    # Every mode pair, ended in order and in reverse; each ticket must return ITS query's hits.
    for lat in (False, True):
        idx.set_int8_latency(lat)
        for (ma, mb) in ((S.EXACT, S.EXACT), (S.INT8_TWO_PASS, S.EXACT), (S.EXACT, S.FOURBIT_TWO_PASS), (S.INT8_TWO_PASS, S.INT8_TWO_PASS)):
            for rev in (False, True):
                ta = idx.search_begin(q[11], k, ma, candidate_multiplier=3)
                tb = idx.search_begin(q[12], k, mb, candidate_multiplier=3)
                if rev:
                    rb, ra = idx.search_end(tb), idx.search_end(ta)
                else:
                    ra, rb = idx.search_end(ta), idx.search_end(tb)
                for (res, qi, mode) in ((ra, 11, ma), (rb, 12, mb)):
                    if mode == S.EXACT:
                        w = whole.search_batch(q[qi], k)
                        assert np.array_equal(res[0], w[0]) and np.array_equal(bits(res[1]), bits(w[1])), (lat, ma, mb, rev, qi)
                    else:
                        hits = (whole.search_top_k_int8_two_pass if mode == S.INT8_TWO_PASS else whole.search_top_k_4bit_two_pass)(q[qi], k, 3)
                        assert [h.index for h in hits] == res[0][0, :res[2][0]].tolist(), (lat, ma, mb, rev, qi)
                        assert np.array_equal(bits([h.score for h in hits]), bits(res[1][0, :res[2][0]])), (lat, ma, mb, rev, qi)
        # a lone ticket and a batch ticket of every batch kind in flight together, in both orders: the lone lanes and the batch scans
        # share the shard index's workspaces (ws_partial_, the quantised / rotated queries) and are ordered on ONE stream
        for bmode in (S.EXACT, S.BATCHED, S.INT8_TWO_PASS):
            for lone_first in (True, False):
                if lone_first:
                    tl = idx.search_begin(q[7], k, S.EXACT)
                    tb = idx.search_begin(q[:40], k, bmode, candidate_multiplier=3)
                    rl, rb = idx.search_end(tl), idx.search_end(tb)
                else:
                    tb = idx.search_begin(q[:40], k, bmode, candidate_multiplier=3)
                    tl = idx.search_begin(q[7], k, S.EXACT)
                    rb, rl = idx.search_end(tb), idx.search_end(tl)
                w = whole.search_batch(q[7], k)
                assert np.array_equal(rl[0], w[0]) and np.array_equal(bits(rl[1]), bits(w[1])), (lat, bmode, lone_first)
                if bmode == S.INT8_TWO_PASS:
                    for qi in (0, 17, 39):
                        hits = whole.search_top_k_int8_two_pass(q[qi], k, 3)
                        assert [h.index for h in hits] == rb[0][qi, :rb[2][qi]].tolist(), (lat, lone_first, qi)
                else:
                    wb = whole.search_batch(q[:40], k)
                    assert np.array_equal(rb[0], wb[0]) and np.array_equal(bits(rb[1]), bits(wb[1])), (lat, bmode, lone_first)
    idx.close()


@pytest.mark.parametrize("n,dim,noise,clusters,nq,k", [(90_000, 256, 0.02, 4, 300, 10), (200_000, 384, 0.04, 24, 520, 33)])
def test_queries_refiltered_in_the_end_half_travel_again(fa, oracle, n, dim, noise, clusters, nq, k):
    """Tight clusters: the int8 filter's lists overflow and the END half of the shards' batched search hands those queries to the f16
    filter, which certifies them (no exact fallback).  Their corrected lists are written AFTER the exchange was enqueued: the handle has
    to exchange again (through round 5 it did so only for exact fallbacks — scripts/fuzz_sharded.py, round 6).  The unsharded twin
    must show the case really re-filters; every layout must return its rows and score bits; small slabs with ranks 25..32 (the
    extended group sample is for full-size samples only) ride along."""
    S = fa.NativeShardedIndex
    rng = np.random.default_rng(n + k)
    cent = rng.standard_normal((clusters, dim)).astype(np.float32)
    x = cent[rng.integers(0, clusters, n)] + (rng.standard_normal((n, dim)) * noise).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-9
    slab = x.astype(np.float16).view(np.uint16)
    q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.15).astype(np.float32)
    whole = fa.VectorIndex.from_slab(slab)
    for kk in (k, 30):
        ref = [np.concatenate(z) for z in zip(*[whole.search_batch(q[s0:s0 + 64], kk, exact=True) for s0 in range(0, nq, 64)])]
        st0 = whole.batched_filter_stats()
        r, s, c, fb = whole.search_batched(q, kk)
        refiltered = whole.batched_filter_stats()["refiltered_f16"] - st0["refiltered_f16"]
        assert np.array_equal(r, ref[0]) and np.array_equal(bits(s), bits(ref[1]))
        if kk == k:
            assert refiltered > 0, "the case no longer re-filters: pick another corpus"
        for groups, shards in ((1, 1), (1, 2), (2, 2)):
            idx = S.from_slab(slab, [0] * (groups * shards), exchange=S.EXCHANGE_PEER_COPY, query_groups=groups)
            r2, s2, c2, _ = idx.search(q, kk, S.BATCHED)
            assert np.array_equal(r2, ref[0]) and np.array_equal(bits(s2), bits(ref[1])), (kk, groups, shards)
            t1 = idx.search_begin(q[:nq // 2], kk, S.BATCHED)
            t2 = idx.search_begin(q[nq // 2:], kk, S.BATCHED)
            a, b = idx.search_end(t1), idx.search_end(t2)
            assert np.array_equal(np.concatenate([a[0], b[0]]), ref[0]) and np.array_equal(bits(np.concatenate([a[1], b[1]])), bits(ref[1])), (kk, groups, shards)
            idx.close()
    whole.close()


def test_lone_and_batch_tickets_overlap_on_a_rotated_filter_copy(fa, oracle):
    """The overlap cases above on a corpus with outlier channels: the batched search and the certified lone pass both go through the
    ROTATED int8 filter copy (rot_q_ is a workspace the lone lane and the batch's begin half share)."""
    S = fa.NativeShardedIndex
    n, dim, k = 120_000, 384, 10
    rng = np.random.default_rng(77)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[:, rng.integers(0, dim, 3)] *= 12.0          # outlier dimensions stretch the corpus-wide scale: the automatic rule rotates
    slab = np.ascontiguousarray((x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float16)).view(np.uint16)
    pick = rng.choice(n, 48, replace=False)
    q = slab[pick].view(np.float16).astype(np.float32)
    q += (0.05 * rng.standard_normal(q.shape)).astype(np.float32)
    whole = fa.VectorIndex.from_slab(slab)
    whole.search_batched(q[:40], k)
    assert whole.filter_rotated()
    for groups, shards in ((1, 2), (2, 2)):
        idx = S.from_slab(slab, [0] * (groups * shards), exchange=S.EXCHANGE_PEER_COPY, query_groups=groups)
        idx.set_int8_latency(True)
        wb = whole.search_batch(q[:40], k)
        for lone_first in (True, False):
            if lone_first:
                tl = idx.search_begin(q[41], k, S.EXACT)
                tb = idx.search_begin(q[:40], k, S.BATCHED)
                rl, rb = idx.search_end(tl), idx.search_end(tb)
            else:
                tb = idx.search_begin(q[:40], k, S.BATCHED)
                tl = idx.search_begin(q[41], k, S.EXACT)
                rb, rl = idx.search_end(tb), idx.search_end(tl)
            w = whole.search_batch(q[41], k)
            assert np.array_equal(rl[0], w[0]) and np.array_equal(bits(rl[1]), bits(w[1])), (groups, shards, lone_first)
            assert np.array_equal(rb[0], wb[0]) and np.array_equal(bits(rb[1]), bits(wb[1])), (groups, shards, lone_first)
        ta, tb2 = idx.search_begin(q[42], k, S.EXACT), idx.search_begin(q[43], k, S.EXACT)
        ra, rb2 = idx.search_end(ta), idx.search_end(tb2)
        for res, qi in ((ra, 42), (rb2, 43)):
            w = whole.search_batch(q[qi], k)
            assert np.array_equal(res[0], w[0]) and np.array_equal(bits(res[1]), bits(w[1])), (groups, shards, qi)
        idx.close()
    er, es = oracle.search_top_k(slab, q[41], k)
    w = whole.search_batch(q[41], k)
    assert np.array_equal(w[0][0], er) and np.array_equal(bits(w[1][0]), bits(es))


def test_queries_resident_in_parts_on_the_devices(fa, oracle):
    """fsgpu_sharded_search_parts: the batch's queries lie in parts in device memory (data-parallel encoders); every
    device fetches its query group's slice — part boundaries and group boundaries do not coincide here."""
    import torch
    S = fa.NativeShardedIndex
    rng = np.random.default_rng(9)
    n, dim, k = 90_001, 128, 10
    slab = oracle.clustered_corpus_f16(0, n, dim)
    q = np.stack([oracle.clustered_query(i, dim) for i in range(200)])
    whole = fa.VectorIndex.from_slab(slab)
    want = whole.search_batch(q, k)
    qd = torch.from_numpy(q).cuda()
    for groups, shards in ((1, 4), (2, 2), (4, 1), (2, 4)):
        idx = S.from_slab(slab, [0] * (groups * shards), exchange=S.EXCHANGE_PEER_COPY, query_groups=groups)
        for cuts in ([200], [50, 150], [1, 66, 133], [7, 7, 7, 179]):
            parts, at = [], 0
            for c in cuts:
                parts.append((qd[at:at + c].data_ptr(), c, 0))
                at += c
            for mode in (S.BATCHED, S.EXACT):
                rows, scores, counts, _ = idx.search_parts(parts, dim, k, mode)
                assert np.array_equal(rows, want[0]) and np.array_equal(bits(scores), bits(want[1])), (groups, shards, cuts, mode)
        idx.close()


def test_two_pass_on_tiny_and_ragged_shards(fa, oracle):
    # more shards than rows, fewer rows than candidates: still the unsharded answer
    S = fa.NativeShardedIndex
    rng = np.random.default_rng(77)
    for n, shards in ((5, 8), (40, 3), (700, 8)):
        slab = rand_slab(rng, n, 64)
        q = rng.standard_normal((4, 64)).astype(np.float32)
        whole = fa.VectorIndex.from_slab(slab)
        idx = S.from_slab(slab, [0] * shards, exchange=2)
        for mode, mult, fn in ((S.INT8_TWO_PASS, 3, whole.search_top_k_int8_two_pass), (S.FOURBIT_TWO_PASS, 5, whole.search_top_k_4bit_two_pass)):
            rows, scores, counts, _ = idx.search(q, 10, mode, candidate_multiplier=mult)
            for qi in range(4):
                hits = fn(q[qi], 10, mult)
                assert [h.index for h in hits] == rows[qi, :counts[qi]].tolist(), (n, shards, mode, qi)
                assert np.array_equal(bits([h.score for h in hits]), bits(scores[qi, :counts[qi]]))
        idx.close()


def test_sharded_fsvi_open_doc_ids_soft_delete_and_wal(fa, oracle, tmp_path):
    """fsgpu_sharded_open_fsvi: the file's record table, doc ids, tombstones and WAL live on the handle; search_top_k with WAL
    merge, shadowing and doc-id dedup (search.rs:426-494, 1449-1558) equals the unsharded VectorIndex.open of the same file."""
    S = fa.NativeShardedIndex
    rng = np.random.default_rng(91)
    n, dim = 4000, 64
    ids = [f"doc-{i % 3700:05d}" for i in range(n)]            # 300 duplicate doc ids
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    p = str(tmp_path / "sharded.fsvi")
    fa.write_fsvi(p, list(zip(ids, vecs)), "potion", "r1")
    whole = fa.VectorIndex.open(p)
    for shards in (1, 3, 8):
        idx = S.open(p, [0] * shards, exchange=2)
        assert idx.record_count() == n and idx.dimension() == dim and idx.shard_count() == shards
        assert [idx.doc_id_at(r) for r in (0, 1, n - 1)] == [whole.doc_id_at(r) for r in (0, 1, n - 1)]
        q = rng.standard_normal((6, dim)).astype(np.float32)

        def same(limit=12):
            for qi in range(q.shape[0]):
                a, b = idx.search_top_k(q[qi], limit), whole.search_top_k(q[qi], limit)
                assert [(h.index, h.doc_id) for h in a] == [(h.index, h.doc_id) for h in b], (shards, qi)
                assert np.array_equal(bits([h.score for h in a]), bits([h.score for h in b]))
        same()
        best = whole.search_top_k(q[0], 1)[0].doc_id
        assert idx.soft_delete(best) and whole.soft_delete(best)
        assert not idx.soft_delete("no-such-doc")
        same()
        fresh = rng.standard_normal(dim).astype(np.float32)
        second = whole.search_top_k(q[1], 1)[0].doc_id
        for h in (idx, whole):
            h.append("brand-new", q[2] * 3.0)                  # a WAL-only document that wins query 2
            h.append(second, fresh)                            # supersedes a main row
        assert idx.wal_record_count() == whole.wal_record_count() == 2
        same()
        assert idx.search_top_k(q[2], 3)[0].doc_id == "brand-new"
        idx.close()
        whole.close()
        whole = fa.VectorIndex.open(p)
