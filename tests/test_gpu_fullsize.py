"""Full-size (BASELINE config 3/4: 10M x 384 f16, ~7.7 GB in HBM) checks through size-independent
properties, plus a filtered-subset comparison against the oracle.  Runs on the GPU box only."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, DIM, K = 10_000_000, 384, 10


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def env():
    import torch
    import frankensearch_amd as fa
    from frankensearch_amd.build import build

    build()
    assert torch.cuda.is_available()
    sys.path.insert(0, ROOT)
    import bench

    dev = torch.device("cuda", 0)
    slab = bench.gen_corpus(0, N, DIM, dev)
    queries = bench.gen_queries(200, DIM, dev)
    idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), N, DIM, device=0, keepalive=slab)
    return {"torch": torch, "fa": fa, "slab": slab, "queries": queries, "idx": idx, "dev": dev}


def test_sorted_counts_and_rescoring_agree(env):
    idx, q = env["idx"], env["queries"].cpu().numpy()
    for k in (1, 10, 64, 200):
        rows, scores, counts = idx.search_batch(q[:4], k)
        assert np.all(counts == k)
        assert np.all(np.diff(scores, axis=1) <= 0)
        for qi in range(4):
            # same rows re-scored by the gather kernel (independent code path) must give identical bits
            assert np.array_equal(bits(idx.gather_dot(q[qi], rows[qi])), bits(scores[qi]))
            assert len(set(rows[qi].tolist())) == k


def test_planted_needles_are_found_in_order(env, oracle):
    torch, slab, idx = env["torch"], env["slab"], env["idx"]
    q = env["queries"][5].cpu().numpy()
    rng = np.random.default_rng(3)
    pos = np.sort(rng.choice(N, 12, replace=False))
    pos[0], pos[-1] = 0, N - 1  # first and last row of the slab
    coef = 1.2 + 0.05 * rng.permutation(12)
    saved = slab[torch.from_numpy(pos).to(slab.device)].clone()
    planted = np.stack([(q * c).astype(np.float16) for c in coef])
    slab[torch.from_numpy(pos).to(slab.device)] = torch.from_numpy(planted).to(slab.device)
    try:
        rows, scores, counts = idx.search_batch(q, 12)
        order = np.argsort(-coef)
        assert rows[0].tolist() == pos[order].tolist()
        want = [oracle.dot_f16_f32(planted[i].view(np.uint16), q) for i in order]
        assert np.array_equal(bits(scores[0]), bits(want))
    finally:
        slab[torch.from_numpy(pos).to(slab.device)] = saved


def test_allow_bitmap_subset_matches_oracle_on_gathered_rows(env, oracle):
    torch, slab, idx = env["torch"], env["slab"], env["idx"]
    q = env["queries"].cpu().numpy()
    rng = np.random.default_rng(9)
    sel = np.sort(rng.choice(N, 150_000, replace=False))
    allow = np.zeros(N, bool)
    allow[sel] = True
    host = slab[torch.from_numpy(sel).to(slab.device)].view(torch.int16).cpu().numpy().view(np.uint16)
    for k in (10, 100):
        rows, scores, counts = idx.search_batch(q[:3], k, allow=allow)
        for qi in range(3):
            er, es = oracle.search_top_k(host, q[qi], k, nthreads=8)
            assert np.array_equal(rows[qi], sel[er]) and np.array_equal(bits(scores[qi]), bits(es))


def test_shards_with_row_base_merge_to_whole(env):
    torch, fa, slab, idx = env["torch"], env["fa"], env["slab"], env["idx"]
    from frankensearch_amd.sharded import GpuShardBackend, shard_range
    q = env["queries"][:4].contiguous()
    whole_rows, whole_scores, _ = idx.search_batch(q.cpu().numpy(), K)
    world = 8
    lists = []
    backends = []
    for r in range(world):
        lo, hi = shard_range(N, r, world)
        sub = fa.VectorIndex.from_device_slab(slab[lo:hi].data_ptr(), hi - lo, DIM, device=0, row_base=lo, keepalive=slab)
        be = GpuShardBackend(sub, env["dev"])
        backends.append(be)
        lists.append(be.search_packed(q, K))
    gathered = torch.stack(lists).contiguous()  # [W, B, K] — the all-gather layout
    rows, scores, counts = backends[0].merge(gathered, K)
    torch.cuda.synchronize()
    assert np.array_equal(rows.cpu().numpy().view(np.uint32), whole_rows)
    assert np.array_equal(bits(scores.cpu().numpy()), bits(whole_scores))
    assert np.all(counts.cpu().numpy() == K)


def test_batched_shards_merge_to_whole(env):
    # the bench's N > 1 path: every rank answers the whole batch on its 1.25M-row shard through the batched matrix-core
    # scan (small shards skip the second sampling stage), packed hits are gathered and merged
    torch, fa, slab, idx = env["torch"], env["fa"], env["slab"], env["idx"]
    from frankensearch_amd.sharded import GpuShardBackend, shard_range
    q = env["queries"][:200].contiguous()
    whole_rows, whole_scores, _, fb = idx.search_batched(q.cpu().numpy(), K)
    exact_rows, exact_scores, _ = idx.search_batch(q[:16].cpu().numpy(), K)
    assert np.array_equal(whole_rows[:16], exact_rows) and np.array_equal(bits(whole_scores[:16]), bits(exact_scores))
    world = 8
    lists, backends, fallbacks = [], [], 0
    for r in range(world):
        lo, hi = shard_range(N, r, world)
        sub = fa.VectorIndex.from_device_slab(slab[lo:hi].data_ptr(), hi - lo, DIM, device=0, row_base=lo, keepalive=slab)
        be = GpuShardBackend(sub, env["dev"], batched=True)
        backends.append(be)
        lists.append(be.search_packed(q, K))
        fallbacks += be.last_fallbacks
    gathered = torch.stack(lists).contiguous()
    rows, scores, counts = backends[0].merge(gathered, K)
    torch.cuda.synchronize()
    assert np.array_equal(rows.cpu().numpy().view(np.uint32), whole_rows)
    assert np.array_equal(bits(scores.cpu().numpy()), bits(whole_scores))
    assert fallbacks < 200


def test_tombstone_removes_exactly_the_deleted_rows(env):
    fa, slab, idx = env["fa"], env["slab"], env["idx"]
    q = env["queries"][2].cpu().numpy()
    rows, scores, _ = idx.search_batch(q, 20)
    live = np.ones(N, bool)
    live[rows[0, [0, 3, 7]]] = False
    idx2 = fa.VectorIndex.from_device_slab(slab.data_ptr(), N, DIM, device=0, keepalive=slab)
    # device-slab indexes take the live bitmap through set_live (host words -> device copy)
    idx2.set_live(live)
    r2, s2, _ = idx2.search_batch(q, 17)
    keep = [i for i in range(20) if i not in (0, 3, 7)]
    assert r2[0].tolist() == rows[0, keep].tolist()
    assert np.array_equal(bits(s2[0]), bits(scores[0, keep]))


def test_ragged_multi_group_rounds_equal_the_per_query_searches(env):
    # 1,101 queries = one round of eight full 128-query groups + a 77-query tail group; every sample stage and selection of
    # the round is a single launch over all groups, so group boundaries (per-group offsets of the candidate lists, spill
    # areas, thresholds) are what this pins — for the f16 path and for the int8 two-pass path
    torch, idx = env["torch"], env["idx"]
    import sys
    sys.path.insert(0, ROOT)
    import bench
    q = bench.gen_queries(1101, DIM, env["dev"]).cpu().numpy()
    rows, scores, counts, fb = idx.search_batched(q, K)
    assert np.all(counts == K) and fb < 20
    pick = list(range(0, 1101, 41)) + [127, 128, 255, 256, 1023, 1024, 1100]
    er, es, _ = idx.search_batch(q[pick], K)
    assert np.array_equal(rows[pick], er) and np.array_equal(bits(scores[pick]), bits(es))
    # every answer is sorted, duplicate-free, and its scores are the exact dots of its rows
    assert np.all(np.diff(scores, axis=1) <= 0)
    for qi in range(0, 1101, 97):
        assert len(set(rows[qi].tolist())) == K
        assert np.array_equal(bits(idx.gather_dot(q[qi], rows[qi])), bits(scores[qi]))
    r8, s8, c8, fb8 = idx.search_int8_two_pass_batched(q[:300], K, 3)
    for qi in (0, 127, 128, 255, 256, 299):
        hits = idx.search_top_k_int8_two_pass(q[qi], K, 3)
        assert [h.index for h in hits] == r8[qi].tolist()
        assert np.array_equal(bits([h.score for h in hits]), bits(s8[qi]))


def test_batched_answers_equal_the_oracle_directly_at_full_size(env, oracle):
    """Eight answers of a 1,024-query batched search (matrix-core path, int8 filter, exact re-score) at 10M rows against the
    ORACLE on the same bytes — not against the per-query HIP kernels: rows and f32 score bits."""
    idx, q = env["idx"], env["queries"]
    import bench

    host = env["slab"].contiguous().view(env["torch"].int16).cpu().numpy().view(np.uint16)
    qh = bench.gen_queries(1024, DIM, env["dev"]).cpu().numpy()
    rows, scores, counts, fb = idx.search_batched(qh, K)
    threads = bench._oracle_threads()
    for qi in (0, 1, 255, 256, 511, 512, 777, 1023):
        er, es = oracle.search_top_k(host, qh[qi], K, nthreads=threads)
        assert counts[qi] == K and np.array_equal(rows[qi], er) and np.array_equal(bits(scores[qi]), bits(es)), qi
    st = idx.batched_filter_stats()
    assert st["int8_active"] and st["int8_queries"] >= 1024


def test_default_lone_query_path_equals_the_oracle_at_full_size(env, oracle):
    """fsgpu_search_topk for ONE query at 10M rows, as a host calls it by default: the index holds the int8 copy (the batched searches
    of this module built it), so the query takes the certified pass over that copy + the exact re-score (round 5) — its rows and f32
    score bits against the ORACLE on the same bytes, and against fsgpu_search_topk_exact (the exact f16 kernels)."""
    idx = env["idx"]
    import bench

    host = env["slab"].contiguous().view(env["torch"].int16).cpu().numpy().view(np.uint16)
    qh = bench.gen_queries(64, DIM, env["dev"]).cpu().numpy()
    idx.search_batched(qh, K)   # (makes sure the copy and its statistics exist)
    before = idx.batched_filter_stats()["int8_queries"]
    threads = bench._oracle_threads()
    for qi in range(12):
        rows, scores, counts = idx.search_batch(qh[qi], K)
        er2, es2, _ = idx.search_batch(qh[qi], K, exact=True)
        assert np.array_equal(rows, er2) and np.array_equal(bits(scores), bits(es2)), qi
        if qi < 6:
            er, es = oracle.search_top_k(host, qh[qi], K, nthreads=threads)
            assert counts[0] == K and np.array_equal(rows[0], er) and np.array_equal(bits(scores[0]), bits(es)), qi
    assert idx.batched_filter_stats()["int8_queries"] - before >= 8   # the certified pass answered (the clustered bench corpus certifies)


@pytest.mark.parametrize("kind", ["uniform", "outlier"])
def test_adversarial_corpora_at_full_size(env, kind):
    """SURVEY 8d's low-separation corpus (uniform-random unit vectors) and an anisotropic one with outlier dimensions and Zipf
    cluster sizes, 10M x 384 each: the batched path — whichever filter it ends up on, whatever it re-filters or hands to the
    exact kernels — returns the oracle's rows and score bits (8 answers against the oracle, 64 against the exact kernels)."""
    import bench

    res = bench.adversarial_section(kind, N, DIM, K, env["dev"], 0, steps=2)
    assert res["batched_equals_oracle_rows_and_bits"] and res["oracle_checked_queries"] == 8, res
    assert res["batched_equals_exact_kernels_64_queries"], res
    assert res["queries_per_sec"] > 0
