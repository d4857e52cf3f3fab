"""The int8 filter of the exact batched search (mfma_scan.hip: prepare_queries_i8_filter_kernel; int8_kernels.hip:
i8_slab_stats_kernel).

The batched search scores the slab approximately on the matrix cores and re-scores, in the reference's operation order
(crates/frankensearch-index/src/simd.rs:398-446), every row that could reach the top k under a PROVEN bound on the
approximation error.  With the int8 copy of the slab as the filter that bound comes from measured statistics of the copy and
of each quantised query.  These tests check (1) the bound itself, against float64 arithmetic and against the exact kernels'
scores, on benign and hostile data; (2) that the hits are the exact search's bit for bit whichever filter ran, including
the hand-over of uncertified queries to the f16 filter and the cases no int8 bound exists for; (3) the automatic choice.
"""
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


def unit_rows(rng, n, dim):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def corpora(rng, n, dim):
    base = unit_rows(rng, n, dim)
    yield "gaussian unit rows", base
    out = base.copy()
    out[:, rng.integers(0, dim, 3)] *= 12.0          # outlier dimensions stretch the corpus-wide scale
    yield "outlier dimensions", out / np.linalg.norm(out, axis=1, keepdims=True)
    yield "tiny magnitudes", base * 3e-3
    yield "large magnitudes", base * 180.0
    cent = unit_rows(rng, 16, dim)
    clustered = cent[rng.integers(0, 16, n)] + 0.3 * rng.standard_normal((n, dim)).astype(np.float32) / np.sqrt(dim)
    yield "clusters", clustered
    sparse = base * (rng.random((n, dim)) < 0.1)       # mostly zeros: quantisation error far below the worst case
    yield "sparse rows", sparse.astype(np.float32)
    one = np.zeros((n, dim), np.float32)
    one[np.arange(n), rng.integers(0, dim, n)] = rng.choice([-1.0, 1.0], n)
    one[::3] = base[::3]
    yield "one-hot and dense rows mixed", one


def hostile_queries(rng, rows, dim):
    nq = 24
    q = rows[rng.integers(0, rows.shape[0], nq)] + (0.2 * rng.standard_normal((nq, dim))).astype(np.float32)
    q[1] *= 37.5
    q[2] *= 1e-6
    q[3] = 0.0
    q[3, 7] = 1.0                                     # one-hot: |p|_1 far below sqrt(dim) |p|_2
    q[4] = np.sign(q[4]) * 0.25                       # every element at the scale's edge: |eta| ~ 0
    q[5] = np.sign(q[5]) * ((rng.integers(0, 126, dim) + 0.5) / 127.0).astype(np.float32)   # ... half a step off a level:
    q[5, 0] = 1.0                                                                              # worst-case eta
    q[6, :] = 0.003
    q[6, 0] = 1.0                                     # one dominant element: the rest quantises to zero
    q[7] = rng.standard_normal(dim).astype(np.float32) * 900.0
    return q


@pytest.mark.parametrize("dim", [128, 384])
def test_bound_covers_every_row_against_float64_and_the_exact_kernels(fa, oracle, dim):
    rng = np.random.default_rng(1000 + dim)
    n = 12_000
    worst = 0.0
    for name, rows in corpora(rng, n, dim):
        slab = rows.astype(np.float16).view(np.uint16)
        idx = fa.VectorIndex.from_slab(slab)
        idx.set_filter_rotation(1)   # (the unrotated copy: the reference's own int8 slab; the rotated one has its own test below)
        q = hostile_queries(rng, rows, dim)
        delta, qscale, sscale, qi8, slab_i8 = idx.int8_filter_bound(q, want_slab=True)
        # the filter uses the reference's quantisers (simd.rs:1865-1886, search.rs:1616-1626): same bytes as the oracle's
        assert np.array_equal(slab_i8, oracle.quantize_slab_i8(slab)), name
        for i in range(q.shape[0]):
            assert np.array_equal(qi8[i], oracle.quantize_query_i8(q[i])), (name, i)
        assert np.all(delta > 0), (name, delta)       # all of these are certifiable
        # the kernels' delta is the one oracle/filter_bound.py restates (whose own CPU test checks it against float64)
        from oracle import filter_bound as fb
        stats = fb.slab_stats(slab)
        assert abs(float(stats[0]) - sscale) <= 1e-6 * sscale, name
        for i in range(q.shape[0]):
            want = fb.query_bound(q[i], stats, dim)[0]
            assert abs(float(delta[i]) - want) <= 2e-3 * want + 2.0, (name, i, float(delta[i]), want)
        idot = slab_i8.astype(np.int64) @ qi8.astype(np.int64).T                       # [n, nq], exact
        x64 = slab.view(np.float16).astype(np.float64)
        s64 = x64 @ q.astype(np.float64).T                                             # real-number scores
        unit = np.float64(sscale) * qscale.astype(np.float64)                          # integer-score units per score unit
        err64 = np.abs(idot - s64 * unit[None, :])
        assert np.all(err64 <= delta[None, :].astype(np.float64)), (name, float((err64 / delta[None, :]).max()))
        # ... and the scores the exact kernels emit (the reference's f32 operation order) sit inside the same bound
        for i in (0, 1, 2, 5, 7):
            exact = idx.gather_dot(q[i], np.arange(n, dtype=np.uint32)).astype(np.float64)
            assert np.all(np.abs(idot[:, i] - exact * unit[i]) <= float(delta[i])), (name, i)
        worst = max(worst, float((err64 / delta[None, :]).max()))
        idx.close()
    # the bound is not vacuous: somewhere the measured error comes within a factor of a few of it
    assert worst > 0.05, worst


@pytest.mark.parametrize("dim", [128, 384])
def test_rotated_filter_copy_keeps_the_bound_and_the_exact_bits(fa, oracle, dim):
    """Round 5: for a slab with outlier channels the filter's int8 copy holds quantised rows of R x (R a fixed random orthogonal
    matrix, applied in f64, rounded once to f32) and scores them against quantised R q.  The PROVEN bound must still cover every
    (row, query) pair — |idot - S c_s c_q| <= delta with S the real-number dot of the ORIGINAL row and query — on the seven corpora x
    hostile queries, rotation forced on; the automatic rule picks the rotation for the outlier corpus and not for the Gaussian one;
    on the outlier corpus the rotated margin is at most half the unrotated one in score units; and the batched search returns the
    exact kernels' rows and score bits either way."""
    rng = np.random.default_rng(2000 + dim)
    n = 12_000
    for name, rows in corpora(rng, n, dim):
        slab = rows.astype(np.float16).view(np.uint16)
        idx = fa.VectorIndex.from_slab(slab)
        idx.set_filter_rotation(2)
        q = hostile_queries(rng, rows, dim)
        delta, qscale, sscale, qi8, slab_i8 = idx.int8_filter_bound(q, want_slab=True)
        assert idx.filter_rotated()
        assert np.all(delta > 0), (name, delta)
        idot = slab_i8.astype(np.int64) @ qi8.astype(np.int64).T
        x64 = slab.view(np.float16).astype(np.float64)
        s64 = x64 @ q.astype(np.float64).T                                             # real-number scores of the ORIGINAL vectors
        unit = np.float64(sscale) * qscale.astype(np.float64)
        err64 = np.abs(idot - s64 * unit[None, :])
        # (the query scale comes back through an f32 division: 1e-6 relative on S c_s c_q, far inside delta's + 1)
        slack = 2e-6 * np.abs(s64 * unit[None, :])
        assert np.all(err64 <= delta[None, :].astype(np.float64) + slack), (name, float((err64 / delta[None, :]).max()))
        for i in (0, 1, 2, 5, 7):
            exact = idx.gather_dot(q[i], np.arange(n, dtype=np.uint32)).astype(np.float64)
            assert np.all(np.abs(idot[:, i] - exact * unit[i]) <= float(delta[i]) + 2e-6 * np.abs(exact * unit[i])), (name, i)
        idx.close()
    # the automatic rule, the margins, and the hits
    k = 10
    for name, rows in corpora(rng, 70_000, dim):
        if name not in ("gaussian unit rows", "outlier dimensions"):
            continue
        slab = rows.astype(np.float16).view(np.uint16)
        q = rows[rng.integers(0, rows.shape[0], 300)] + (0.2 * rng.standard_normal((300, dim)) / np.sqrt(dim)).astype(np.float32)
        auto, off = fa.VectorIndex.from_slab(slab), fa.VectorIndex.from_slab(slab)
        off.set_filter_rotation(1)
        for idx in (auto, off):
            idx.set_batched_filter(2)
        d_auto, qs_auto, ss_auto, _, _ = auto.int8_filter_bound(q[:16])
        d_off, qs_off, ss_off, _, _ = off.int8_filter_bound(q[:16])
        assert auto.filter_rotated() == (name == "outlier dimensions") and not off.filter_rotated(), name
        if name == "outlier dimensions":   # margins in score units: delta / (c_s c_q)
            m_auto = np.median(d_auto / (ss_auto * qs_auto)), np.median(d_off / (ss_off * qs_off))
            assert m_auto[0] < 0.5 * m_auto[1], m_auto
        er, es, ec = [np.concatenate(z) for z in zip(*[auto.search_batch(q[s0:s0 + 60], k, exact=True) for s0 in range(0, 300, 60)])]
        for idx in (auto, off):
            br, bs, bc, fb = idx.search_batched(q, k)
            assert np.array_equal(br, er) and np.array_equal(bits(bs), bits(es)) and np.array_equal(bc, ec), (name, idx is auto)
        orow, osc = oracle.search_top_k(slab, q[0], k)
        br, bs, _, _ = auto.search_batched(q[:16], k)
        assert np.array_equal(br[0], orow) and np.array_equal(bits(bs[0]), bits(osc)), name
        # a lone query through the certified pass over the (rotated) copy
        for i in range(6):
            r1 = auto.search_batch(q[i], k)
            assert np.array_equal(r1[0][0], er[i]) and np.array_equal(bits(r1[1][0]), bits(es[i])), (name, i)
        # an allow bitmap and tombstones over the same (rotated) copy: the filter sees them through the kernels' bitmaps, not through the copy
        allow = rng.random(rows.shape[0]) < 0.5
        fr_, fs_, fc_ = [np.concatenate(z) for z in zip(*[auto.search_batch(q[s0:s0 + 60], k, allow=allow, exact=True) for s0 in range(0, 300, 60)])]
        br, bs, bc, fb = auto.search_batched(q, k, allow=allow)
        assert np.array_equal(br, fr_) and np.array_equal(bits(bs), bits(fs_)) and np.array_equal(bc, fc_), (name, "allow bitmap")
        live = rng.random(rows.shape[0]) < 0.9
        auto.set_live(live)
        lr, ls, lc = [np.concatenate(z) for z in zip(*[auto.search_batch(q[s0:s0 + 60], k, exact=True) for s0 in range(0, 300, 60)])]
        br, bs, bc, fb = auto.search_batched(q, k)
        assert np.array_equal(br, lr) and np.array_equal(bits(bs), bits(ls)) and np.array_equal(bc, lc), (name, "tombstones")
        assert live[br[bc[:, None] > np.arange(k)[None, :]]].all(), name
        auto.close()
        off.close()


def test_uncertifiable_inputs_are_marked_and_still_answered_exactly(fa, oracle):
    rng = np.random.default_rng(7)
    n, dim, k = 70_000, 384, 10
    rows = unit_rows(rng, n, dim)
    q = rows[rng.integers(0, n, 40)] + (0.2 * rng.standard_normal((40, dim))).astype(np.float32)
    q[0] = 0.0                       # zero query
    q[1, 3] = np.nan
    q[2, 5] = np.inf
    q[3, 9] = 70000.0                # finite, but an f32 product with an f16 element could overflow past the bound's reach
    q[4] *= 1e-38                    # scales overflow: no finite bound
    slab = rows.astype(np.float16).view(np.uint16)
    idx = fa.VectorIndex.from_slab(slab)
    delta = idx.int8_filter_bound(q)[0]
    assert np.all(delta[:5] < 0) and np.all(delta[5:] > 0), delta[:8]
    idx.set_batched_filter(2)
    br, bs, bc, fb = idx.search_batched(q, k)
    er, es, ec = idx.search_batch(q, k)
    assert np.array_equal(bc, ec) and np.array_equal(br, er) and np.array_equal(bits(bs), bits(es))
    st = idx.batched_filter_stats()
    assert st["int8_queries"] == 40 and 5 <= st["refiltered_f16"] <= 8, st
    for qi in (0, 3, 4, 17):
        orow, osc = oracle.search_top_k(slab, q[qi], k)
        assert np.array_equal(br[qi], orow) and np.array_equal(bits(bs[qi]), bits(osc)), qi
    idx.close()
    # a slab with a NaN, an infinity, or nothing but zeros has no int8 bound at all: every query is handed on
    for poison in (0x7e00, 0x7c00, 0xfc00, None):
        bad = slab.copy()
        if poison is None:
            bad[:] = 0
        else:
            bad[12345, 17] = poison
        idx = fa.VectorIndex.from_slab(bad)
        assert np.all(idx.int8_filter_bound(q)[0] < 0)
        idx.set_batched_filter(2)
        br, bs, bc, fb = idx.search_batched(q, k)
        er, es, ec = idx.search_batch(q, k)
        assert np.array_equal(bc, ec) and np.array_equal(br, er) and np.array_equal(bits(bs), bits(es)), poison
        assert idx.batched_filter_stats()["refiltered_f16"] == 40
        idx.close()


def test_both_filters_emit_the_exact_search_bits_on_ties_filters_and_tombstones(fa, oracle):
    rng = np.random.default_rng(11)
    n, dim = 220_003, 384
    cent = unit_rows(rng, 40, dim)
    # (the reference bench's recipe: unit centroids + noise of 0.3 per ELEMENT, so neighbours in a cluster spread over ~0.05 in
    # cosine — tighter clusters than the int8 margin are the subject of the last test)
    rows = cent[np.sort(rng.integers(0, 40, n))] + 0.3 * rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    rows[5000:5060] = rows[4999]          # a run of identical rows: integer AND exact scores tie, lower row wins
    rows[n - 3:] = rows[4999]
    slab = rows.astype(np.float16).view(np.uint16)
    live = rng.random(n) > 0.1
    allow = rng.random(n) > 0.5
    nq = 530                              # 4 groups on the 512-query int8 shape + a ragged tail
    q = cent[rng.integers(0, 40, nq)] + 0.3 * rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    q[2] = rows[4999]
    idx = fa.VectorIndex.from_slab(slab, live=live)
    for k, mask in ((10, None), (64, None), (1, allow), (10, allow)):
        exact = [idx.search_batch(q[s:s + 64], k, allow=mask) for s in range(0, nq, 64)]
        er = np.concatenate([e[0] for e in exact])
        es = np.concatenate([e[1] for e in exact])
        ec = np.concatenate([e[2] for e in exact])
        for filt in (2, 1, 0):
            idx.set_batched_filter(filt)
            br, bs, bc, fb = idx.search_batched(q, k, allow=mask)
            assert np.array_equal(bc, ec) and np.array_equal(br, er) and np.array_equal(bits(bs), bits(es)), (k, filt)
            assert fb < nq // 8, (k, filt, fb)
    eff = live & allow
    for qi in (2, 300, 529):
        orow, osc = oracle.search_top_k(slab, q[qi], 10, live=eff)
        assert np.array_equal(br[qi], orow) and np.array_equal(bits(bs[qi]), bits(osc)), qi
    st = idx.batched_filter_stats()
    assert st["int8_queries"] >= 8 * nq and st["refiltered_f16"] * 8 < st["int8_queries"], st
    idx.close()


def test_automatic_choice_leaves_the_int8_filter_when_its_margin_does_not_separate(fa, oracle):
    """Rows whose scores against queries near their centre spread by ~2e-3: twenty times the f16 filter's margin, a fraction of
    the int8 one's.  A cluster of 6,000 such rows sits entirely inside the int8 margin but inside its finish's reach (8,192
    candidates re-scored exactly per query): answered on the int8 filter.  A cluster of 20,000 is beyond it: those queries
    are handed to the f16 filter (which separates them), and after two such batches the index stays with the f16 filter."""
    rng = np.random.default_rng(5)
    n, dim, k = 120_000, 384, 10
    base = unit_rows(rng, 1, dim)[0]
    nq = 64
    q = base + 0.2 * rng.standard_normal((nq, dim)).astype(np.float32) / np.sqrt(dim)
    for members_n, handed_on in ((6000, False), (20_000, True)):
        rows = unit_rows(rng, n, dim)
        members = rng.choice(n, members_n, replace=False)
        rows[members] = base + 0.2 * rng.standard_normal((members_n, dim)).astype(np.float32) / np.sqrt(dim)
        rows[members] /= np.linalg.norm(rows[members], axis=1, keepdims=True)
        slab = rows.astype(np.float16).view(np.uint16)
        idx = fa.VectorIndex.from_slab(slab)
        er, es, ec = idx.search_batch(q, k)
        assert idx.batched_filter_stats()["int8_active"]
        for round_ in range(3):
            br, bs, bc, fb = idx.search_batched(q, k)
            assert np.array_equal(bc, ec) and np.array_equal(br, er) and np.array_equal(bits(bs), bits(es)), (members_n, round_)
            assert fb <= 2, fb                       # certified by one filter or the other: no exact-kernel passes
        st = idx.batched_filter_stats()
        if handed_on:
            assert st["int8_queries"] == 2 * nq and st["refiltered_f16"] > nq and not st["int8_active"], st
            idx.set_batched_filter(2)                # pinned to int8 it keeps trying (and keeps handing on): same bits
            br, bs, bc, fb = idx.search_batched(q, k)
            assert np.array_equal(br, er) and np.array_equal(bits(bs), bits(es))
            assert idx.batched_filter_stats()["int8_queries"] == 3 * nq
        else:
            assert st["int8_queries"] == 3 * nq and st["refiltered_f16"] <= 2 and st["int8_active"], st
        orow, osc = oracle.search_top_k(slab, q[0], k)
        assert np.array_equal(br[0], orow) and np.array_equal(bits(bs[0]), bits(osc))
        idx.close()


def test_rows_next_to_a_nan_row_keep_their_place(fa, oracle):
    """A row with a NaN element scores NaN and ranks last (score_key, search.rs:1655-1686) — its NEIGHBOURS must not be affected.
    (Found by scripts/fuzz_batched.py once it drew slabs with non-finite values: the LDS-query kernel rejected a lane's four rows
    together when the second of them scored NaN — `a > NaN ? a : NaN` — so the best hits next to a poisoned row went missing on
    the f16 filter.)  The planted NaN rows sit at every position of a four-row group, right next to each query's best hits."""
    rng = np.random.default_rng(61)
    n, dim, k = 60_000, 128, 10
    rows = unit_rows(rng, n, dim)
    targets = np.array([4000, 8001, 12002, 16003, 20004, 24005, 28006, 32007], dtype=np.int64)   # all residues mod 4 (and mod 16)
    q = np.tile(rows[targets], (40, 1))[:300] + (0.05 * rng.standard_normal((300, dim))).astype(np.float32) / np.sqrt(dim)
    slab = rows.astype(np.float16).view(np.uint16).copy()
    for t in targets:                      # poison the rows around each target, never the target itself
        for off in (-2, -1, 1, 2):
            slab[t + off, int(rng.integers(0, dim))] = 0x7e00
    idx = fa.VectorIndex.from_slab(slab)
    for nq in (70, 128, 300):              # 64-query, 128-query and wide main passes
        er, es, ec = [np.concatenate(z) for z in zip(*[idx.search_batch(q[s:min(s + 64, nq)], k) for s in range(0, nq, 64)])]
        assert all(int(er[i, 0]) == int(targets[i % 8]) for i in range(nq))      # the target is the best hit (exact kernels)
        for filt in (1, 2):
            idx.set_batched_filter(filt)
            br, bs, bc, fb = idx.search_batched(q[:nq], k)
            assert np.array_equal(bc, ec) and np.array_equal(br, er) and np.array_equal(bits(bs), bits(es)), (nq, filt)
    orow, osc = oracle.search_top_k(slab, q[3], k)
    assert np.array_equal(br[3], orow) and np.array_equal(bits(bs[3]), bits(osc))
    idx.close()


def test_int8_latency_path_returns_the_exact_kernels_answers(fa, oracle):
    """fsgpu_index_set_int8_latency: few-query fsgpu_search_topk calls through the int8 filter + exact re-score — every output
    word (rows, score bits, counts, padding) as the exact kernels give it, incl. queries the filter cannot certify, k above the
    number of live rows, and a filtered call (which keeps the exact path)."""
    rng = np.random.default_rng(77)
    n, dim = 90_000, 384
    rows = unit_rows(rng, n, dim)
    slab = rows.astype(np.float16).view(np.uint16)
    live = np.zeros(n, bool)
    live[rng.choice(n, 40_000, replace=False)] = True
    q = rows[rng.integers(0, n, 16)] + (0.2 * rng.standard_normal((16, dim))).astype(np.float32)
    q[1] = 0.0
    q[2, 7] = np.nan
    q[3] *= 250.0
    a, b = fa.VectorIndex.from_slab(slab, live=live), fa.VectorIndex.from_slab(slab, live=live)
    b.set_int8_latency(True)
    for nq in (1, 3, 16):
        for k in (1, 10, 64):
            ra, sa, ca = a.search_batch(q[:nq], k)
            rb, sb, cb = b.search_batch(q[:nq], k)
            assert np.array_equal(ca, cb) and np.array_equal(ra, rb) and np.array_equal(bits(sa), bits(sb)), (nq, k)
    assert b.batched_filter_stats()["int8_queries"] > 0          # it did take the filter path
    allow = rng.random(n) > 0.5
    ra, sa, ca = a.search_batch(q[:4], 10, allow=allow)
    rb, sb, cb = b.search_batch(q[:4], 10, allow=allow)
    assert np.array_equal(ra, rb) and np.array_equal(bits(sa), bits(sb))
    orow, osc = oracle.search_top_k(slab, q[5], 10, live=live)
    rb, sb, cb = b.search_batch(q[5], 10)
    assert np.array_equal(rb[0], orow) and np.array_equal(bits(sb[0]), bits(osc))
    sparse_live = np.zeros(n, bool)
    sparse_live[:7] = True                                          # 7 live rows, k = 10: counts and padding
    c, d = fa.VectorIndex.from_slab(slab, live=sparse_live), fa.VectorIndex.from_slab(slab, live=sparse_live)
    d.set_int8_latency(True)
    rc, sc, cc = c.search_batch(q[:2], 10)
    rd, sd, cd = d.search_batch(q[:2], 10)
    assert cc.tolist() == [7, 7] and np.array_equal(cc, cd) and np.array_equal(rc[:, :7], rd[:, :7]) and np.array_equal(bits(sc[:, :7]), bits(sd[:, :7]))
    for i in (a, b, c, d):
        i.close()


def test_repeated_batches_return_the_same_bits(fa):
    """The same 520-query batch (one 512-query round on the register-resident-query kernels + an 8-query tail round; tombstones;
    k = 30: selections with a few hundred candidates) forty times under each filter: every repetition must return the exact
    kernels' rows and score bits.  (scripts/r03/determinism.py is the long form of this check.)"""
    rng = np.random.default_rng(7)
    dim, n, nq, k = 384, 118_597, 520, 30
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    live = rng.random(n) > 0.2
    q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.2).astype(np.float32)
    idx = fa.VectorIndex.from_slab(x.astype(np.float16).view(np.uint16), live=live)
    exact = [idx.search_batch(q[s:s + 64], k) for s in range(0, nq, 64)]
    er = np.concatenate([e[0] for e in exact])
    es = np.concatenate([e[1] for e in exact])
    for filt in (2, 1):
        idx.set_batched_filter(filt)
        for rep in range(40):
            br, bs, bc, _ = idx.search_batched(q, k)
            assert np.array_equal(br, er), (filt, rep)
            assert np.array_equal(bs.view(np.uint32), es.view(np.uint32)), (filt, rep)
    idx.close()


def test_three_thousand_filtered_repetitions_return_the_same_bits(fa):
    """The r03 failure — one filtered batch in ~16,000 with a wrong tombstone / allow word in the wide kernel's append path
    (profiles/r04/bitmap_soak_noinv.txt: 10 in 161,737 without the per-wave scalar-cache invalidate, 0 in 160,400 with it, 0 with
    vector loads) — guarded in the suite, not only by a builder-run soak: 3,000 repetitions of filtered 520-query batches (dim 384:
    the shape every failing repetition had; tombstones, tombstones + allow bitmap) on the int8 filter, every one compared with the
    exact kernels' rows and score bits.  (At the measured rate this length catches a regression to the unguarded loads with
    probability ~1/6 per run; the soak scripts/r04/bitmap_soak.py is the long form.)"""
    rng = np.random.default_rng(17)
    dim, n, nq, k = 384, 150_011, 520, 30
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    slab = x.astype(np.float16).view(np.uint16)
    live = rng.random(n) > 0.2
    allow = rng.random(n) > 0.3
    q = x[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * 0.2).astype(np.float32)
    idx = fa.VectorIndex.from_slab(slab, live=live)
    idx.set_batched_filter(2)
    for bitmap, reps in ((None, 1500), (allow, 1500)):
        exact = [idx.search_batch(q[s:s + 64], k, allow=bitmap) for s in range(0, nq, 64)]
        er = np.concatenate([e[0] for e in exact])
        es = np.concatenate([e[1] for e in exact]).view(np.uint32)
        for rep in range(reps):
            br, bs, _, _ = idx.search_batched(q, k, allow=bitmap)
            assert np.array_equal(br, er), (bitmap is not None, rep)
            assert np.array_equal(bs.view(np.uint32), es), (bitmap is not None, rep)
    idx.close()


def test_lone_query_certified_single_pass_equals_the_exact_kernels(fa, oracle):
    """The int8 latency path's lone query (fsgpu_index_set_int8_latency): one pass over the int8 copy in which every block keeps its 32
    best integer scores, the finish re-scores the entries within 2 delta of the k-th best from the f16 slab, and the host certifies
    the answer when no block can have dropped a row within that margin (the best of the full lists' last entries lies below the
    threshold) and the candidates fit the finish.  Rows and score bits equal the exact kernels' and the oracle's — on a corpus where
    the certificate holds, with tombstones; on one where thousands of rows lie within the margin (the staged filter path answers,
    same bits); and on one where 40 near-duplicates sit in ONE block's share of the rows (its list of 32 drops eight of them)."""
    rng = np.random.default_rng(23)
    dim, n = 384, 200_003
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    dense = x.copy()
    dense[5000:11000] = dense[5000] + (rng.standard_normal((6000, dim)) * 1e-3).astype(np.float32)   # 6,000 near-duplicates
    one_block = x.copy()   # rows 448..511 are the tiles 28..31: with 256 blocks of 4 waves, all of them block 7's
    one_block[448:488] = one_block[448] + (rng.standard_normal((40, dim)) * 1e-3).astype(np.float32)
    for corpus, probe_row in ((x, 77), (dense, 5003), (one_block, 448)):
        slab = corpus.astype(np.float16).view(np.uint16)
        live = rng.random(n) > 0.1
        a, b = fa.VectorIndex.from_slab(slab, live=live), fa.VectorIndex.from_slab(slab, live=live)
        b.set_int8_latency(True)
        q = corpus[rng.integers(0, n, 12)] + (rng.standard_normal((12, dim)) * 0.1).astype(np.float32)
        q[0] = corpus[probe_row]
        for rep in range(2):   # (the first call builds the int8 copy and its statistics through the staged path)
            for k in (1, 10, 30, 32):
                for i in range(12):
                    ra, sa, ca = a.search_batch(q[i], k)
                    rb, sb, cb = b.search_batch(q[i], k)
                    assert np.array_equal(ca, cb) and np.array_equal(ra, rb) and np.array_equal(bits(sa), bits(sb)), (rep, k, i)
        orow, osc = oracle.search_top_k(slab, q[0], 10, live=live)
        rb, sb, _ = b.search_batch(q[0], 10)
        assert np.array_equal(rb[0], orow) and np.array_equal(bits(sb[0]), bits(osc))
        a.close()
        b.close()


@pytest.mark.parametrize("n", [200_003, 1_000_000])
def test_default_lone_query_takes_the_certified_pass_once_the_int8_copy_exists(fa, oracle, n):
    """Round 5: fsgpu_search_topk answers a lone unfiltered query with the certified int8 pass BY DEFAULT once the index holds the int8
    copy of its slab and its statistics (a batched search builds them) — no opt-in, nothing built for it; before that, and for
    fsgpu_search_topk_exact always, the exact f16 kernels.  Rows and score bits against the ORACLE (not only the exact kernels),
    1M x 384 included (BASELINE config 2's size); tombstones; a query the certificate does not cover (a pile of near-duplicates) is
    answered by the exact kernels with the same bits."""
    rng = np.random.default_rng(29 + n)
    dim, k = 384, 10
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x[5000:11000] = x[5000] + (rng.standard_normal((6000, dim)) * 1e-3).astype(np.float32)   # 6,000 near-duplicates: not certifiable
    slab = x.astype(np.float16).view(np.uint16)
    live = rng.random(n) > 0.1
    idx = fa.VectorIndex.from_slab(slab, live=live)
    q = x[rng.integers(0, n, 40)] + (rng.standard_normal((40, dim)) * 0.1).astype(np.float32)
    q[3] = x[5003]
    assert idx.batched_filter_stats()["int8_queries"] == 0
    first = idx.search_batch(q[0], k)                       # no copy yet: the exact kernels
    assert idx.batched_filter_stats()["int8_queries"] == 0
    idx.search_batched(q, k)                                # builds the int8 copy + statistics
    base = idx.batched_filter_stats()["int8_queries"]
    for i in range(16):
        rows, scores, counts = idx.search_batch(q[i], k)
        er, es, ec = idx.search_batch(q[i], k, exact=True)
        assert np.array_equal(rows, er) and np.array_equal(bits(scores), bits(es)) and np.array_equal(counts, ec), i
        if i < 6:
            orow, osc = oracle.search_top_k(slab, q[i], k, live=live)
            assert np.array_equal(rows[0], orow) and np.array_equal(bits(scores[0]), bits(osc)), i
    assert np.array_equal(first[0], idx.search_batch(q[0], k)[0])
    took = idx.batched_filter_stats()["int8_queries"] - base
    assert took >= 8, took     # most of them were certified (query 3 is not; a failure backs the pass off for a few calls)
    idx.set_batched_filter(1)   # the f16 filter pinned: no int8 pass either
    mid = idx.batched_filter_stats()["int8_queries"]
    idx.search_batch(q[1], k)
    assert idx.batched_filter_stats()["int8_queries"] == mid
    idx.close()


def test_int8_latency_build_now_puts_the_first_lone_query_on_the_certified_pass(fa, oracle):
    """FSGPU_INT8_LATENCY_BUILD_NOW (round 6): the int8 copy and its statistics are built inside fsgpu_index_set_int8_latency, so the
    FIRST lone query already takes the certified int8 pass — its latency does not depend on which searches came before (r05 verdict).
    Same rows and score bits as the oracle and as the exact kernels."""
    rng = np.random.default_rng(61)
    dim, n, k = 384, 200_003, 10
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    slab = x.astype(np.float16).view(np.uint16)
    idx = fa.VectorIndex.from_slab(slab)
    assert not idx.batched_filter_stats()["int8_active"] or idx.batched_filter_stats()["int8_queries"] == 0
    idx.set_int8_latency(True, build_now=True)
    q = x[rng.integers(0, n, 4)] + (rng.standard_normal((4, dim)) * 0.1).astype(np.float32)
    rows, scores, counts = idx.search_batch(q[0], k)            # the very first search of this index
    assert idx.batched_filter_stats()["int8_queries"] == 1      # ... was answered by the int8 pass
    orow, osc = oracle.search_top_k(slab, q[0], k)
    assert np.array_equal(rows[0], orow) and np.array_equal(bits(scores[0]), bits(osc)) and counts[0] == k
    er, es, _ = idx.search_batch(q[0], k, exact=True)
    assert np.array_equal(rows, er) and np.array_equal(bits(scores), bits(es))
    idx.close()


@pytest.mark.parametrize("dim", [128, 256, 384])
def test_group_maxima_sample_stage_gives_the_exact_search_bits(fa, oracle, dim):
    """The int8 filter's append-free sample stage (MfmaScanArgs::stage 3 + select_groups_kernel: every block reports its four best
    groups of 8 rows per query, the best 24 groups' rows are re-scored exactly, tau = max(a_k - 2 delta, S_k x unit - delta)):
    batches of 256 ... 1,030 queries (2-5 query tiles per wave, ragged tails), ranks up to 24 (25: the thresholded stages), heavy
    tombstones (the anchor counts live rows only), a slab whose last sub-tile pair is incomplete and holds the best rows, runs of
    identical rows — rows, score bits and counts equal the exact kernels'; one query per case against the oracle."""
    rng = np.random.default_rng(1000 + dim)
    n = 262_144 + 19                                   # the last pair of the slab is incomplete
    cent = unit_rows(rng, 32, dim)
    rows = cent[rng.integers(0, 32, n)] + 0.3 * rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    rows[70_000:70_040] = rows[69_999]                 # identical rows: the lower row wins
    rows[n - 19:] = rows[123] + 1e-3 * rng.standard_normal((19, dim)).astype(np.float32)   # near-duplicates of a probe in the ragged tail
    slab = rows.astype(np.float16).view(np.uint16)
    for live in (None, rng.random(n) > 0.4):
        idx = fa.VectorIndex.from_slab(slab, live=live)
        idx.set_batched_filter(2)
        for nq, k in ((256, 10), (384, 1), (640, 24), (1030, 10), (530, 25)):
            q = cent[rng.integers(0, 32, nq)] + 0.3 * rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
            q[0] = rows[123]
            q[1] = rows[69_999]
            exact = [idx.search_batch(q[s:s + 64], k) for s in range(0, nq, 64)]
            er, es, ec = (np.concatenate([e[i] for e in exact]) for i in range(3))
            for rep in range(2):
                br, bs, bc, fb = idx.search_batched(q, k)
                assert np.array_equal(bc, ec) and np.array_equal(br, er) and np.array_equal(bits(bs), bits(es)), (dim, nq, k, live is not None, rep)
                assert fb <= nq // 8, (nq, k, fb)
            orow, osc = oracle.search_top_k(slab, q[0], k, live=live)
            assert np.array_equal(br[0][:len(orow)], orow) and np.array_equal(bits(bs[0][:len(osc)]), bits(osc))
        idx.close()


@pytest.mark.parametrize("dim", [256, 384])
def test_batches_of_129_to_383_queries_ride_padded_wide_rounds_with_the_exact_bits(fa, oracle, dim):
    """Round 6: a round of the batched search takes 128-query groups, and from 129 queries on the register-resident-query main pass —
    the last group of a launch may be PADDING (129..255 queries = one 256-slot pass, 257..383 = one 384-slot pass; rounds 3-5 answered
    them as 128 + the rest on the LDS-query kernel).  Padding slots must not append, select or count: rows, score bits and counts of
    every real query equal the exact kernels' (and the oracle's for a probe), with and without tombstones, for the exact search
    (both filters) and the int8 two-pass; a 1,024 + 200 batch crosses a round boundary into a padded round."""
    rng = np.random.default_rng(4200 + dim)
    n = 200_003
    cent = unit_rows(rng, 32, dim)
    rows = cent[rng.integers(0, 32, n)] + 0.3 * rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    slab = rows.astype(np.float16).view(np.uint16)
    for live in (None, rng.random(n) > 0.3):
        idx = fa.VectorIndex.from_slab(slab, live=live)
        for nq, k in ((129, 10), (200, 10), (255, 30), (257, 10), (300, 1), (383, 24), (1024 + 200, 10)):
            q = cent[rng.integers(0, 32, nq)] + 0.3 * rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
            q[nq - 1] = rows[123]                       # the last real query sits next to the padding
            exact = [idx.search_batch(q[s:s + 64], k, exact=True) for s in range(0, nq, 64)]
            er, es, ec = (np.concatenate([e[i] for e in exact]) for i in range(3))
            for filt in (2, 1):
                idx.set_batched_filter(filt)
                br, bs, bc, fb = idx.search_batched(q, k)
                assert br.shape[0] == nq
                assert np.array_equal(bc, ec) and np.array_equal(br, er) and np.array_equal(bits(bs), bits(es)), (dim, nq, k, filt, live is not None)
            orow, osc = oracle.search_top_k(slab, q[nq - 1], k, live=live)
            assert np.array_equal(br[nq - 1][:len(orow)], orow) and np.array_equal(bits(bs[nq - 1][:len(osc)]), bits(osc))
            if nq <= 300:
                # the int8 two-pass (search.rs:571-661) per query against the batched form
                tr, ts, tc = idx.search_int8_two_pass_batched(q, k, 3)[:3]
                for i in (0, 127, 128, nq - 1):
                    hits = idx.search_top_k_int8_two_pass(q[i], k, 3)
                    assert tc[i] == len(hits), (dim, nq, i)
                    assert [int(r) for r in tr[i][:len(hits)]] == [h.index for h in hits], (dim, nq, i)
                    assert np.array_equal(bits(ts[i][:len(hits)]), bits(np.array([h.score for h in hits], np.float32))), (dim, nq, i)
        idx.close()
