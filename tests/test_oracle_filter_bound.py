"""The int8 filter's certificate on the CPU (oracle/filter_bound.py): the bound it states must cover, for every row and query,
the distance between the integer score and the real-number score in integer-score units — checked with float64 on data chosen
to stress each term (outlier dimensions, tiny and large magnitudes, sparse and one-hot rows, peaky and worst-case-rounding
queries).  The GPU suite checks the kernels' delta against this restatement (tests/test_gpu_int8_filter.py)."""
import numpy as np
import pytest


def _unit(rng, n, dim):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def _corpora(rng, n, dim):
    base = _unit(rng, n, dim)
    yield "gaussian", base
    out = base.copy()
    out[:, rng.integers(0, dim, 3)] *= 12.0
    yield "outlier dimensions", out / np.linalg.norm(out, axis=1, keepdims=True)
    yield "tiny", base * 3e-3
    yield "large", base * 180.0
    yield "sparse", (base * (rng.random((n, dim)) < 0.1)).astype(np.float32)
    one = np.zeros((n, dim), np.float32)
    one[np.arange(n), rng.integers(0, dim, n)] = rng.choice([-1.0, 1.0], n)
    one[::3] = base[::3]
    yield "one-hot mixed", one


def _queries(rng, rows, dim):
    q = rows[rng.integers(0, rows.shape[0], 12)] + (0.2 * rng.standard_normal((12, dim))).astype(np.float32)
    q[1] *= 37.5
    q[2] *= 1e-6
    q[3] = 0.0
    q[3, 7] = 1.0
    q[4] = np.sign(q[4]) * 0.25
    q[5] = np.sign(q[5]) * ((rng.integers(0, 126, dim) + 0.5) / 127.0).astype(np.float32)
    q[5, 0] = 1.0
    q[6, :] = 0.003
    q[6, 0] = 1.0
    return q


@pytest.mark.parametrize("dim", [64, 384])
def test_bound_covers_float64_scores(oracle, dim):
    from oracle import filter_bound as fb

    rng = np.random.default_rng(31 + dim)
    n = 3000
    tightest = 0.0
    for name, rows in _corpora(rng, n, dim):
        slab = rows.astype(np.float16).view(np.uint16)
        stats = fb.slab_stats(slab)
        assert stats[4], name
        x64 = slab.view(np.float16).astype(np.float64)
        r = oracle.quantize_slab_i8(slab).astype(np.int64)
        for qi, q in enumerate(_queries(rng, rows, dim)):
            delta, c_q, p = fb.query_bound(q, stats, dim)
            assert delta > 0, (name, qi)
            idot = r @ p
            s = x64 @ q.astype(np.float64)
            err = np.abs(idot - s * float(stats[0]) * float(c_q))
            assert err.max() <= delta, (name, qi, float(err.max()), delta)
            tightest = max(tightest, float(err.max()) / delta)
    assert tightest > 0.05, tightest       # not vacuous


def test_uncertifiable_inputs_are_marked(oracle):
    from oracle import filter_bound as fb

    rng = np.random.default_rng(3)
    rows = _unit(rng, 500, 64)
    slab = rows.astype(np.float16).view(np.uint16)
    stats = fb.slab_stats(slab)
    q = rows[0].copy()
    assert fb.query_bound(q, stats, 64)[0] > 0
    for bad in (np.zeros(64, np.float32), np.where(np.arange(64) == 3, np.nan, q), np.where(np.arange(64) == 3, np.inf, q),
                np.where(np.arange(64) == 3, 70000.0, q), q * np.float32(1e-38)):
        assert fb.query_bound(bad.astype(np.float32), stats, 64)[0] < 0
    for poison in (0x7e00, 0x7c00, 0xfc00):
        s2 = slab.copy()
        s2[17, 5] = poison
        assert not fb.slab_stats(s2)[4]
    assert not fb.slab_stats(np.zeros((10, 64), np.uint16))[4]
    assert fb.query_bound(q, stats, 2048)[0] < 0            # integer scores no longer convert to f32 exactly
