"""The library's own descending radix sort of 64-bit sortkeys (csrc/sort_general.hip; round 6 — rounds 1-5 called rocPRIM here), the
collect-all / large-k path of search.rs:449-473, against numpy at the tile boundaries of its kernels (64-key chunks, 2,048-key wave
shares, 8,192-key block tiles) and with heavy duplicates in every digit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sort(fa, keys, varying=2**64 - 1):
    from frankensearch_amd import _lib
    from frankensearch_amd.errors import check
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    out = np.empty_like(keys)
    check(_lib.lib().fsgpu_lab_sort_keys_desc(0, keys.ctypes.data_as(C.c_void_p), keys.size, varying, out.ctypes.data_as(C.c_void_p)))
    return out


@pytest.fixture(scope="module")
def fa():
    import frankensearch_amd as fa
    return fa


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 2047, 2048, 2049, 8191, 8192, 8193, 16384 + 77, 1_000_003])
def test_radix_sort_matches_numpy_at_tile_boundaries(fa, n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    assert np.array_equal(_sort(fa, keys), np.sort(keys)[::-1])


def test_radix_sort_with_duplicates_constant_digits_and_extremes(fa):
    rng = np.random.default_rng(5)
    n = 300_000
    few = rng.integers(0, 2**64, size=7, dtype=np.uint64)
    keys = few[rng.integers(0, 7, size=n)]                       # seven distinct values: every pass moves long equal runs
    assert np.array_equal(_sort(fa, keys), np.sort(keys)[::-1])
    keys = (rng.integers(0, 2**20, size=n, dtype=np.uint64) << np.uint64(22))   # most digits constant
    keys[::1000] = np.uint64(2**64 - 1)
    keys[1::1000] = 0
    assert np.array_equal(_sort(fa, keys), np.sort(keys)[::-1])
    keys = np.arange(n, dtype=np.uint64)                         # already ascending -> fully reversed
    assert np.array_equal(_sort(fa, keys), keys[::-1])
    assert _sort(fa, np.zeros(0, np.uint64)).size == 0


def test_sortkeys_of_scores_and_rows_sort_like_the_reference_order(fa):
    """Sortkey = (order-preserving score bits << 32) | ~row: descending key order = score descending, row ascending on ties
    (search.rs:1704-1720's comparator) — the property the collect-all path relies on."""
    rng = np.random.default_rng(9)
    n = 200_000
    scores = rng.standard_normal(n).astype(np.float32)
    scores[rng.integers(0, n, 5000)] = np.float32(0.25)          # ties
    b = scores.view(np.uint32).astype(np.uint64)
    ordered = np.where(b >> np.uint64(31), ~b & np.uint64(0xffffffff), b | np.uint64(0x80000000))
    rows = np.arange(n, dtype=np.uint64)
    keys = (ordered << np.uint64(32)) | (~rows & np.uint64(0xffffffff))
    got = _sort(fa, keys)
    got_rows = (~got & np.uint64(0xffffffff)).astype(np.int64)
    want = np.lexsort((rows.astype(np.int64), -scores.astype(np.float64)))
    assert np.array_equal(got_rows, want)


@pytest.mark.parametrize("varying", [0xffffffff00ffffff, 0x00ffff0000000fff, 0xff, 0xff00000000000000, 0])
def test_digits_without_a_varying_bit_are_skipped_and_the_order_is_unchanged(fa, varying):
    """The callers' hint (the high byte of the row half is constant on slabs below 16.7M rows): odd and even numbers of remaining
    passes both end in the output buffer; no varying bit at all = a copy."""
    rng = np.random.default_rng(varying & 0xffff)
    n = 50_001
    const = np.uint64(0x5a5a5a5a5a5a5a5a) & ~np.uint64(varying)
    keys = (rng.integers(0, 2**64, size=n, dtype=np.uint64) & np.uint64(varying)) | const
    got = _sort(fa, keys, varying)
    if varying == 0:
        assert np.array_equal(got, keys)
    else:
        assert np.array_equal(got, np.sort(keys)[::-1])
