#!/usr/bin/env python3
"""What a PROVEN bound costs a 6-bit floating filter (fp6 E2M3, the operand format of v_mfma_scale_f32_16x16x128_f8f6f4) next to the int8
filter, on the bench corpus (CPU, numpy; rows of the reference's bench recipe, rotated by a random orthogonal map as the filter copy of
an outlier corpus is — after the rotation the components are near-Gaussian, the fp6 code's best case).

For both codes: the corpus-wide scale, the measured maxima E2 = max_row |eps|_2 and R2 = max_row |r|_2, per query H2 = |eta|_2 and P2, the
Cauchy-Schwarz bound delta = E2 P2 + H2 R2 + E2 H2 (the int8 filter's proven form, oracle/filter_bound.py, without the min() with the Hoelder
terms, which only matter for spiky queries), converted to cosine units; then, per query, how many rows lie within ONE delta below the k-th
best exact score (what the main pass appends with an exact anchor) and within TWO (what the finish re-scores without one)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle  # noqa: E402

n, dim, k = int(os.environ.get("ROWS", 1_000_000)), 384, int(os.environ.get("K", 10))
x = oracle.clustered_corpus_f16(0, n, dim).view(np.float16).astype(np.float32)
rng = np.random.default_rng(1)
rot, _ = np.linalg.qr(rng.standard_normal((dim, dim)))
xr = (x @ rot.astype(np.float32)).astype(np.float32)
queries = np.stack([oracle.clustered_query(i, dim) for i in range(16)]).astype(np.float32)
qr = (queries @ rot.astype(np.float32)).astype(np.float32)

# fp6 E2M3 magnitudes: subnormals 0 .. 0.875 (step 1/8), then [1, 2) step 1/8, [2, 4) step 1/4, [4, 7.5] step 1/2
grid = np.unique(np.concatenate([np.arange(0, 8) / 8.0, 1 + np.arange(0, 8) / 8.0, 2 + np.arange(0, 8) / 4.0, 4 + np.arange(0, 8) / 2.0])).astype(np.float32)


def q_fp6(v, scale):
    s = np.abs(v) * scale
    idx = np.clip(np.searchsorted(grid, s), 1, grid.size - 1)
    lo, hi = grid[idx - 1], grid[idx]
    r = np.where(s - lo <= hi - s, lo, hi)
    return np.sign(v) * np.minimum(r, grid[-1])


def q_i8(v, scale):
    return np.clip(np.rint(v * scale), -127, 127)


def q_i4(v, scale):   # the reference's signed nibbles (simd.rs:2153-2215: scale 7 / max_abs)
    return np.clip(np.rint(v * scale), -7, 7)


for name, quant, top in (("int8", q_i8, 127.0), ("fp6 E2M3", q_fp6, 7.5), ("int4", q_i4, 7.0)):
    c_s = np.float32(top) / np.abs(xr).max()
    r = quant(xr, c_s)
    eps = xr * c_s - r
    e2 = float(np.sqrt((eps.astype(np.float64) ** 2).sum(axis=1).max()))
    r2 = float(np.sqrt((r.astype(np.float64) ** 2).sum(axis=1).max()))
    one, two, deltas = [], [], []
    for qi in range(queries.shape[0]):
        c_q = np.float32(top) / np.abs(qr[qi]).max()
        p = quant(qr[qi], c_q)
        eta = qr[qi] * c_q - p
        h2 = float(np.sqrt((eta.astype(np.float64) ** 2).sum()))
        p2 = float(np.sqrt((p.astype(np.float64) ** 2).sum()))
        delta = (e2 * p2 + h2 * r2 + e2 * h2) / (float(c_s) * float(c_q))   # cosine units
        exact = x @ queries[qi]
        kth = np.partition(exact, n - k)[n - k]
        one.append(int((exact >= kth - delta).sum()))
        two.append(int((exact >= kth - 2 * delta).sum()))
        deltas.append(delta)
        approx = (r @ p) / (float(c_s) * float(c_q))
        assert np.max(np.abs(approx - xr @ qr[qi])) <= delta, "the bound must hold"
    print(f"{name:9s} rows {n}: delta (cosine units) mean {np.mean(deltas):.4f} max {np.max(deltas):.4f} | rows within ONE delta of the k-th best: "
          f"median {int(np.median(one))} max {max(one)} | within TWO: median {int(np.median(two))} max {max(two)}   (x {10_000_000 // n} at 10M rows)")
