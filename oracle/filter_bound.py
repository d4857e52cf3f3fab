"""CPU restatement of the int8 filter's certificate (test infrastructure, like the rest of oracle/).

The batched exact search may score the slab on an int8 copy (quantize_f16_le_bytes_to_i8_generic, simd.rs:1865-1886: ONE
corpus-wide scale) against int8 queries (quantize_i8_query, search.rs:1616-1626) and then re-score, in the reference's order,
every row whose integer score could belong to the top k.  "Could" rests on a bound delta_q with

    |idot(row, q) - S(row, q) * c_s * c_q| <= delta_q        for every row,

S = the real-number dot of the f16 row with the f32 query, c_s = fl(127 / max|x|), c_q = fl(127 / max|q|).  This module computes
that bound with numpy exactly as frankensearch_amd/csrc/mfma_scan.hip (prepare_queries_i8_filter_kernel) and int8_kernels.hip
(i8_slab_stats_kernel) document it; tests/test_oracle_filter_bound.py checks it against float64 arithmetic on the CPU, and the
GPU suite checks that the kernels' delta agrees with this one.
"""
from __future__ import annotations

import numpy as np

from . import oracle

F32 = np.float32


def slab_stats(slab_u16: np.ndarray):
    """(c_s, E2, R1, R2, finite): the scale the quantiser used and the measured maxima over rows of |eps|_2 (eps = x c_s - r,
    each |eps_i| taken with + 8e-6 for the rounding of the product), |r|_1, |r|_2."""
    x = np.ascontiguousarray(slab_u16).view(np.float16).astype(F32)
    finite = bool(np.all(np.isfinite(x)))
    max_abs = F32(np.nanmax(np.abs(x))) if x.size else F32(0)
    r = oracle.quantize_slab_i8(slab_u16).astype(np.int64)
    if not (max_abs > 0) or not finite:
        return F32(0), 0.0, 0.0, 0.0, False
    c_s = F32(127.0) / max_abs
    eps = np.abs((x * c_s).astype(F32) - r.astype(F32)).astype(np.float64) + 8e-6
    e2 = float(np.sqrt((eps * eps).sum(axis=1).max())) * 1.001
    r1 = float(np.abs(r).sum(axis=1).max())
    r2 = float(np.sqrt((r * r).sum(axis=1).max()))
    return c_s, e2, r1, r2, True


def query_bound(q: np.ndarray, stats, dim: int):
    """(delta, c_q, p): the bound for one f32 query (delta < 0: not certifiable), its scale and its int8 image."""
    c_s, e2, r1, r2, ok = stats
    q = np.ascontiguousarray(q, dtype=F32)
    p = oracle.quantize_query_i8(q).astype(np.int64)
    if not ok or not np.all(np.abs(q) <= 65504.0) or dim > 1040:   # (NaN fails the comparison)
        return -1.0, F32(0), p
    max_abs = F32(np.max(np.abs(q)))
    if not (max_abs > 0):
        return -1.0, F32(0), p
    with np.errstate(over="ignore", invalid="ignore"):
        c_q = F32(127.0) / max_abs
        scaled = (q * c_q).astype(F32)
    if not np.isfinite(c_q) or not np.all(np.isfinite(scaled)):
        return -1.0, c_q, p
    eta = np.abs((q * c_q).astype(F32) - p.astype(F32)).astype(np.float64) + 8e-6
    h2 = float(np.sqrt((eta * eta).sum())) * 1.001
    p1 = float(np.abs(p).sum())
    p2 = float(np.sqrt((p * p).sum()))
    n = float(dim)
    d = min(0.50001 * p1, e2 * p2) + min(0.50001 * r1, h2 * r2) + min(0.25001 * n, e2 * h2)
    d += n * 2.0 ** -23 * (r2 + e2) * (p2 + h2) + n * 1.5e-45 * (float(c_s) * 1.000001) * (float(c_q) * 1.000001)
    d = d * 1.001 + 1.0
    if not np.isfinite(d) or not d < 1e9:
        return -1.0, c_q, p
    return d, c_q, p
