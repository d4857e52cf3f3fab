"""CPU study (numpy) of the int8 filter's margin on the anisotropic corpus: how many rows stay within the proven margin of the k-th
best score under (a) the r03 bound — one corpus-wide scale, max-over-rows norms —, (b) per-dimension scales s_j (rows stored as
x_j s_j, queries as q_j / s_j) with s_j = 1 / max_i |x_ij| ("flat rows") or the geometric compromise s_j = 1 / sqrt(max_i |x_ij|),
(c) per-row error / int8 norms in place of the corpus maxima.  No product code involved; numbers feed DESIGN 3.1f."""
import sys
import numpy as np

rng = np.random.default_rng(1)
N, dim, k = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000, 384, 10
OUT = (3, 57, 101, 160, 222, 287, 313, 380)


def outlier_corpus(n):
    cent = rng.standard_normal((256, dim)).astype(np.float32)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    w = 1.0 / np.arange(1, 257)
    cl = np.searchsorted(np.cumsum(w / w.sum()), rng.random(n)).clip(max=255)
    x = cent[cl] + 0.30 * rng.standard_normal((n, dim)).astype(np.float32)
    scale = np.ones(dim, np.float32)
    scale[list(OUT)] = 10.0
    x *= scale
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float16).astype(np.float32)


def uniform_corpus(n):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float16).astype(np.float32)


def study(name, x, queries):
    print(f"== {name}: {x.shape[0]} rows")
    amax = np.abs(x).max(axis=0)                      # per-dimension max
    for label, s in (("one corpus-wide scale (r03)", np.full(dim, 1.0 / amax.max(), np.float32)),
                     ("per-dimension scale, flat rows", (1.0 / amax).astype(np.float32)),
                     ("per-dimension scale, geometric", (1.0 / np.sqrt(amax * amax.max())).astype(np.float32))):
        xs = x * s * 127.0                            # rows in integer units
        xs *= 1.0 / np.abs(xs).max() * 127.0          # (the geometric form leaves headroom: renormalise to +-127)
        r = np.rint(xs).clip(-127, 127)
        eps = xs - r
        E2_row = np.linalg.norm(eps, axis=1)
        R2_row = np.linalg.norm(r, axis=1)
        R1_row = np.abs(r).sum(axis=1)
        E2, R2, R1 = E2_row.max(), R2_row.max(), R1_row.max()
        row_scale = (s * 127.0) * (127.0 / np.abs(x * s * 127.0).max())   # x_j -> integer units
        tot = {"max": [], "row": []}
        for q in queries:
            qs = q / row_scale                        # so that sum r_j p_j ~ c_q * S
            cq = 127.0 / np.abs(qs).max()
            ps = qs * cq
            p = np.rint(ps)
            eta = ps - p
            H2, P2, P1 = np.linalg.norm(eta), np.linalg.norm(p), np.abs(p).sum()
            idot = r @ p
            S = x @ q
            err = np.abs(idot - S * cq)
            d_max = min(0.5 * P1, E2 * P2) + min(0.5 * R1, H2 * R2) + min(0.25 * dim, E2 * H2)
            d_row = np.minimum(0.5 * P1, E2_row * P2) + np.minimum(0.5 * R1_row, H2 * R2_row) + np.minimum(0.25 * dim, E2_row * H2)
            assert np.all(err <= d_row * 1.0001 + 1e-3), (err.max(), d_row.max())
            kth_exact = np.sort(S)[-k] * cq           # exact anchor: every true top-k row has idot >= S_k c - delta(row)
            n_max = int(np.sum(idot >= kth_exact - d_max))
            n_row = int(np.sum(idot + d_row >= kth_exact))
            tot["max"].append(n_max)
            tot["row"].append(n_row)
            last = (d_max / cq, np.median(d_row) / cq, err.max() / cq, np.sort(S)[-k])
        print(f"  {label:34s}: delta (cosine units) {last[0]:.4f} (median per-row {last[1]:.4f}; largest actual error {last[2]:.5f}; k-th best {last[3]:.3f})"
              f" | rows within the margin of the k-th best, median over {len(queries)} queries: corpus maxima {int(np.median(tot['max']))}, per-row norms {int(np.median(tot['row']))}"
              f" (worst query {max(tot['max'])} / {max(tot['row'])})")


for name, gen in (("uniform unit vectors", uniform_corpus), ("anisotropic / Zipf clusters", outlier_corpus)):
    x = gen(N)
    pick = rng.integers(0, N, 12)
    if name.startswith("uniform"):
        q = rng.standard_normal((12, dim)).astype(np.float32)
    else:
        q = x[pick] + 0.2 / np.sqrt(dim) * rng.standard_normal((12, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    study(name, x, q)
