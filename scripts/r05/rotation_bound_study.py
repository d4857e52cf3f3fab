"""CPU study (numpy, float64 checks): the int8 filter's proven margin delta, in COSINE units, with and without a random orthogonal
rotation of slab and queries before quantisation (dot products are invariant; outlier channels are not).  Feeds DESIGN 3.1f (round 5).
usage: rotation_bound_study.py [rows]"""
import sys
import numpy as np

sys.path.insert(0, ".")
rng = np.random.default_rng(1)
N, dim, k = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000, 384, 10
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
    return x.astype(np.float16).astype(np.float64)


def clustered_corpus(n):   # the bench's friendly corpus, roughly: 64 centroids + 0.30 noise
    cent = rng.uniform(-1, 1, (64, dim))
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    x = cent[np.arange(n) % 64] + 0.30 * rng.uniform(-1, 1, (n, dim))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float16).astype(np.float64)


def bound(x, queries):
    """delta per query in cosine units + rows within 2 delta of the k-th best + the largest actual error (cosine units)."""
    cs = 127.0 / np.abs(x).max()
    xs = x * cs
    r = np.rint(xs).clip(-127, 127)
    eps = xs - r
    E2, R2, R1 = np.linalg.norm(eps, axis=1).max(), np.linalg.norm(r, axis=1).max(), np.abs(r).sum(axis=1).max()
    out = []
    for q in queries:
        cq = 127.0 / np.abs(q).max()
        ps = q * cq
        p = np.rint(ps)
        eta = ps - p
        H2, P2, P1 = np.linalg.norm(eta), np.linalg.norm(p), np.abs(p).sum()
        d = min(0.5 * P1, E2 * P2) + min(0.5 * R1, H2 * R2) + min(0.25 * dim, E2 * H2)
        idot = r @ p
        S = x @ q
        err = np.abs(idot - S * cs * cq).max()
        assert err <= d * 1.0001 + 1e-6
        kth = np.sort(idot)[-k]
        within = int(np.sum(idot >= kth - 2 * d))
        out.append((d / (cs * cq), within, err / (cs * cq)))
    return np.array(out)


Q, _ = np.linalg.qr(rng.standard_normal((dim, dim)))
for name, x in (("outlier", outlier_corpus(N)), ("clustered", clustered_corpus(N))):
    pick = rng.integers(0, N, 24)
    qs = x[pick] + 0.2 / dim ** 0.5 * rng.standard_normal((24, dim))
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    qs = qs.astype(np.float32).astype(np.float64)
    a = bound(x, qs)
    b = bound(x @ Q.T, qs @ Q.T)
    print(f"== {name}: {N} rows, max|x| {np.abs(x).max():.3f} -> rotated {np.abs(x @ Q.T).max():.3f}")
    for label, t in (("as stored   ", a), ("rotated     ", b)):
        print(f"   {label} delta (cosine) median {np.median(t[:, 0]):.4f} max {t[:, 0].max():.4f} | rows within 2 delta of the k-th best: median "
              f"{int(np.median(t[:, 1]))} max {int(t[:, 1].max())} | largest actual error {t[:, 2].max():.4f}")
