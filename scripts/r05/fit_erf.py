"""The erf of the encoder's GELU epilogues (bert_gemm_w.hip gelu_as_w): erf(z) = 1 - 2^q(z) on z >= 0, q a polynomial without
constant term fitted to log2(erfc) on [0, 4] by iteratively reweighted least squares (weights -> the absolute error of erf).  Prints
the coefficients and the largest error against erf and against the reference's f32 evaluation of Abramowitz-Stegun 7.1.26
(native.rs:190-200), both evaluated in f32, and of the GELU built on it."""
import numpy as np
from scipy.special import erf, erfc


def as_erf(z):
    z = z.astype(np.float32)
    az = np.abs(z)
    t = (np.float32(1.0) / (np.float32(1.0) + np.float32(0.3275911) * az)).astype(np.float32)
    poly = t * (np.float32(0.254829592) + t * (np.float32(-0.284496736) + t * (np.float32(1.421413741) + t * (np.float32(-1.453152027) + t * np.float32(1.061405429)))))
    return np.copysign((np.float32(1.0) - poly * np.exp(-(z * z))).astype(np.float32), z)


def fit(deg, zmax=4.0):
    zs = np.linspace(0, zmax, 200001)
    target = np.log2(np.maximum(erfc(zs), 1e-300))
    A = np.stack([zs ** (i + 1) for i in range(deg)], axis=1)
    w = erfc(zs) + 1e-9
    for _ in range(30):
        coef, *_ = np.linalg.lstsq(A * w[:, None], target * w, rcond=None)
        err = np.abs((1 - 2.0 ** (A @ coef)) - erf(zs))
        w = w * (1 + 2 * err / err.max())
    return coef


def gelu32(x, c):
    x = x.astype(np.float32)
    az = (np.abs(x) * np.float32(0.70710678118654752440)).astype(np.float32)
    p = np.full_like(az, np.float32(c[-1]))
    for k in c[-2::-1]:
        p = (p * az + np.float32(k)).astype(np.float32)
    e_half = np.exp2((p * az - np.float32(1.0)).astype(np.float32)).astype(np.float32)
    return (np.maximum(x, np.float32(0)) - np.abs(x) * e_half).astype(np.float32)


if __name__ == "__main__":
    c = fit(5)
    print("coefficients a1..a5:", ", ".join("%.8f" % v for v in c))
    xs = np.linspace(-12, 12, 2000001)
    ref = (0.5 * xs * (1 + erf(xs / np.sqrt(2))))
    ref_as = (np.float32(0.5) * xs.astype(np.float32) * (np.float32(1) + as_erf((xs / np.sqrt(2)).astype(np.float32))))
    got = gelu32(xs, c)
    print("GELU: max |error| vs exact %.3e, vs the reference's 7.1.26 form in f32 %.3e (f16 rounding of a value near 1: 4.9e-4)"
          % (np.abs(got - ref).max(), np.abs(got - ref_as).max()))
    print("largest x with q > 0:", xs[xs > 0][np.argmax(gelu32(xs[xs > 0], c) > xs[xs > 0] + 1e-6)] if np.any(gelu32(xs[xs > 0], c) > xs[xs > 0] + 1e-6) else "none")
