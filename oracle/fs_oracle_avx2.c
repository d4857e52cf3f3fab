/*
 * fs_oracle_avx2.c — AVX2+F16C form of the oracle's f16·f32 dot (TEST INFRASTRUCTURE ONLY).
 * Same arithmetic, same order as fso_dot_f16_f32 and therefore bit-identical to it; mirrors the
 * reference's runtime-dispatched kernel crates/frankensearch-index/src/simd.rs:398-446
 * (vcvtph2ps decode, separate vmulps + vaddps into four accumulators, (s0+s1)+(s2+s3), the
 * `wide::f32x8::reduce_add` horizontal step, fused scalar tail).  Used as the CPU baseline in
 * bench.py because it runs at the reference's own speed class.
 * Compiled alone with -mavx2 -mf16c -mfma (fmaf for the tail) -ffp-contract=off.
 */
#include "fs_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <string.h>

float fso_dot_f16_f32_avx2_impl(const uint8_t *row, const float *q, size_t dim, int hreduce) {
    size_t chunks = dim / 8;
    __m256 s0 = _mm256_setzero_ps(), s1 = s0, s2 = s0, s3 = s0;
    size_t c = 0;
#define PROD(ci) _mm256_mul_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(row + (ci) * 16))), _mm256_loadu_ps(q + (ci) * 8))
    while (c + 4 <= chunks) {
        s0 = _mm256_add_ps(s0, PROD(c));
        s1 = _mm256_add_ps(s1, PROD(c + 1));
        s2 = _mm256_add_ps(s2, PROD(c + 2));
        s3 = _mm256_add_ps(s3, PROD(c + 3));
        c += 4;
    }
    while (c < chunks) {
        s0 = _mm256_add_ps(s0, PROD(c));
        c += 1;
    }
#undef PROD
    __m256 sum = _mm256_add_ps(_mm256_add_ps(s0, s1), _mm256_add_ps(s2, s3));
    float v[8];
    _mm256_storeu_ps(v, sum);
    float result;
    if (hreduce == FSO_HREDUCE_SEQ) {
        float a = ((v[0] + v[1]) + v[2]) + v[3];
        float b = ((v[4] + v[5]) + v[6]) + v[7];
        result = a + b;
    } else if (hreduce == FSO_HREDUCE_AVX) {
        float a = v[0] + v[4], b = v[1] + v[5], cc = v[2] + v[6], d = v[3] + v[7];
        float lo = a + cc, hi = b + d;
        result = lo + hi;
    } else {
        float a = (v[0] + v[2]) + (v[1] + v[3]);
        float b = (v[4] + v[6]) + (v[5] + v[7]);
        result = a + b;
    }
    for (size_t i = chunks * 8; i < dim; ++i) {
        uint16_t h = (uint16_t)(row[2 * i] | ((uint16_t)row[2 * i + 1] << 8));
        float w = _cvtsh_ss(h);
        result = fmaf(w, q[i], result);
    }
    return result;
}
