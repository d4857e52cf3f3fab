// bench_fixture.hip — the reference's own bench corpus, generated where it is consumed (in HBM).
//
// frankensearch/benches/fsvi_4bit_vs_incumbent.rs:56-101,344-365: raw_vector(seed) is a xorshift64 stream
// (state = seed | 1; per dimension state ^= state << 13; ^= state >> 7; ^= state << 17; v = (state >> 40) / 2^23 - 1);
// 64 centroids normalize(raw_vector(0xc0000000 + i)); row i = normalize(centroid[i % 64] + 0.30 * raw_vector(i + 1));
// query q = the same with seed 0xdead0000 + q; rows are then stored as f16 (RNE), queries stay f32.
// Every f32 operation is issued in the reference's order (sequential norm, separate multiply and add, IEEE divide and
// sqrt), so the bytes equal the CPU generator's — tests compare a prefix with the oracle's restatement.
// A bench/test fixture, not a search path: one thread per row, the stream is regenerated for the second sweep.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace fsgpu {

namespace {

__device__ __forceinline__ float xorshift_next(uint64_t& state) {
    state ^= state << 13;
    state ^= state >> 7;
    state ^= state << 17;
    return (float)(state >> 40) / 8388608.0f - 1.0f;
}

// centroid c -> cent[c][dim] (f32, normalised)
__global__ void fixture_centroids_kernel(uint32_t clusters, uint32_t dim, float* __restrict__ cent) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= clusters) return;
    float* out = cent + (size_t)c * dim;
    uint64_t state = (0xc0000000ull + c) | 1ull;
    float acc = 0.0f;
    for (uint32_t d = 0; d < dim; ++d) {
        const float v = xorshift_next(state);
        out[d] = v;
        const float p = v * v;
        acc = acc + p;
    }
    const float norm = sqrtf(acc);
    if (norm > 1e-12f)
        for (uint32_t d = 0; d < dim; ++d) out[d] = out[d] / norm;
}

// make_vector(centroids, index % clusters, seed_base + index): f16 rows (out_f16) or f32 queries (out_f32)
__global__ void fixture_vectors_kernel(const float* __restrict__ cent, uint64_t first, uint64_t n, uint32_t dim,
                                       uint32_t clusters, float noise, uint64_t seed_base, unsigned short* __restrict__ out_f16,
                                       float* __restrict__ out_f32) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t index = first + i;
    const float* c = cent + (size_t)(index % clusters) * dim;
    const uint64_t seed = (seed_base + index) | 1ull;
    uint64_t state = seed;
    float acc = 0.0f;
    for (uint32_t d = 0; d < dim; ++d) {
        const float p = noise * xorshift_next(state);
        const float v = c[d] + p;
        const float sq = v * v;
        acc = acc + sq;
    }
    const float norm = sqrtf(acc);
    const bool scale = norm > 1e-12f;
    state = seed;
    if (out_f16 && dim % 8 == 0) {  // 16-byte stores: eight halves at a time
        uint4* dst = reinterpret_cast<uint4*>(out_f16 + (size_t)i * dim);
        for (uint32_t d = 0; d < dim; d += 8) {
            uint32_t w[4];
            for (int h = 0; h < 4; ++h) {
                unsigned short lo_hi[2];
                for (int e = 0; e < 2; ++e) {
                    const float p = noise * xorshift_next(state);
                    float v = c[d + 2 * h + e] + p;
                    if (scale) v = v / norm;
                    lo_hi[e] = __builtin_bit_cast(unsigned short, (_Float16)v);
                }
                w[h] = (uint32_t)lo_hi[0] | ((uint32_t)lo_hi[1] << 16);
            }
            dst[d / 8] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return;
    }
    for (uint32_t d = 0; d < dim; ++d) {
        const float p = noise * xorshift_next(state);
        float v = c[d] + p;
        if (scale) v = v / norm;
        if (out_f16) out_f16[(size_t)i * dim + d] = __builtin_bit_cast(unsigned short, (_Float16)v);
        else out_f32[(size_t)i * dim + d] = v;
    }
}

}  // namespace

hipError_t launch_bench_fixture(uint64_t first, uint64_t n, uint32_t dim, uint32_t clusters, float noise, uint64_t seed_base,
                                float* centroid_scratch, unsigned short* out_f16, float* out_f32, hipStream_t stream) {
    hipLaunchKernelGGL(fixture_centroids_kernel, dim3((clusters + 63) / 64), dim3(64), 0, stream, clusters, dim, centroid_scratch);
    if (n)
        hipLaunchKernelGGL(fixture_vectors_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, centroid_scratch,
                           first, n, dim, clusters, noise, seed_base, out_f16, out_f32);
    return hipGetLastError();
}

}  // namespace fsgpu
