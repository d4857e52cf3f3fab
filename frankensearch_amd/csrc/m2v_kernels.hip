// m2v_kernels.hip — potion / Model2Vec static embedder: row gather -> sum -> mean -> L2 (gfx950).
//
// Replaces Model2VecEmbedder::embed_token_ids for a batch of token-id sequences
// (crates/frankensearch-embed/src/model2vec_embedder.rs:310-335,409-419,435-451;
//  crates/frankensearch-embed/src/simd.rs:273-289).  Bit-exact by construction: each output
// dimension is summed in token order by one lane (ids >= vocab skipped), scaled by 1/count, the
// squared norm is accumulated left-to-right over dimensions with separate multiply and add, and
// the vector is scaled by 1/sqrt(norm) only when norm^2 is finite and > f32::EPSILON.
// HBM-bound gather: T tokens x dim x 4 bytes per text, coalesced across the dim axis.
#pragma clang fp contract(off)

#include "kernels.hpp"

namespace fsgpu {

__global__ __launch_bounds__(256) void m2v_embed_kernel(const float* __restrict__ table, uint32_t vocab, uint32_t dim,
                                                        const uint32_t* __restrict__ ids,
                                                        const uint32_t* __restrict__ offsets, float* __restrict__ out) {
    extern __shared__ float mean[];  // [dim]
    __shared__ float s_scale;
    const uint32_t text = blockIdx.x;
    const uint32_t begin = offsets[text], end = offsets[text + 1];
    const int tid = threadIdx.x;
    uint32_t count = 0;
    for (uint32_t t = begin; t < end; ++t) count += ids[t] < vocab ? 1u : 0u;  // uniform, L1/L2-resident
    float* o = out + (size_t)text * dim;
    if (count == 0) {
        for (uint32_t d = tid; d < dim; d += 256) o[d] = 0.f;
        return;
    }
    const float inv = 1.0f / (float)count;
    for (uint32_t d = tid; d < dim; d += 256) {
        float s = 0.f;
        for (uint32_t t = begin; t < end; ++t) {
            const uint32_t id = ids[t];
            if (id < vocab) s = s + table[(size_t)id * dim + d];
        }
        mean[d] = s * inv;
    }
    __syncthreads();
    if (tid == 0) {
        float norm_sq = 0.f;
        for (uint32_t d = 0; d < dim; ++d) {
            const float p = mean[d] * mean[d];
            norm_sq = norm_sq + p;
        }
        float scale = 0.f;
        if (__builtin_isfinite(norm_sq) && norm_sq > 1.1920929e-7f) scale = 1.0f / __builtin_sqrtf(norm_sq);
        s_scale = scale;
    }
    __syncthreads();
    const float scale = s_scale;
    for (uint32_t d = tid; d < dim; d += 256) o[d] = scale != 0.f ? mean[d] * scale : 0.f;
}

hipError_t launch_m2v_embed(const float* table, uint32_t vocab, uint32_t dim, const uint32_t* ids,
                            const uint32_t* offsets, uint32_t n, float* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(m2v_embed_kernel, dim3(n), dim3(256), (size_t)dim * 4, stream, table, vocab, dim, ids, offsets,
                       out);
    return hipGetLastError();
}

}  // namespace fsgpu
