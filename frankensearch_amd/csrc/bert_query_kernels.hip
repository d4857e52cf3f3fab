// bert_query_kernels.hip — the MiniLM-L6 forward for ONE short input (a query: at most 32 tokens in total, any number of
// texts among them) in 25 launches instead of 44.
//
// Model::embed_forward (crates/frankensearch-rerank/src/native.rs:1142-1236) on a handful of tokens is a chain of
// dependent stages, each a few microseconds of latency and no throughput to speak of: what it costs is the NUMBER of
// stages.  The general path (bert_kernels.hip) runs 7 kernels per layer: QKV GEMM, attention, out-projection GEMM,
// add+LayerNorm, FFN-up GEMM (+GELU), FFN-down GEMM, add+LayerNorm (encoder_layer_raw, native.rs:587-626).  Here a layer
// is 4 kernels, every elementwise / normalisation step riding in the prologue of the GEMM that consumes it:
//
//   K1  bert_q_qkv_attn_kernel   one block per head: [pending add+LN of the previous stage | embedding gather + LN]
//                                -> the head's Q, K, V columns (3 x 32 of the 1152) on the matrix cores -> softmax(QK^T)V in
//                                the same block (32 x 32 scores: two MFMAs a tile) -> ctx[:, head]
//   K2  bert_q_gemm_kernel       out-projection: ctx x Wao^T -> a partial slab (bias and residual are added by the consumer)
//   K3  bert_q_gemm_kernel       [x = LN1(x + slab + bias)] x Wi^T, +bias, GELU -> inter (f16)
//   K4  bert_q_gemm_kernel       FFN-down split 4 ways along K (24 blocks instead of 6 stream the 1.2 MB of weights):
//                                four partial slabs, summed by the consumer's prologue
//   and after the last layer     bert_q_pool_kernel: x = LN2(x + slabs + bias) -> mean over each text's tokens -> L2.
//
// Every block of a consumer recomputes the (tiny) add+LayerNorm of the 32 rows it needs; block 0 also stores the new
// residual stream into the other of two x buffers (the blocks of one launch read the old one).  A GEMM block is 32 rows x
// 64 (or 32) columns x K = 384: each wave requests ALL its weight fragments up front (one memory round trip instead of a
// k-loop of them) and reads the A fragments of the shared 32 x 384 f16 tile from LDS.
// Precision class as in bert_kernels.hip: f16 x f16 -> f32 MFMA linears, f32 residual / LN / softmax statistics; the
// tolerance tests (tests/test_gpu_bert.py: cosine >= 0.999, max-abs <= 2e-3 vs the f32 oracle) cover this path too.
// MiniLM-L6 shape only (hidden 384, FFN 1536, heads of 32); other shapes take the general path.
#include "device_util.hpp"
#include "kernels.hpp"

namespace fsgpu {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int QH = 384;          // hidden
constexpr int QROWS = 32;        // rows (tokens) a launch covers
constexpr int QKS = QH / 32;     // MFMA k-steps of one 384-wide slice
constexpr int QPITCH = QH + 16;  // halves per A-tile row in LDS: 800 bytes, (pitch/16) mod 16 = 2 -> conflict-free ds_read_b128

__device__ __forceinline__ float q_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float q_row16_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false)));
    return v;
}
__device__ __forceinline__ float q_row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
    return v;
}
// exact-form GELU with the Abramowitz-Stegun 7.1.26 erf (native.rs:190-200)
__device__ __forceinline__ float q_gelu(float x) {
    const float z = x * 0.70710678118654752440f;
    const float az = fabsf(z);
    const float t = 1.0f / (1.0f + 0.3275911f * az);
    const float poly = t * (0.2548296f + t * (-0.28449673f + t * (1.4214137f + t * (-1.453152f + t * 1.0614054f))));
    const float erf_abs = 1.0f - poly * __expf(-(z * z));
    return 0.5f * x * (1.0f + copysignf(erf_abs, z));
}

// The 32 rows a consumer needs: LayerNorm(x_in + the producer's NP partial slabs + the producer's bias) (add_ln_raw,
// native.rs:560-578), or for NP = 0 LayerNorm(word + position + token-type embedding) (native.rs:1176-1192); rows >= tokens
// are zero.  A 16-lane group owns a row (6 float4 per lane, statistics by DPP row reductions), a wave 4 rows at a time, 8 in
// all: every load of a group of rows is independent, so the prologue costs two memory round trips, not one per row.
// tile: f16 rows for the MFMA (pitch QPITCH halves) or null; rows_f32: f32 rows (pitch QH) or null (the pooling stage's LDS
// copy / the new residual stream in global memory).
template <int NP>
__device__ __forceinline__ void q_ln_rows(const BertQueryArgs& a, _Float16* tile, float* rows_f32, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    float4 lw[6], lb[6], pb[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        lw[j] = reinterpret_cast<const float4*>(a.lnw)[li + 16 * j];
        lb[j] = reinterpret_cast<const float4*>(a.lnb)[li + 16 * j];
        if (NP > 0) pb[j] = reinterpret_cast<const float4*>(a.prev_bias)[li + 16 * j];
        else pb[j] = reinterpret_cast<const float4*>(a.type0)[li + 16 * j];
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int r = wave * 8 + it * 4 + g;
        const bool live = r < a.tokens;
        const int rc = live ? r : 0;   // dead rows read row 0 (in bounds) and store zeros
        float4 v[6];
        if constexpr (NP == 0) {
            const float4* wr = reinterpret_cast<const float4*>(a.word + (size_t)a.ids[rc] * QH);
            const float4* pr = reinterpret_cast<const float4*>(a.pos + (size_t)a.positions[rc] * QH);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float4 w4 = wr[li + 16 * j], p4 = pr[li + 16 * j];
                v[j] = make_float4((w4.x + p4.x) + pb[j].x, (w4.y + p4.y) + pb[j].y, (w4.z + p4.z) + pb[j].z, (w4.w + p4.w) + pb[j].w);
            }
        } else {
            const float4* xr = reinterpret_cast<const float4*>(a.x_in + (size_t)rc * QH);
            float4 d[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) d[j] = pb[j];
#pragma unroll
            for (int sl = 0; sl < NP; ++sl) {
                const float4* pr = reinterpret_cast<const float4*>(a.parts + ((size_t)sl * QROWS + rc) * QH);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const float4 p4 = pr[li + 16 * j];
                    d[j].x += p4.x; d[j].y += p4.y; d[j].z += p4.z; d[j].w += p4.w;
                }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float4 x4 = xr[li + 16 * j];
                v[j] = make_float4(x4.x + d[j].x, x4.y + d[j].y, x4.z + d[j].z, x4.w + d[j].w);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        const float mean = q_row16_sum(s) / (float)QH;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        const float inv = 1.0f / sqrtf(q_row16_sum(q) / (float)QH + a.eps);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 y = make_float4((v[j].x - mean) * inv * lw[j].x + lb[j].x, (v[j].y - mean) * inv * lw[j].y + lb[j].y,
                                   (v[j].z - mean) * inv * lw[j].z + lb[j].z, (v[j].w - mean) * inv * lw[j].w + lb[j].w);
            if (!live) y = make_float4(0.f, 0.f, 0.f, 0.f);
            const int d0 = (li + 16 * j) * 4;
            if (tile) {
                typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                half4 h;
                h[0] = (_Float16)y.x; h[1] = (_Float16)y.y; h[2] = (_Float16)y.z; h[3] = (_Float16)y.w;
                *reinterpret_cast<half4*>(tile + (size_t)r * QPITCH + d0) = h;
            }
            if (rows_f32 && live) *reinterpret_cast<float4*>(rows_f32 + (size_t)r * QH + d0) = y;
        }
    }
}

// the prologue of a GEMM stage: fills the LDS tile; write_x: this block also stores the new residual stream
__device__ __forceinline__ void q_fill_tile_ln(const BertQueryArgs& a, _Float16* tile, bool write_x, int tid) {
    float* xo = write_x ? a.x_out : nullptr;
    if (a.ids) q_ln_rows<0>(a, tile, xo, tid);
    else if (a.n_parts == 1) q_ln_rows<1>(a, tile, xo, tid);
    else q_ln_rows<4>(a, tile, xo, tid);
}

// A wave's weight fragments for columns c0 + 16 j .., K slice [k0, k0 + 384): all requested at once — before the prologue that
// builds the A tile, so their memory round trip runs underneath it.  W is [N, ldw] f16 row-major (HF layout).
template <int NT, int NW = NT>   // (NW: rows of the caller's register array, of which the first NT are used)
__device__ __forceinline__ void q_load_b(const _Float16* __restrict__ W, int ldw, int k0, int c0, int lane, half8 (&b)[NW][QKS]) {
    const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const half8* bp = reinterpret_cast<const half8*>(W + (size_t)(c0 + 16 * j + fr) * ldw + k0 + fk);
#pragma unroll
        for (int ks = 0; ks < QKS; ++ks) b[j][ks] = bp[ks * 4];
    }
}
// acc[ri][j] += tile rows (16 ri ..) x those columns
template <int NT, int NW = NT>
__device__ __forceinline__ void q_gemm_384(const _Float16* tile, const half8 (&b)[NW][QKS], int lane, f32x4 (&acc)[2][NT]) {
    const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
    for (int ks = 0; ks < QKS; ++ks) {
        half8 af[2];
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) af[ri] = *reinterpret_cast<const half8*>(tile + (size_t)(ri * 16 + fr) * QPITCH + ks * 32 + fk);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) acc[ri][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ri], b[j][ks], acc[ri][j], 0, 0, 0);
    }
}

}  // namespace

namespace {

// LDS of the three stage shapes (the one-launch form carves them out of one block)
struct QSmemAttn {
    __attribute__((aligned(16))) _Float16 tile[QROWS * QPITCH];       // 25 KB
    __attribute__((aligned(16))) _Float16 qs[QROWS][48], ks_[QROWS][48], vt[32][48], ps[QROWS][48];  // 96-byte pitch
    int doc_of[QROWS];
};
struct QSmemGemm {
    __attribute__((aligned(16))) _Float16 tile[QROWS * QPITCH];
};
struct QSmemPool {
    __attribute__((aligned(16))) float xs[QROWS][QH];   // 48 KB: the normalised rows
    float red[4];
};

// K1's weights: part 0 = Q, 1 = K, 2 = V (waves 0..2): columns part * 384 + head * 32 + [0, 32) of the stacked projection
// (native.rs:1500-1540)
__device__ __forceinline__ void q_attn_load(const BertQueryArgs& a, int head, int tid, half8 (&wb)[2][QKS]) {
    const int lane = tid & 63, wave = tid >> 6;
    const int c0 = (wave < 3 ? wave : 0) * QH + head * 32;
    if (wave < 3) q_load_b<2>(a.w, QH, 0, c0, lane, wb);
}

// K1: one block per head.  LN / embedding prologue -> Q, K, V of the head (waves 0..2: 32 columns each) -> attention.
__device__ __forceinline__ void q_attn_compute(const BertQueryArgs& a, QSmemAttn& sm, int head, int tid, const half8 (&wb)[2][QKS]) {
    _Float16* tile = sm.tile;
    auto& qs = sm.qs;
    auto& ks_ = sm.ks_;
    auto& vt = sm.vt;
    auto& ps = sm.ps;
    int* doc_of = sm.doc_of;
    const int lane = tid & 63, wave = tid >> 6;
    const int c0 = (wave < 3 ? wave : 0) * QH + head * 32;
    q_fill_tile_ln(a, tile, head == 0 && a.x_out != nullptr, tid);
    if (tid < QROWS) {
        int d = -1;
        if (tid < a.tokens)
            for (int i = 0; i < a.n_docs; ++i)
                if ((uint32_t)tid >= a.offsets[i] && (uint32_t)tid < a.offsets[i + 1]) d = i;
        doc_of[tid] = d;
    }
    __syncthreads();
    const int fr = lane & 15, crow = (lane >> 4) * 4;
    if (wave < 3) {
        f32x4 acc[2][2] = {};
        q_gemm_384<2>(tile, wb, lane, acc);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 16 + fr;  // dimension inside the head
            const float bv = a.bias[c0 + col];
#pragma unroll
            for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = ri * 16 + crow + r;
                    const _Float16 y = (_Float16)(acc[ri][j][r] + bv);
                    if (wave == 0) qs[row][col] = y;
                    else if (wave == 1) ks_[row][col] = y;
                    else vt[col][row] = y;  // V transposed: the PV product wants keys contiguous per dimension
                }
        }
    }
    __syncthreads();
    if (wave < 2) {
        // query rows 16 wave .. +16 against all 32 key positions: S = Q K^T is one k-step (head dimension 32) per 16 x 16 tile
        const int fk = (lane >> 4) * 8;
        const half8 qf = *reinterpret_cast<const half8*>(&qs[wave * 16 + fr][fk]);
        f32x4 s[2];
#pragma unroll
        for (int cj = 0; cj < 2; ++cj) {
            const half8 kf = *reinterpret_cast<const half8*>(&ks_[cj * 16 + fr][fk]);
            s[cj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf, kf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        // C layout: this lane holds scores of query rows crow + r against key columns cj * 16 + fr.  Softmax per query row
        // over the keys of the same text (fused_attention / fast_softmax_inplace, native.rs:82-163,366-432: the scale
        // 1/sqrt(32) goes inside the exponential).
        const float scale = a.attn_scale;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qrow = wave * 16 + crow + r;
            const int qd = doc_of[qrow];
            float v[2], m = -INFINITY;
#pragma unroll
            for (int cj = 0; cj < 2; ++cj) {
                const bool ok = qd >= 0 && doc_of[cj * 16 + fr] == qd;
                v[cj] = ok ? s[cj][r] : -INFINITY;
                m = fmaxf(m, v[cj]);
            }
            m = q_row16_max(m);
            float e[2], sum = 0.f;
#pragma unroll
            for (int cj = 0; cj < 2; ++cj) {
                e[cj] = v[cj] == -INFINITY ? 0.f : __expf((v[cj] - m) * scale);
                sum += e[cj];
            }
            sum = q_row16_sum(sum);
            const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
            for (int cj = 0; cj < 2; ++cj) ps[qrow][cj * 16 + fr] = (_Float16)(e[cj] * inv);
        }
        wave_lds_fence();  // the wave reads back only the 16 rows of P it wrote itself
        const half8 pf = *reinterpret_cast<const half8*>(&ps[wave * 16 + fr][fk]);
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            const half8 vf = *reinterpret_cast<const half8*>(&vt[dj * 16 + fr][fk]);
            const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf, vf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 16 + crow + r;
                if (row < a.tokens) a.out_h[(size_t)row * QH + head * 32 + dj * 16 + fr] = (_Float16)o[r];
            }
        }
    }
}

}  // namespace

__global__ __launch_bounds__(256) void bert_q_qkv_attn_kernel(BertQueryArgs a) {
    __shared__ QSmemAttn sm;
    half8 wb[2][QKS];
    q_attn_load(a, (int)blockIdx.x, (int)threadIdx.x, wb);
    q_attn_compute(a, sm, (int)blockIdx.x, (int)threadIdx.x, wb);
}

namespace {

// K2 / K3 / K4.  PRO 0: the A tile is rows of a_h (f16, leading dimension lda), K slice blockIdx.y; PRO 1: pending-LN
// prologue.  EPI 0: partial slab blockIdx.y of out_f32 ([slab][32][N], no bias); EPI 1: GELU(acc + bias) -> out_h (f16).
// NT: 16-column tiles per wave (4 waves: 64 NT columns per block).
template <int NT, int NW = NT>
__device__ __forceinline__ void q_gemm_load(const BertQueryArgs& a, int bx, int by, int tid, half8 (&wb)[NW][QKS]) {
    const int lane = tid & 63, wave = tid >> 6;
    q_load_b<NT, NW>(a.w, a.ldw, by * QH, (bx * 4 + wave) * 16 * NT, lane, wb);
}

template <int PRO, int EPI, int NT, int NW = NT>
__device__ __forceinline__ void q_gemm_compute(const BertQueryArgs& a, QSmemGemm& sm, int bx, int by, int tid, const half8 (&wb)[NW][QKS]) {
    _Float16* tile = sm.tile;
    const int lane = tid & 63, wave = tid >> 6;
    const int k0 = by * QH;
    const int c0 = (bx * 4 + wave) * 16 * NT;
    if constexpr (PRO == 1) {
        q_fill_tile_ln(a, tile, bx == 0 && a.x_out != nullptr, tid);
    } else {
        // 32 rows x 384 halves of the slice, 16-byte pieces (48 per row)
        for (int i = tid; i < QROWS * 48; i += 256) {
            const int r = i / 48, p = i - r * 48;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (r < a.tokens) v = *reinterpret_cast<const u32x4*>(a.a_h + (size_t)r * a.lda + k0 + p * 8);
            *reinterpret_cast<u32x4*>(tile + (size_t)r * QPITCH + p * 8) = v;
        }
    }
    __syncthreads();
    f32x4 acc[2][NT] = {};
    q_gemm_384<NT, NW>(tile, wb, lane, acc);
    const int fr = lane & 15, crow = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = c0 + 16 * j + fr;
        const float bv = EPI == 1 ? a.bias[col] : 0.f;
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ri * 16 + crow + r;
                if (row >= a.tokens) continue;
                if constexpr (EPI == 0) a.out_f32[((size_t)by * QROWS + row) * a.n + col] = acc[ri][j][r];
                else a.out_h[(size_t)row * a.n + col] = (_Float16)q_gelu(acc[ri][j][r] + bv);
            }
    }
}

}  // namespace

template <int PRO, int EPI, int NT>
__global__ __launch_bounds__(256) void bert_q_gemm_kernel(BertQueryArgs a) {
    __shared__ QSmemGemm sm;
    half8 wb[NT][QKS];
    q_gemm_load<NT>(a, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, wb);
    q_gemm_compute<PRO, EPI, NT>(a, sm, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, wb);
}

namespace {

// Final stage: x = LN2(x + slabs + bias), mean over each text's tokens, L2 with the zero guard (native.rs:1209-1235;
// fastembed_embedder.rs:416-426).  One block.
__device__ __forceinline__ void q_pool_compute(const BertQueryArgs& a, QSmemPool& sm, int tid, float* __restrict__ out) {
    auto& xs = sm.xs;
    float* red = sm.red;
    const int lane = tid & 63, wave = tid >> 6;
    q_ln_rows<4>(a, nullptr, &xs[0][0], tid);
    __syncthreads();
    for (int doc = 0; doc < a.n_docs; ++doc) {
        const uint32_t t0 = a.offsets[doc], t1 = a.offsets[doc + 1];
        const int n = (int)(t1 - t0);
        float vals[2], sq = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int d = tid + 256 * i;
            float acc = 0.f;
            if (d < QH && n > 0) {
                for (uint32_t t = t0; t < t1; ++t) acc += xs[t][d];
                acc *= 1.0f / (float)n;
            }
            vals[i] = acc;
            sq += acc * acc;
        }
        sq = q_wave_sum(sq);
        __syncthreads();
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        const float norm_sq = (red[0] + red[1]) + (red[2] + red[3]);
        float scale = 0.f;
        if (__builtin_isfinite(norm_sq) && norm_sq > 1.1920929e-7f) scale = 1.0f / sqrtf(norm_sq);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int d = tid + 256 * i;
            if (d < QH) out[(size_t)doc * QH + d] = vals[i] * scale;
        }
    }
}

}  // namespace

__global__ __launch_bounds__(256) void bert_q_pool_kernel(BertQueryArgs a, float* __restrict__ out) {
    __shared__ QSmemPool sm;
    q_pool_compute(a, sm, (int)threadIdx.x, out);
}

// ---- the whole forward in ONE launch (experiments builds only: measured, slower, not shipped) -----------------------------------
// The same 25 stages, run by 24 resident blocks that meet at a grid-wide barrier between stages instead of at a kernel boundary
// (r05 verdict: "one persistent launch for the 6 layers"); a block requests the NEXT stage's weight fragments after it has arrived
// and before it waits, so that memory round trip runs underneath the barrier instead of at the head of the stage.
// Result (profiles/r06/encoder_one_launch_ab.txt, outputs bit-identical to the 25-launch form): 0.223 ms per 3-token query against
// 0.191 ms for the replayed graph of 25 launches (32 tokens: 0.284 against 0.240).  A kernel boundary inside a replayed graph costs
// ~4 us on this stack; a barrier across XCDs needs an agent-scope release (L2 write-back) and acquire (invalidate) in EVERY wave — with
// them in one wave only the output is wrong (3e-3), with none at all (not coherent: a floor, not a design) 0.177 ms.  So even stage
// data that bypassed the caches entirely could buy 7 % — the time is the 25 dependent stages (~7 us each: two or three memory round
// trips), not their boundaries.  FSGPU_BERT_ONE_LAUNCH=1 in a build with -DFSGPU_EXPERIMENTS selects it.
#ifdef FSGPU_EXPERIMENTS
//   * stages[s]: the BertQueryArgs the host would have launched stage s with (device array, built once per workspace layout);
//     the per-call fields (tokens, texts, offsets, ids, positions, the pooled output) travel in the kernel's argument block.
//   * barrier: arrive = agent-scope release fence (L2 write-back: the next stage's readers sit on other XCDs) + one atomic add;
//     wait = thread 0 polls the counter, the block's second s_barrier, an agent-scope acquire fence (invalidate).  The counter only
//     grows: launch i waits for base + (s + 1) x blocks, base = i x 24 x blocks (wrap-safe signed difference).
//   * the 24 blocks are co-resident on any idle-or-busy device (a block needs 49 KB of LDS of a CU's 160 and nothing else waits on
//     them), but a poll loop that never ends would take the box down with it: after ~2^20 polls a block gives up, raises
//     *status (mapped host memory), pushes the counter past every barrier still ahead and runs on without waiting; the host then
//     discards the result, answers through the 25-launch form and stays there.
struct BertQueryOneLaunchArgs {
    const BertQueryArgs* stages;
    const unsigned char* kinds;     // 0 attention (12 blocks), 1 out-projection (6), 2 FFN-up (24), 3 FFN-down (6 x 4), 4 pool (1)
    int n_stages;
    int tokens, n_docs;
    const uint32_t* offsets;
    const int32_t *ids, *positions;
    float* out;
    unsigned int* counter;          // device memory, zeroed once
    unsigned int base;
    unsigned int* status;           // mapped host word: 1 = a barrier timed out
};

namespace {

constexpr int kOneLaunchBlocks = 24;

__device__ __forceinline__ int q_stage_blocks(int kind) { return kind == 0 ? 12 : kind == 1 ? 6 : kind == 4 ? 1 : 24; }

// A pointer read from the stage table is "generic" to the compiler (flat loads: both wait counters, an aperture check per access);
// every one of them points into device memory, and saying so gets the global_load forms the 25-launch kernels use.
template <class T>
__device__ __forceinline__ T* q_global(T* ptr) {
    return (T*)(__attribute__((address_space(1))) T*)ptr;
}

__device__ __forceinline__ BertQueryArgs q_stage_args(const BertQueryOneLaunchArgs& p, int s) {
    BertQueryArgs a = p.stages[s];
    a.tokens = p.tokens;
    a.n_docs = p.n_docs;
    a.offsets = p.offsets;
    if (a.ids) {
        a.ids = p.ids;
        a.positions = p.positions;
    }
    a.word = q_global(a.word);
    a.pos = q_global(a.pos);
    a.type0 = q_global(a.type0);
    a.x_in = q_global(a.x_in);
    a.x_out = q_global(a.x_out);
    a.parts = q_global(a.parts);
    a.prev_bias = q_global(a.prev_bias);
    a.lnw = q_global(a.lnw);
    a.lnb = q_global(a.lnb);
    a.a_h = q_global(a.a_h);
    a.w = q_global(a.w);
    a.bias = q_global(a.bias);
    a.out_f32 = q_global(a.out_f32);
    a.out_h = q_global(a.out_h);
    return a;
}

// (two register sets, one per stage shape: with one array serving both the compiler kept it in scratch)
__device__ __forceinline__ void q_stage_load(const BertQueryArgs& a, int kind, int b, int tid, half8 (&wa)[2][QKS], half8 (&wg)[1][QKS]) {
    if (b >= q_stage_blocks(kind)) return;
    if (kind == 0) q_attn_load(a, b, tid, wa);
    else if (kind == 3) q_gemm_load<1>(a, b % 6, b / 6, tid, wg);
    else if (kind != 4) q_gemm_load<1>(a, b, 0, tid, wg);
}

}  // namespace

__global__ __launch_bounds__(256) void bert_q_one_launch_kernel(BertQueryOneLaunchArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(QSmemPool) > sizeof(QSmemAttn) ? sizeof(QSmemPool) : sizeof(QSmemAttn)];
    __shared__ int gave_up;
    const int tid = threadIdx.x, b = blockIdx.x;
    if (tid == 0) gave_up = 0;
    half8 wa[2][QKS], wg[1][QKS];
    {
        const BertQueryArgs a0 = q_stage_args(p, 0);
        q_stage_load(a0, p.kinds[0], b, tid, wa, wg);
    }
    for (int s = 0; s < p.n_stages; ++s) {
        const int kind = p.kinds[s];
        const BertQueryArgs a = q_stage_args(p, s);
        __syncthreads();   // (the previous stage's LDS is free; gave_up is set)
        if (b < q_stage_blocks(kind)) {
            if (kind == 0) q_attn_compute(a, *reinterpret_cast<QSmemAttn*>(smem), b, tid, wa);
            else if (kind == 1) q_gemm_compute<0, 0, 1>(a, *reinterpret_cast<QSmemGemm*>(smem), b, 0, tid, wg);
            else if (kind == 2) q_gemm_compute<1, 1, 1>(a, *reinterpret_cast<QSmemGemm*>(smem), b, 0, tid, wg);
            else if (kind == 3) q_gemm_compute<0, 0, 1>(a, *reinterpret_cast<QSmemGemm*>(smem), b % 6, b / 6, tid, wg);
            else q_pool_compute(a, *reinterpret_cast<QSmemPool*>(smem), tid, p.out);
        }
        if (s + 1 == p.n_stages) break;
        // ---- arrive
#ifndef FSGPU_Q1L_FENCE
#define FSGPU_Q1L_FENCE 2   // 2: every wave fences; 1: thread 0's wave only (WRONG output: measured); 0: none (a timing floor, not coherent)
#endif
        if (FSGPU_Q1L_FENCE == 2 || (FSGPU_Q1L_FENCE == 1 && tid < 64)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        if (tid == 0) (void)__hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- the next stage's weights: in flight while the block waits
        const BertQueryArgs an = q_stage_args(p, s + 1);
        q_stage_load(an, p.kinds[s + 1], b, tid, wa, wg);
        // ---- wait
        if (tid == 0 && !gave_up) {
            const unsigned int target = p.base + (unsigned int)(s + 1) * (unsigned int)kOneLaunchBlocks;
            unsigned int polls = 0;
            while ((int)(__hip_atomic_load(p.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > (1u << 20)) {
                    gave_up = 1;
                    __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    // (and lets every other block through all the barriers still ahead: the host resets the counter)
                    (void)__hip_atomic_fetch_add(p.counter, 0x40000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __builtin_amdgcn_s_barrier();
        if (FSGPU_Q1L_FENCE == 2 || (FSGPU_Q1L_FENCE == 1 && tid < 64)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

#endif  // FSGPU_EXPERIMENTS

// ---- launchers ----------------------------------------------------------------------------------------------------

bool bert_query_path_supported(int hidden, int inter, int heads) { return hidden == QH && inter == 4 * QH && heads * 32 == QH; }

hipError_t launch_bert_q_qkv_attn(const BertQueryArgs& a, int heads, hipStream_t stream) {
    hipLaunchKernelGGL(bert_q_qkv_attn_kernel, dim3(heads), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// mode 0: out-projection (A = a_h, one partial slab, N = 384: 6 blocks of 64 columns)
// mode 1: FFN-up (LN prologue, GELU epilogue, N = 1536: 24 blocks)
// mode 2: FFN-down (A = a_h sliced 4 ways along K = 1536, four partial slabs, N = 384: 6 x 4 blocks)
hipError_t launch_bert_q_gemm(const BertQueryArgs& a, int mode, hipStream_t stream) {
    if (mode == 0) hipLaunchKernelGGL((bert_q_gemm_kernel<0, 0, 1>), dim3(a.n / 64, 1), dim3(256), 0, stream, a);
    else if (mode == 1) hipLaunchKernelGGL((bert_q_gemm_kernel<1, 1, 1>), dim3(a.n / 64, 1), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((bert_q_gemm_kernel<0, 0, 1>), dim3(a.n / 64, 4), dim3(256), 0, stream, a);
    return hipGetLastError();
}

#ifdef FSGPU_EXPERIMENTS
int bert_q_one_launch_blocks() { return kOneLaunchBlocks; }

hipError_t launch_bert_q_one_launch(const BertQueryArgs* stages_dev, const unsigned char* kinds_dev, int n_stages, int tokens, int n_docs,
                                    const uint32_t* offsets, const int32_t* ids, const int32_t* positions, float* out,
                                    unsigned int* counter_dev, unsigned int base, unsigned int* status_mapped, hipStream_t stream) {
    BertQueryOneLaunchArgs p{};
    p.stages = stages_dev;
    p.kinds = kinds_dev;
    p.n_stages = n_stages;
    p.tokens = tokens;
    p.n_docs = n_docs;
    p.offsets = offsets;
    p.ids = ids;
    p.positions = positions;
    p.out = out;
    p.counter = counter_dev;
    p.base = base;
    p.status = status_mapped;
    hipLaunchKernelGGL(bert_q_one_launch_kernel, dim3(kOneLaunchBlocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}
#else
int bert_q_one_launch_blocks() { return 0; }
hipError_t launch_bert_q_one_launch(const BertQueryArgs*, const unsigned char*, int, int, int, const uint32_t*, const int32_t*, const int32_t*,
                                    float*, unsigned int*, unsigned int, unsigned int*, hipStream_t) {
    return hipErrorNotSupported;
}
#endif

hipError_t launch_bert_q_pool(const BertQueryArgs& a, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(bert_q_pool_kernel, dim3(1), dim3(256), 0, stream, a, out);
    return hipGetLastError();
}

}  // namespace fsgpu
