// bert_gemm_w.hip — the batch-path linears of the MiniLM-class encoder (encoder_layer_raw, native.rs:587-626: QKV, attention
// output + add_ln_raw, FFN up + GELU, FFN down + add_ln_raw) with the weights PRE-PACKED in matrix-core fragment order.
//
// Why: at 5k tokens these GEMMs are 1.5-6 GFLOP each with K = 384 — a block's whole k-loop is a dozen steps, and the
// LDS-tiled kernels of bert_kernels.hip (one barrier and one global round trip per 32-wide k-step, prefetch depth 1) spend
// their time waiting: 250-275 TFLOP/s.  Here nothing in the k-loop waits on memory:
//   * the weights never change, so at model load they are re-laid so that the B fragment of one (16-column tile, k-step) —
//     lane l holds W[16 t + (l & 15)][32 ks + 8 (l >> 4) .. + 8] — is one contiguous 1 KB run.  A wave fetches its fragments
//     with perfectly coalesced 16-byte loads straight into registers: no LDS staging, no barrier, no 16-rows-per-load
//     address pattern.  K = hidden: a wave holds its whole weight slice (24 fragments for K = 384) before the first MFMA;
//     K = inter (FFN down): the fragments stream through a register ring six k-steps deep, private to the wave;
//   * the activation tile of a block (BM rows x the WHOLE K) is fetched once, with every load issued up front, parked in
//     LDS at a conflict-free pitch, and one barrier later each wave runs its k-loop on LDS reads and MFMAs alone.
// The MFMA operands are swapped (D = W_frag x A_frag^T) so a lane ends up with FOUR CONSECUTIVE output columns of one row:
// bias, residual and LayerNorm weights are float4 loads, the stores are 16 (f32) / 8 (f16) bytes per lane.
//
// Same arithmetic as bert_kernels.hip (f16 x f16 -> f32 v_mfma_f32_16x16x32_f16, f32 bias / GELU / residual / LayerNorm);
// tolerance against the f32 oracle asserted in tests/test_gpu_bert.py.
#include <algorithm>

#include "device_util.hpp"
#include "kernels.hpp"

namespace fsgpu {

#ifndef FSGPU_GEMM_WP_MIN_TILES
#define FSGPU_GEMM_WP_MIN_TILES 96   // 64-row tiles (6,144 rows) from which a K = hidden GEMM runs weight-stationary
#endif
#ifndef FSGPU_GEMM_WQ_MIN_TILES
#define FSGPU_GEMM_WQ_MIN_TILES 96   // 64-row tiles (6,144 rows) from which a K = hidden GEMM runs on the 32 x 64-per-wave weight-stationary kernel
#endif
#ifndef FSGPU_LN_RING
#define FSGPU_LN_RING 12
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

namespace {

// GELU(x) = x (1 + erf(x / sqrt 2)) / 2 (native.rs:190-200, where erf is Abramowitz-Stegun 7.1.26).  Here erf(|z|) = 1 - 2^q(|z|)
// with q a degree-5 polynomial without constant term fitted to log2(erfc) on [0, 4] (scripts/r05/fit_erf.py: |error| <= 7e-7 against
// erf, 1.1e-6 against the reference's f32 evaluation of 7.1.26 — the result is rounded to f16, 5e-4, two lines later; q(z) < 0 and
// falls to -inf for all z > 0, so no clamp).  One transcendental and nine full-rate instructions per value:
//   GELU(x) = max(x, 0) - |x| 2^(q(|z|) - 1)
// where the 7.1.26 form takes a reciprocal, an exponential and thirteen more — the GELUs of the FFN-up epilogue were 30 % of the
// post-attention block's life at 16k tokens (profiles/r05/ffn_stamps.txt).
__device__ __forceinline__ float gelu_as_w(float x) {
    const float az = fabsf(x) * 0.70710678118654752440f;
    float p = fmaf(az, -0.00294418f, 0.02959011f);
    p = fmaf(p, az, -0.14866571f);
    p = fmaf(p, az, -0.91850934f);
    p = fmaf(p, az, -1.62788901f);
    const float e_half = __builtin_amdgcn_exp2f(fmaf(p, az, -1.0f));   // (1 - erf(|z|)) / 2
    return fmaxf(x, 0.0f) - fabsf(x) * e_half;
}

}  // namespace

// [N, K] f16 row-major -> fragment order: piece ((t * K/32 + ks) * 64 + lane) = W[16 t + (lane & 15)][32 ks + 8 (lane >> 4) ..+8]
__global__ void bert_pack_w_kernel(const _Float16* __restrict__ src, _Float16* __restrict__ dst, int N, int K) {
    const size_t piece = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ksteps = K / 32;
    if (piece >= (size_t)(N / 16) * ksteps * 64) return;
    const int lane = (int)(piece & 63);
    const size_t tk = piece >> 6;
    const int ks = (int)(tk % ksteps), t = (int)(tk / ksteps);
    const half8 v = *reinterpret_cast<const half8*>(src + (size_t)(t * 16 + (lane & 15)) * K + ks * 32 + (lane >> 4) * 8);
    reinterpret_cast<half8*>(dst)[piece] = v;
}

// C[M, N] = A[M, K] W^T + bias for K = 32 KS = hidden.  Block = 64 rows x 128 columns, wave w = 64 rows x columns
// [32 w, 32 w + 32).  EPI 0: f32 output; EPI 1: GELU, f16 output; EPI 2: f16 output.
template <int EPI, int KS>
__global__ __launch_bounds__(256) void bert_gemm_w_kernel(const _Float16* __restrict__ A, const half8* __restrict__ Wp,
                                                          const float* __restrict__ bias, float* __restrict__ out_f32,
                                                          _Float16* __restrict__ out_h, int M, int N) {
    constexpr int K = 32 * KS, BM = 64;
    constexpr int PITCH = K + 16;                      // halves: (2 K + 32) / 16 = K / 8 + 2 slots, = 2 mod 16 for K % 128 == 0
    constexpr int PIECES = K / 8;                      // 16-byte pieces per row
    constexpr int A_LOADS = BM * PIECES / 256;
    static_assert(BM * PIECES % 256 == 0, "tile must divide among 256 loader threads");
    __shared__ __attribute__((aligned(16))) _Float16 As[BM * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm0 = blockIdx.y * BM, bn0 = blockIdx.x * 128 + wave * 32;
    // this wave's whole weight slice: 2 column tiles x KS k-steps, each fragment a contiguous 1 KB run
    half8 wf[2][KS];
    {
        const half8* wp = Wp + (size_t)(bn0 / 16) * KS * 64 + lane;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[j][ks] = wp[(j * KS + ks) * 64];
    }
    // the block's activation tile: rows bm0 .. bm0 + 63 are one contiguous run of A (clamped at M)
    {
        half8 ra[A_LOADS];
#pragma unroll
        for (int x = 0; x < A_LOADS; ++x) {
            const int p = tid + 256 * x, r = p / PIECES, c = p % PIECES;
            int row = bm0 + r;
            row = row < M ? row : M - 1;
            ra[x] = *(reinterpret_cast<const half8*>(A + (size_t)row * K) + c);
        }
#pragma unroll
        for (int x = 0; x < A_LOADS; ++x) {
            const int p = tid + 256 * x, r = p / PIECES, c = p % PIECES;
            *reinterpret_cast<half8*>(&As[r * PITCH + c * 8]) = ra[x];
        }
    }
    __syncthreads();
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        half8 af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const half8*>(&As[(i * 16 + fr) * PITCH + ks * 32 + fk]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j][ks], af[i], acc[i][j], 0, 0, 0);
    }
    // D = W_frag x A_frag^T: lane holds output row (lane & 15) of the m-tile, columns 4 (lane >> 4) .. + 3 of the n-tile
    const int cq = (lane >> 4) * 4;
    if (EPI == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = bn0 + j * 16 + cq;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = bm0 + i * 16 + fr;
                if (row < M) *reinterpret_cast<f32x4*>(out_f32 + (size_t)row * N + col) = acc[i][j] + bv;
            }
        }
        return;
    }
    // f16 outputs leave through LDS (the activation tile's space): a lane's 8 bytes of 16 different rows per store
    // instruction made the stores half of the kernel; transposed, 16 lanes write 256 contiguous bytes of a row
    constexpr int CP = 128 + 8;  // halves per row of the output tile (272 bytes)
    __syncthreads();             // every wave is done reading the activation tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + bn0 + j * 16 + cq);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 y = acc[i][j] + bv;
            half4 h;
            h[0] = (_Float16)(EPI == 1 ? gelu_as_w(y[0]) : y[0]);
            h[1] = (_Float16)(EPI == 1 ? gelu_as_w(y[1]) : y[1]);
            h[2] = (_Float16)(EPI == 1 ? gelu_as_w(y[2]) : y[2]);
            h[3] = (_Float16)(EPI == 1 ? gelu_as_w(y[3]) : y[3]);
            *reinterpret_cast<half4*>(&As[(i * 16 + fr) * CP + wave * 32 + j * 16 + cq]) = h;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 16 + (tid >> 4), c = (tid & 15) * 8;
        const int row = bm0 + r;
        if (row < M)
            *reinterpret_cast<half8*>(out_h + (size_t)row * N + blockIdx.x * 128 + c) = *reinterpret_cast<const half8*>(&As[r * CP + c]);
    }
}

// The same GEMM for LARGE M (thousands of rows: the documents of an index build), WEIGHT-STATIONARY: bert_gemm_w_kernel re-reads a
// wave's 24 KB weight slice for every 64-row tile — 2,304 blocks x 96 KB at M = 16,384, N = 1,152: the L2 -> CU weight stream and the
// per-block prologue (weights, activation tile, barrier) are what that kernel's 29 us are made of.  Here a block keeps its 128-column
// weight slice in registers for its whole life and walks the row tiles t = blockIdx.y, + gridDim.y, ...: the next tile's activations
// are requested into registers before the current tile's MFMAs, outputs leave through their own LDS staging tile, two barriers per
// tile.  Same arithmetic and epilogues (bit-identical outputs: the per-element operation order is unchanged).
template <int EPI, int KS>
__global__ __launch_bounds__(256, 2) void bert_gemm_wp_kernel(const _Float16* __restrict__ A, const half8* __restrict__ Wp,
                                                           const float* __restrict__ bias, float* __restrict__ out_f32,
                                                           _Float16* __restrict__ out_h, int M, int N) {
    constexpr int K = 32 * KS, BM = 64;
    constexpr int PITCH = K + 16;
    constexpr int PIECES = K / 8;
    constexpr int A_LOADS = BM * PIECES / 256;
    constexpr int CP = 128 + 8;   // halves per row of the f16 output tile
    static_assert(BM * PIECES % 256 == 0, "tile must divide among 256 loader threads");
    __shared__ __attribute__((aligned(16))) _Float16 As[BM * PITCH];
    __shared__ __attribute__((aligned(16))) _Float16 Os[EPI == 0 ? 8 : BM * CP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bn0 = blockIdx.x * 128 + wave * 32;
    const int tiles = (M + BM - 1) / BM;
    half8 wf[2][KS];
    {
        const half8* wp = Wp + (size_t)(bn0 / 16) * KS * 64 + lane;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[j][ks] = wp[(j * KS + ks) * 64];
    }
    const int fr = lane & 15, fk = (lane >> 4) * 8, cq = (lane >> 4) * 4;
    f32x4 bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bias + bn0 + j * 16 + cq);
    half8 ra[A_LOADS];
    auto fetch = [&](int t) {   // tile t's rows (clamped at M) into registers: every load in flight at once
#pragma unroll
        for (int x = 0; x < A_LOADS; ++x) {
            const int p = tid + 256 * x, r = p / PIECES, c = p % PIECES;
            int row = t * BM + r;
            row = row < M ? row : M - 1;
            ra[x] = *(reinterpret_cast<const half8*>(A + (size_t)row * K) + c);
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int x = 0; x < A_LOADS; ++x) {
            const int p = tid + 256 * x, r = p / PIECES, c = p % PIECES;
            *reinterpret_cast<half8*>(&As[r * PITCH + c * 8]) = ra[x];
        }
    };
    int t = blockIdx.y;
    if (t >= tiles) return;
    fetch(t);
    park();
    __syncthreads();
    for (; t < tiles; t += gridDim.y) {
        const int tn = t + (int)gridDim.y;
        if (tn < tiles) fetch(tn);   // the next tile's activations travel under this tile's MFMAs
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const half8*>(&As[(i * 16 + fr) * PITCH + ks * 32 + fk]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j][ks], af[i], acc[i][j], 0, 0, 0);
        }
        const int bm0 = t * BM;
        if (EPI == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = bn0 + j * 16 + cq;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = bm0 + i * 16 + fr;
                    if (row < M) *reinterpret_cast<f32x4*>(out_f32 + (size_t)row * N + col) = acc[i][j] + bv[j];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 y = acc[i][j] + bv[j];
                    half4 h;
                    h[0] = (_Float16)(EPI == 1 ? gelu_as_w(y[0]) : y[0]);
                    h[1] = (_Float16)(EPI == 1 ? gelu_as_w(y[1]) : y[1]);
                    h[2] = (_Float16)(EPI == 1 ? gelu_as_w(y[2]) : y[2]);
                    h[3] = (_Float16)(EPI == 1 ? gelu_as_w(y[3]) : y[3]);
                    *reinterpret_cast<half4*>(&Os[(i * 16 + fr) * CP + wave * 32 + j * 16 + cq]) = h;
                }
        }
        __syncthreads();   // every wave is done reading As (and the output tile is complete)
        if (tn < tiles) park();
        if (EPI != 0) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 16 + (tid >> 4), c = (tid & 15) * 8;
                const int row = bm0 + r;
                if (row < M)
                    *reinterpret_cast<half8*>(out_h + (size_t)row * N + blockIdx.x * 128 + c) = *reinterpret_cast<const half8*>(&Os[r * CP + c]);
            }
        }
        __syncthreads();   // the next tile is parked; the output tile may be overwritten
    }
}

// Round 5: what the launcher picks for thousands of rows.  (A 128 x 128-tile form with one 512-register wave per SIMD and a 64 x 64
// wave tile was built first and measured the same 30 us as bert_gemm_wp_kernel: one block per CU serialises a tile's phases — fetch,
// park, matrix work, epilogue, stores.  What all forms had in common was the placement of their blocks: see the mapping below.)  A
// block is 64 rows x 128 columns, wave (wr, wc) = rows [32 wr, +32) x columns [64 wc, +64): the same 192-register weight slice
// (4 column tiles x KS k-steps), every activation fragment feeds FOUR matrix instructions (bert_gemm_wp_kernel: two — its four
// waves read each row tile four times over, as much LDS time as matrix time), accumulators 32 registers, no staging registers for a
// prefetch: ~250 registers -> two blocks per CU, and what hides one block's loads, barriers and stores is the OTHER block's matrix
// work.  Same per-element arithmetic (bit-identical outputs).
template <int EPI, int KS, int CT>
__global__ __launch_bounds__(256, 2) void bert_gemm_wq_kernel(const _Float16* __restrict__ A, const half8* __restrict__ Wp,
                                                              const float* __restrict__ bias, float* __restrict__ out_f32,
                                                              _Float16* __restrict__ out_h, int M, int N, int qcols, int walkers) {
    constexpr int K = 32 * KS, BM = 64;
    constexpr int PITCH = K + 16;
    constexpr int PIECES = K / 8;
    constexpr int A_LOADS = BM * PIECES / 256;
    constexpr int BN = 32 * CT;   // columns per block: two waves side by side, CT column tiles of 16 each
    constexpr int CP = BN + 8;    // halves per row of the f16 output tile
    static_assert(BM * PIECES % 256 == 0, "tile must divide among 256 loader threads");
    __shared__ __attribute__((aligned(16))) _Float16 As[BM * PITCH];
    __shared__ __attribute__((aligned(16))) _Float16 Os[EPI == 0 ? 8 : BM * CP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware mapping (block b runs on XCD b % 8; every XCD has its own 4 MB L2): the qcols column blocks that read the SAME row
    // tile sit on ONE XCD and walk the XCD's tiles in step — XCD c owns the row tiles t = c (mod 8), its blocks are (column block,
    // walker) pairs, walker w of `walkers` takes every walkers-th of those tiles.  With the 2-D grid order every row tile was fetched
    // by up to eight L2s, and at different times by the column blocks of one L2: TCC hit rate 44 %, 101 MB fetched from the fabric
    // for 13.5 MB of operands (profiles/r05/encoder_large_m_pmc.txt) — that traffic, not the matrix pipe or LDS, was the kernel's time.
    const int xcd = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3);
    const int col_block = slot % qcols, walker = slot / qcols;
    const int bn0 = col_block * BN + wc * 16 * CT;
    const int tiles = (M + BM - 1) / BM;
    const int stride = 8 * walkers;
    int t = xcd + 8 * walker;
    if (t >= tiles) return;
    half8 wf[CT][KS];
    {
        const half8* wp = Wp + (size_t)(bn0 / 16) * KS * 64 + lane;
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[j][ks] = wp[(j * KS + ks) * 64];
    }
    const int fr = lane & 15, fk = (lane >> 4) * 8, cq = (lane >> 4) * 4;
    const _Float16* a_base = As + (wr * 32 + fr) * PITCH + fk;
    for (; t < tiles; t += stride) {
        {   // the tile's rows (clamped at M), parked at the conflict-free pitch — four loads in flight per lane at a time (16 registers:
            // the weights hold 192 of the wave's 256; the other block of this CU computes meanwhile)
            constexpr int CH = 4;
            static_assert(A_LOADS % CH == 0, "the tile's loads split into groups of four");
#pragma unroll
            for (int x0 = 0; x0 < A_LOADS; x0 += CH) {
                half8 ra[CH];
#pragma unroll
                for (int x = 0; x < CH; ++x) {
                    const int p = tid + 256 * (x0 + x), r = p / PIECES, c = p % PIECES;
                    int row = t * BM + r;
                    row = row < M ? row : M - 1;
                    ra[x] = *(reinterpret_cast<const half8*>(A + (size_t)row * K) + c);
                }
#pragma unroll
                for (int x = 0; x < CH; ++x) {
                    const int p = tid + 256 * (x0 + x), r = p / PIECES, c = p % PIECES;
                    *reinterpret_cast<half8*>(&As[r * PITCH + c * 8]) = ra[x];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        f32x4 acc[2][CT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(a_base + i * 16 * PITCH + ks * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j][ks], af[i], acc[i][j], 0, 0, 0);
        }
        const int bm0 = t * BM;
        if (EPI == 0) {
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int col = bn0 + j * 16 + cq;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = bm0 + wr * 32 + i * 16 + fr;
                    if (row < M) *reinterpret_cast<f32x4*>(out_f32 + (size_t)row * N + col) = acc[i][j] + bv;
                }
            }
            __syncthreads();   // every wave is done reading As: the next tile may be parked
            continue;
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + bn0 + j * 16 + cq);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 y = acc[i][j] + bv;
                half4 h;
                h[0] = (_Float16)(EPI == 1 ? gelu_as_w(y[0]) : y[0]);
                h[1] = (_Float16)(EPI == 1 ? gelu_as_w(y[1]) : y[1]);
                h[2] = (_Float16)(EPI == 1 ? gelu_as_w(y[2]) : y[2]);
                h[3] = (_Float16)(EPI == 1 ? gelu_as_w(y[3]) : y[3]);
                *reinterpret_cast<half4*>(&Os[(wr * 32 + i * 16 + fr) * CP + wc * 16 * CT + j * 16 + cq]) = h;
                __builtin_amdgcn_sched_barrier(0);   // (one output fragment at a time: the GELU chains must not pile up next to 192 weight registers)
            }
        }
        __syncthreads();   // the output tile is complete, and every wave is done reading As
        constexpr int OP = BN / 8;   // 16-byte pieces per output row
        for (int p = tid; p < BM * OP; p += 256) {
            const int r = p / OP, c = (p % OP) * 8;
            const int row = bm0 + r;
            if (row < M)
                *reinterpret_cast<half8*>(out_h + (size_t)row * N + col_block * BN + c) = *reinterpret_cast<const half8*>(&Os[r * CP + c]);
        }
        // (the next iteration's park is behind these reads of Os only through its own barrier: Os and As are different arrays)
    }
}

// x = LayerNorm(x + A W^T + bias) (add_ln_raw, native.rs:560-578) for N = hidden = 64 CT, any K % 32 == 0.  A block owns 32
// complete rows; wave w owns columns [16 CT w, 16 CT (w + 1)).  The 32 x K activation tile sits in LDS for the whole kernel;
// the wave's weight fragments stream through a register ring RING k-steps deep (no barrier in the k-loop).  Row statistics:
// in-lane over the lane's columns, two xor-shuffles over the four column quads, one LDS exchange over the four waves; two
// passes (mean, then centred variance) like bert_add_ln_kernel.
template <int CT, int RING>
__global__ __launch_bounds__(256) void bert_gemm_ln_w_kernel(const _Float16* __restrict__ A, const half8* __restrict__ Wp,
                                                             const float* __restrict__ bias, float* __restrict__ x_f32,
                                                             _Float16* __restrict__ x_h, const float* __restrict__ lnw,
                                                             const float* __restrict__ lnb, int M, int K, float eps) {
    constexpr int H = 64 * CT, BM = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char glw_smem[];
    float* red = reinterpret_cast<float*>(glw_smem);                          // [BM][4]
    _Float16* As = reinterpret_cast<_Float16*>(glw_smem + BM * 4 * 4);        // [BM][K + 16]
    constexpr int XP = H + 4;                                                 // floats per row of the residual / output tile
    float* Xs = reinterpret_cast<float*>(glw_smem + BM * 4 * 4 + (size_t)BM * (K + 16) * 2);   // [BM][XP]
    const int pitch = K + 16, pieces = K / 8, ksteps = K / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm0 = blockIdx.x * BM;
    const half8* wp = Wp + (size_t)(wave * CT) * ksteps * 64 + lane;          // fragment (j, ks) at wp[(j * ksteps + ks) * 64]
    half8 wr[RING][CT];
#pragma unroll
    for (int d = 0; d < RING; ++d)
        if (d < ksteps)
#pragma unroll
            for (int j = 0; j < CT; ++j) wr[d][j] = wp[((size_t)j * ksteps + d) * 64];
    // activation tile: 32 rows x K, every load in flight before the first LDS write (up to 24 pieces per thread: K = 1536 in one round)
    for (int p0 = 0; p0 < BM * pieces; p0 += 256 * 24) {
        half8 ra[24];
#pragma unroll
        for (int x = 0; x < 24; ++x) {
            const int p = p0 + tid + 256 * x;
            if (p < BM * pieces) {
                const int r = p / pieces, c = p - r * pieces;
                int row = bm0 + r;
                row = row < M ? row : M - 1;
                ra[x] = *(reinterpret_cast<const half8*>(A + (size_t)row * K) + c);
            }
        }
#pragma unroll
        for (int x = 0; x < 24; ++x) {
            const int p = p0 + tid + 256 * x;
            if (p < BM * pieces) {
                const int r = p / pieces, c = p - r * pieces;
                *reinterpret_cast<half8*>(&As[r * pitch + c * 8]) = ra[x];
            }
        }
    }
    // the residual rows, fetched row-contiguous like everything else; the epilogue reads them (and returns the new rows)
    // through LDS, so no global access of this kernel touches 16 rows per instruction
    {
        constexpr int XL = BM * (H / 4) / 256;
        f32x4 rx[XL];
#pragma unroll
        for (int x = 0; x < XL; ++x) {
            const int p = tid + 256 * x, r = p / (H / 4), c4 = p % (H / 4);
            int row = bm0 + r;
            row = row < M ? row : M - 1;
            rx[x] = *(reinterpret_cast<const f32x4*>(x_f32 + (size_t)row * H) + c4);
        }
#pragma unroll
        for (int x = 0; x < XL; ++x) {
            const int p = tid + 256 * x, r = p / (H / 4), c4 = p % (H / 4);
            *reinterpret_cast<f32x4*>(&Xs[r * XP + c4 * 4]) = rx[x];
        }
    }
    __syncthreads();
    f32x4 acc[2][CT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    for (int ks0 = 0; ks0 < ksteps; ks0 += RING) {
#pragma unroll
        for (int d = 0; d < RING; ++d) {
            const int ks = ks0 + d;
            if (ks < ksteps) {
                half8 af[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(&As[(i * 16 + fr) * pitch + ks * 32 + fk]);
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[d][j], af[i], acc[i][j], 0, 0, 0);
                if (ks + RING < ksteps)
#pragma unroll
                    for (int j = 0; j < CT; ++j) wr[d][j] = wp[((size_t)j * ksteps + ks + RING) * 64];
            }
        }
    }
    // epilogue.  Lane: rows i * 16 + fr (i = 0, 1), columns wave * 16 CT + j * 16 + cq .. + 3
    const int cq = (lane >> 4) * 4;
    float psum[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int col = wave * 16 * CT + j * 16 + cq;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + col);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(&Xs[(i * 16 + fr) * XP + col]);
            const f32x4 y = (acc[i][j] + bv) + xv;
            acc[i][j] = y;
            s += (y[0] + y[1]) + (y[2] + y[3]);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        psum[i] = s;
    }
    if (lane < 16) {
        red[(0 * 16 + lane) * 4 + wave] = psum[0];
        red[(1 * 16 + lane) * 4 + wave] = psum[1];
    }
    __syncthreads();
    float mean[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* p = red + (i * 16 + fr) * 4;
        mean[i] = ((p[0] + p[1]) + (p[2] + p[3])) / (float)H;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const f32x4 d = acc[i][j] - mean[i];
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        if (lane < 16) red[(i * 16 + lane) * 4 + wave] = q;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* p = red + (i * 16 + fr) * 4;
        const float var = ((p[0] + p[1]) + (p[2] + p[3])) / (float)H;
        const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int col = wave * 16 * CT + j * 16 + cq;
            const f32x4 g = *reinterpret_cast<const f32x4*>(lnw + col);
            const f32x4 b = *reinterpret_cast<const f32x4*>(lnb + col);
            *reinterpret_cast<f32x4*>(&Xs[(i * 16 + fr) * XP + col]) = (acc[i][j] - mean[i]) * inv * g + b;   // own element
        }
    }
    __syncthreads();
    constexpr int XL = BM * (H / 4) / 256;
#pragma unroll
    for (int x = 0; x < XL; ++x) {
        const int pp = tid + 256 * x, r = pp / (H / 4), c4 = pp % (H / 4);
        const int row = bm0 + r;
        if (row >= M) continue;
        const f32x4 y = *reinterpret_cast<const f32x4*>(&Xs[r * XP + c4 * 4]);
        *(reinterpret_cast<f32x4*>(x_f32 + (size_t)row * H) + c4) = y;
        half4 h;
        h[0] = (_Float16)y[0];
        h[1] = (_Float16)y[1];
        h[2] = (_Float16)y[2];
        h[3] = (_Float16)y[3];
        *(reinterpret_cast<half4*>(x_h + (size_t)row * H) + c4) = h;
    }
}

// The whole feed-forward block of a layer in one launch: x = LayerNorm(x + GELU(x W1^T + b1) W2^T + b2)
// (encoder_layer_raw, native.rs:606-626) for hidden = 64 CT.  A 512-thread block owns 32 complete rows:
//   phase 0  the f16 rows of x -> LDS (the up-projection's A operand);
//   phase 1  wave w computes intermediate columns [inter/8 w, inter/8 (w + 1)) CH 16-column tiles at a time, W1 fragments
//            streaming through a register ring one whole chunk deep (both projections are bound by the latency of the
//            weight stream — ~1.3 us under load —, so what counts is bytes in flight: 8 waves x KS1 x CH KB);
//            bias + GELU -> f16 -> the 32 x inter tile in LDS.  The intermediate activations (16 MB per layer at 5k
//            tokens) never leave the CU;
//   phase 2  the down-projection over that tile, W2 fragments through a second ring (its first loads, and the f32 residual
//            rows, are requested before the barrier that ends phase 1), then the LayerNorm epilogue of
//            bert_gemm_ln_w_kernel over 8 waves; the residual / output tile reuses the intermediate tile's LDS.
// One launch, 12 MB read and 12 MB written per layer instead of two launches moving 56 MB.
// AO = true puts the attention-output projection and the first LayerNorm in front (phase A): everything of a layer after the
// attention is then this one launch, and the rows between the two LayerNorms exist only in registers and LDS.
// IC = the intermediate size as a compile-time constant (0: the run-time argument I).  With IC every loop below unrolls completely and
// the weight rings' refills are unconditional straight-line code: the compiler then counts the loads in flight exactly (s_waitcnt
// vmcnt(N)).  With run-time trip counts and refills under a condition it cannot, and puts s_waitcnt vmcnt(0) at every loop head: the
// W2 ring drained every R2 k-steps and the W1 ring at every chunk — one full L2 round trip each, ~36 of them in a block's life, which
// IS most of that life (round 5, read off the ISA: profiles/r05/encoder_ring_waits.txt).
template <int CT, int CH, bool AO, int IC = 0>
__global__ __launch_bounds__(512) void bert_ffn_w_kernel(const _Float16* __restrict__ ctx, const half8* __restrict__ W0p,
                                                         const float* __restrict__ b0, const float* __restrict__ ln0w,
                                                         const float* __restrict__ ln0b, const half8* __restrict__ W1p,
                                                         const float* __restrict__ b1,
                                                         const half8* __restrict__ W2p, const float* __restrict__ b2,
                                                         float* __restrict__ x_f32, _Float16* __restrict__ x_h,
                                                         const float* __restrict__ lnw, const float* __restrict__ lnb, int M,
                                                         int I_arg, float eps) {
    constexpr int H = 64 * CT, BM = 32, NW = 8;
    constexpr bool FIXED = IC > 0;
    const int I = FIXED ? IC : I_arg;
    constexpr int KS1 = H / 32;            // k-steps of the up-projection = ring depth of the W1 stream
    constexpr int NT2 = 4 * CT / NW;       // 16-column tiles of the output per wave
#ifndef FSGPU_FFN_R2_AO
#define FSGPU_FFN_R2_AO 8
#endif
    constexpr int R2 = AO ? FSGPU_FFN_R2_AO : 12;        // ring depth of the W2 stream (AO also carries the rows between the LayerNorms)
    constexpr int XP = H + 4;              // floats per row of the residual / output tile
    constexpr int HP = H + 16;             // halves per row of the f16 x tile
    constexpr int XL = BM * (H / 4) / 512; // float4 pieces of the residual tile per thread
    static_assert((4 * CT) % NW == 0, "hidden must be a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) unsigned char ffn_smem[];
    float* red = reinterpret_cast<float*>(ffn_smem);                                      // [BM][NW]
    _Float16* Xh = reinterpret_cast<_Float16*>(ffn_smem + BM * NW * 4);                   // [BM][HP]
    _Float16* Is = Xh + BM * HP;                                                          // [BM][I + 16]
    float* Xs = reinterpret_cast<float*>(Is);                                             // [BM][XP], after phase 2
    const int ip = I + 16, ksteps2 = I / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm0 = blockIdx.x * BM;
    const int fr = lane & 15, q = lane >> 4, cq = q * 4;
    const int tpw = I / 16 / NW;           // intermediate tiles per wave (a multiple of CH)
    const int nchunks = tpw / CH;
    // W1 stream of this wave: fragment (chunk c, tile j, k-step ks) at w1[((CH c + j) * KS1 + ks) * 64]
    const half8* w1 = W1p + (size_t)(wave * tpw) * KS1 * 64 + lane;
    half8 r1[KS1][CH];
    if (!AO)
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int j = 0; j < CH; ++j) r1[ks][j] = w1[(j * KS1 + ks) * 64];
    f32x4 x1[2][NT2];                      // AO: the rows after the first LayerNorm (this lane's elements), the FFN's residual
    if (AO) {
        // phase A: x1 = LayerNorm(x + ctx W0^T + b0) (attention output projection, add_ln_raw) — bert_gemm_ln_w_kernel's work
        // on 8 waves, its f16 result going straight into the x tile of phase 1 and its f32 result staying in registers.
        // The context tile and the f32 residual rows borrow the intermediate tile's LDS.
        _Float16* Cs = Is;                                                    // [BM][HP]
        float* Xa = reinterpret_cast<float*>(Is + BM * HP);                   // [BM][XP]
        const half8* w0 = W0p + (size_t)(wave * NT2) * KS1 * 64 + lane;      // fragment (tile j, k-step ks) at w0[(j KS1 + ks) 64]
        half8 r0[KS1][NT2];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int j = 0; j < NT2; ++j) r0[ks][j] = w0[(j * KS1 + ks) * 64];
        {
            constexpr int PIECES = H / 8, AL = (BM * PIECES + 511) / 512;
            half8 ra[AL];
            f32x4 rx[XL];
#pragma unroll
            for (int x = 0; x < AL; ++x) {
                const int p = tid + 512 * x;
                if (p < BM * PIECES) {
                    const int r = p / PIECES, c = p % PIECES;
                    int row = bm0 + r;
                    row = row < M ? row : M - 1;
                    ra[x] = *(reinterpret_cast<const half8*>(ctx + (size_t)row * H) + c);
                }
            }
#pragma unroll
            for (int x = 0; x < XL; ++x) {
                const int p = tid + 512 * x, r = p / (H / 4), c4 = p % (H / 4);
                int row = bm0 + r;
                row = row < M ? row : M - 1;
                rx[x] = *(reinterpret_cast<const f32x4*>(x_f32 + (size_t)row * H) + c4);
            }
#pragma unroll
            for (int x = 0; x < AL; ++x) {
                const int p = tid + 512 * x;
                if (p < BM * PIECES) {
                    const int r = p / PIECES, c = p % PIECES;
                    *reinterpret_cast<half8*>(&Cs[r * HP + c * 8]) = ra[x];
                }
            }
#pragma unroll
            for (int x = 0; x < XL; ++x) {
                const int p = tid + 512 * x, r = p / (H / 4), c4 = p % (H / 4);
                *reinterpret_cast<f32x4*>(&Xa[r * XP + c4 * 4]) = rx[x];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT2; ++j) x1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            half8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(&Cs[(i * 16 + fr) * HP + ks * 32 + q * 8]);
#pragma unroll
            for (int j = 0; j < NT2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    x1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r0[ks][j], af[i], x1[i][j], 0, 0, 0);
        }
        float ps[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < NT2; ++j) {
                const int col = wave * 16 * NT2 + j * 16 + cq;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(b0 + col);
                const f32x4 xv = *reinterpret_cast<const f32x4*>(&Xa[(i * 16 + fr) * XP + col]);
                const f32x4 y = (x1[i][j] + bv) + xv;
                x1[i][j] = y;
                sm += (y[0] + y[1]) + (y[2] + y[3]);
            }
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            ps[i] = sm;
        }
        if (lane < 16) {
            red[(0 * 16 + lane) * NW + wave] = ps[0];
            red[(1 * 16 + lane) * NW + wave] = ps[1];
        }
        __syncthreads();
        float mu[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* p = red + (i * 16 + fr) * NW;
            mu[i] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) / (float)H;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float qs = 0.f;
#pragma unroll
            for (int j = 0; j < NT2; ++j) {
                const f32x4 d = x1[i][j] - mu[i];
                qs += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
            qs += __shfl_xor(qs, 16);
            qs += __shfl_xor(qs, 32);
            if (lane < 16) red[(i * 16 + lane) * NW + wave] = qs;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* p = red + (i * 16 + fr) * NW;
            const float var = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) / (float)H;
            const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
            for (int j = 0; j < NT2; ++j) {
                const int col = wave * 16 * NT2 + j * 16 + cq;
                const f32x4 g = *reinterpret_cast<const f32x4*>(ln0w + col);
                const f32x4 b = *reinterpret_cast<const f32x4*>(ln0b + col);
                const f32x4 y = (x1[i][j] - mu[i]) * inv * g + b;
                x1[i][j] = y;
                half4 h;
                h[0] = (_Float16)y[0];
                h[1] = (_Float16)y[1];
                h[2] = (_Float16)y[2];
                h[3] = (_Float16)y[3];
                *reinterpret_cast<half4*>(&Xh[(i * 16 + fr) * HP + col]) = h;
            }
        }
        __syncthreads();   // the x tile is complete; the statistics buffer and the borrowed LDS are free again
    } else {
    // phase 0: f16 rows of x -> LDS (one contiguous run of x_h, clamped at M)
    {
        constexpr int PIECES = H / 8, AL = (BM * PIECES + 511) / 512;
        half8 ra[AL];
#pragma unroll
        for (int x = 0; x < AL; ++x) {
            const int p = tid + 512 * x;
            if (p < BM * PIECES) {
                const int r = p / PIECES, c = p % PIECES;
                int row = bm0 + r;
                row = row < M ? row : M - 1;
                ra[x] = *(reinterpret_cast<const half8*>(x_h + (size_t)row * H) + c);
            }
        }
#pragma unroll
        for (int x = 0; x < AL; ++x) {
            const int p = tid + 512 * x;
            if (p < BM * PIECES) {
                const int r = p / PIECES, c = p % PIECES;
                *reinterpret_cast<half8*>(&Xh[r * HP + c * 8]) = ra[x];
            }
        }
    }
    __syncthreads();
    }
    // phase 1
    if (AO)
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int j = 0; j < CH; ++j) r1[ks][j] = w1[(j * KS1 + ks) * 64];
    auto chunk = [&](int c) {
        f32x4 acc[2][CH];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < CH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (the chunk's biases are requested BEFORE the ring's refills: loads return in order, so waiting for them in the epilogue does
        // not wait for the refills behind them)
        f32x4 bvs[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) bvs[j] = *reinterpret_cast<const f32x4*>(b1 + (wave * tpw + CH * c + j) * 16 + cq);
        const half8* wn = w1 + (size_t)(CH * (c + 1)) * KS1 * 64;   // the next chunk
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            half8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(&Xh[(i * 16 + fr) * HP + ks * 32 + q * 8]);
#pragma unroll
            for (int j = 0; j < CH; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r1[ks][j], af[i], acc[i][j], 0, 0, 0);
            if (c + 1 < nchunks)
#pragma unroll
                for (int j = 0; j < CH; ++j) r1[ks][j] = wn[(j * KS1 + ks) * 64];
            __builtin_amdgcn_sched_barrier(0);   // the refill is issued HERE, a ring's depth ahead of its use (left free, the scheduler sinks it next to the use: profiles/r05/enc_loop_forms_ab.txt)
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int col = (wave * tpw + CH * c + j) * 16 + cq;
            const f32x4 bv = bvs[j];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 y = acc[i][j] + bv;
                half4 h;
                h[0] = (_Float16)gelu_as_w(y[0]);
                h[1] = (_Float16)gelu_as_w(y[1]);
                h[2] = (_Float16)gelu_as_w(y[2]);
                h[3] = (_Float16)gelu_as_w(y[3]);
                *reinterpret_cast<half4*>(&Is[(i * 16 + fr) * ip + col]) = h;
            }
        }
    };
    if constexpr (FIXED) {
#pragma unroll
        for (int c = 0; c < IC / 16 / NW / CH; ++c) chunk(c);
    } else {
#pragma unroll 1
        for (int c = 0; c < nchunks; ++c) chunk(c);
    }
    // phase 2: W2 fragment (tile j, k-step ks) at w2[(j * ksteps2 + ks) * 64]
    const half8* w2 = W2p + (size_t)(wave * NT2) * ksteps2 * 64 + lane;
    half8 r2[R2][NT2];
#pragma unroll
    for (int d = 0; d < R2; ++d)
        if (d < ksteps2)
#pragma unroll
            for (int j = 0; j < NT2; ++j) r2[d][j] = w2[((size_t)j * ksteps2 + d) * 64];
    f32x4 rx[XL];                          // !AO: the residual rows, row-contiguous; parked in LDS after the k-loop
    if (!AO)
#pragma unroll
        for (int x = 0; x < XL; ++x) {
            const int p = tid + 512 * x, r = p / (H / 4), c4 = p % (H / 4);
            int row = bm0 + r;
            row = row < M ? row : M - 1;
            rx[x] = *(reinterpret_cast<const f32x4*>(x_f32 + (size_t)row * H) + c4);
        }
    __syncthreads();   // the intermediate tile is complete
    f32x4 acc[2][NT2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto ring_round = [&](int ks0) {
#pragma unroll
        for (int d = 0; d < R2; ++d) {
            const int ks = ks0 + d;
            if (ks < ksteps2) {
                half8 af[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(&Is[(i * 16 + fr) * ip + ks * 32 + q * 8]);
#pragma unroll
                for (int j = 0; j < NT2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r2[d][j], af[i], acc[i][j], 0, 0, 0);
                if (ks + R2 < ksteps2)
#pragma unroll
                    for (int j = 0; j < NT2; ++j) r2[d][j] = w2[((size_t)j * ksteps2 + ks + R2) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if constexpr (FIXED) {
#pragma unroll
        for (int ks0 = 0; ks0 < IC / 32; ks0 += R2) ring_round(ks0);
    } else {
#pragma unroll 1
        for (int ks0 = 0; ks0 < ksteps2; ks0 += R2) ring_round(ks0);
    }
    __syncthreads();   // every wave is done with the intermediate tile: its space becomes the residual / output tile
    if (!AO) {
#pragma unroll
        for (int x = 0; x < XL; ++x) {
            const int p = tid + 512 * x, r = p / (H / 4), c4 = p % (H / 4);
            *reinterpret_cast<f32x4*>(&Xs[r * XP + c4 * 4]) = rx[x];
        }
        __syncthreads();
    }
    // epilogue: rows i * 16 + fr, columns wave * 16 NT2 + j * 16 + cq .. + 3
    float psum[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const int col = wave * 16 * NT2 + j * 16 + cq;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + col);
            const f32x4 xv = AO ? x1[i][j] : *reinterpret_cast<const f32x4*>(&Xs[(i * 16 + fr) * XP + col]);
            const f32x4 y = (acc[i][j] + bv) + xv;
            acc[i][j] = y;
            s += (y[0] + y[1]) + (y[2] + y[3]);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        psum[i] = s;
    }
    if (lane < 16) {
        red[(0 * 16 + lane) * NW + wave] = psum[0];
        red[(1 * 16 + lane) * NW + wave] = psum[1];
    }
    __syncthreads();
    float mean[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* p = red + (i * 16 + fr) * NW;
        mean[i] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) / (float)H;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float qs = 0.f;
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const f32x4 d = acc[i][j] - mean[i];
            qs += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        qs += __shfl_xor(qs, 16);
        qs += __shfl_xor(qs, 32);
        if (lane < 16) red[(i * 16 + lane) * NW + wave] = qs;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* p = red + (i * 16 + fr) * NW;
        const float var = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) / (float)H;
        const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const int col = wave * 16 * NT2 + j * 16 + cq;
            const f32x4 g = *reinterpret_cast<const f32x4*>(lnw + col);
            const f32x4 b = *reinterpret_cast<const f32x4*>(lnb + col);
            *reinterpret_cast<f32x4*>(&Xs[(i * 16 + fr) * XP + col]) = (acc[i][j] - mean[i]) * inv * g + b;   // own element
        }
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < XL; ++x) {
        const int pp = tid + 512 * x, r = pp / (H / 4), c4 = pp % (H / 4);
        const int row = bm0 + r;
        if (row >= M) continue;
        const f32x4 y = *reinterpret_cast<const f32x4*>(&Xs[r * XP + c4 * 4]);
        *(reinterpret_cast<f32x4*>(x_f32 + (size_t)row * H) + c4) = y;
        half4 h;
        h[0] = (_Float16)y[0];
        h[1] = (_Float16)y[1];
        h[2] = (_Float16)y[2];
        h[3] = (_Float16)y[3];
        *(reinterpret_cast<half4*>(x_h + (size_t)row * H) + c4) = h;
    }
}

// The post-attention block of a layer (bert_ffn_w_kernel<CT, 2, true>) for LARGE M — the documents of an index build —, 64 rows per
// block.  Why: that kernel streams the layer's 2.65 MB of packed weights out of L2 once per 32-row block, 1.36 GB per layer at
// 16,384 tokens, and every 1 KB fragment a wave fetches feeds two matrix instructions: 25 B / clock / CU of weight stream is a fifth
// of what four SIMDs consume at full rate (profiles/r05/encoder_large_m_pmc.txt: 88 us per layer, matrix pipe 0.20 busy, the L2s
// delivering 15.4 TB/s).  With 64 rows a fragment feeds FOUR instructions and the block count halves: the same stream does twice
// the work.  What had to change to fit 64 rows into 160 KB of LDS and 256 registers:
//   * the intermediate tile holds HALF the intermediate columns at a time: up-projection + GELU of half h, then the down-projection's
//     partial sums over that half (accumulators live across both halves), barrier, the other half;
//   * the f32 residual of the first LayerNorm is read in FRAGMENT layout straight from global memory (64-byte runs per row) instead
//     of through an LDS tile, and the rows between the two LayerNorms — the FFN's residual — go back to x_f32 in the same layout
//     (the lane that wrote an element is the lane that reads it again in the epilogue: no barrier involved) instead of staying in 48
//     registers through the FFN.
// Same per-element arithmetic and operation order as bert_ffn_w_kernel: the outputs are bit-identical.
// Lab (-DFSGPU_FFN_STAMPS): where a block of bert_ffn_w64_kernel spends its life.  Wave 0 and wave 7 of every block add the clock
// at seven points to a device table (fsgpu_lab_ffn_stamps reads and clears it): phase A | up-projection + GELU of half 0 | down-
// projection over half 0 | the same for half 1 | epilogue.
#ifdef FSGPU_FFN_STAMPS
__device__ unsigned long long g_ffn_stamps[2][8];
#define FFN_STAMP(slot)                                                                                                        \
    do {                                                                                                                         \
        if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 7))                                      \
            atomicAdd(&g_ffn_stamps[(threadIdx.x >> 6) == 7][(slot)], (unsigned long long)__builtin_readcyclecounter());          \
        if ((slot) == 6 && threadIdx.x == 0) atomicAdd(&g_ffn_stamps[0][7], 1ull);                                               \
    } while (0)
#else
#define FFN_STAMP(slot) \
    do {                \
    } while (0)
#endif

template <int CT, int IC>   // IC = the intermediate size, a compile-time constant here (see bert_ffn_w_kernel)
__global__ __launch_bounds__(512) void bert_ffn_w64_kernel(const _Float16* __restrict__ ctx, const half8* __restrict__ W0p,
                                                           const float* __restrict__ b0, const float* __restrict__ ln0w,
                                                           const float* __restrict__ ln0b, const half8* __restrict__ W1p,
                                                           const float* __restrict__ b1, const half8* __restrict__ W2p,
                                                           const float* __restrict__ b2, float* x_f32, _Float16* __restrict__ x_h,
                                                           const float* __restrict__ lnw, const float* __restrict__ lnb, int M, int I_arg,
                                                           float eps) {
#ifndef FSGPU_FFN64_R2
#define FSGPU_FFN64_R2 8
#endif
    constexpr int H = 64 * CT, BM = 64, RT = BM / 16, NW = 8, CH = 2, R2 = FSGPU_FFN64_R2;
    static_assert(IC > 0 && IC % 512 == 0, "halves of the intermediate in chunks of two tiles over 8 waves");
    constexpr int I = IC;
    (void)I_arg;
    constexpr int KS1 = H / 32;            // k-steps of the up-projection = ring depth of the W1 stream
    constexpr int NT2 = 4 * CT / NW;       // 16-column tiles of the output per wave
    constexpr int XP = H + 4;              // floats per row of the output tile
    constexpr int HP = H + 16;             // halves per row of the f16 x tile
    constexpr int XL = BM * (H / 4) / 512; // float4 pieces of the output tile per thread
    static_assert((4 * CT) % NW == 0, "hidden must be a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) unsigned char ffn_smem[];
    float* red = reinterpret_cast<float*>(ffn_smem);                                      // [BM][NW]
    _Float16* Xh = reinterpret_cast<_Float16*>(ffn_smem + BM * NW * 4);                   // [BM][HP]
    _Float16* Is = Xh + BM * HP;                                                          // [BM][I / 2 + 16]
    float* Xs = reinterpret_cast<float*>(Is);                                             // [BM][XP], after the last half
    constexpr int IH = I / 2, ip = IH + 16, ksteps2 = I / 32, ksh = IH / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm0 = blockIdx.x * BM;
    const int fr = lane & 15, q = lane >> 4, cq = q * 4;
    constexpr int tpw = IH / 16 / NW;      // intermediate tiles per wave and half (a multiple of CH)
    constexpr int nchunks = tpw / CH;
    constexpr int NSTEP = nchunks * KS1;   // (chunk, k-step) pairs of a half, the order the W1 stream is consumed in
    constexpr int D1 = NSTEP < 8 ? NSTEP : 8;   // ring depth of the W1 stream in k-steps: CONTINUOUS over the chunks (a ring one whole chunk deep — 96 registers at hidden = 384 — does not fit next to 64 rows of accumulators)
    FFN_STAMP(0);
    int grow[RT];                          // this lane's rows (fragment layout), clamped at M
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int r = bm0 + i * 16 + fr;
        grow[i] = r < M ? r : M - 1;
    }
    // ---- phase A: x1 = LayerNorm(x + ctx W0^T + b0) -> f16 into the x tile, f32 back to x_f32 ----------------------------------
    {
        _Float16* Cs = Is;                                                    // [BM][HP], borrowed
        const half8* w0 = W0p + (size_t)(wave * NT2) * KS1 * 64 + lane;      // fragment (tile j, k-step ks) at w0[(j KS1 + ks) 64]
        constexpr int R0 = KS1 <= 8 ? KS1 : 6;   // W0 streams through a ring (the whole slice — 144 registers at hidden = 384 — does not fit next to 64 rows of accumulators)
        static_assert(KS1 % R0 == 0, "ring depth divides the k-steps");
        half8 r0[R0][NT2];
#pragma unroll
        for (int d = 0; d < R0; ++d)
#pragma unroll
            for (int j = 0; j < NT2; ++j) r0[d][j] = w0[(j * KS1 + d) * 64];
        {
            constexpr int PIECES = H / 8, AL = (BM * PIECES + 511) / 512;
            half8 ra[AL];
#pragma unroll
            for (int x = 0; x < AL; ++x) {
                const int p = tid + 512 * x;
                if (p < BM * PIECES) {
                    const int r = p / PIECES, c = p % PIECES;
                    int row = bm0 + r;
                    row = row < M ? row : M - 1;
                    ra[x] = *(reinterpret_cast<const half8*>(ctx + (size_t)row * H) + c);
                }
            }
#pragma unroll
            for (int x = 0; x < AL; ++x) {
                const int p = tid + 512 * x;
                if (p < BM * PIECES) {
                    const int r = p / PIECES, c = p % PIECES;
                    *reinterpret_cast<half8*>(&Cs[r * HP + c * 8]) = ra[x];
                }
            }
        }
        __syncthreads();
        f32x4 x1[RT][NT2];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < NT2; ++j) x1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const int d = ks % R0;
            half8 af[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) af[i] = *reinterpret_cast<const half8*>(&Cs[(i * 16 + fr) * HP + ks * 32 + q * 8]);
#pragma unroll
            for (int j = 0; j < NT2; ++j)
#pragma unroll
                for (int i = 0; i < RT; ++i)
                    x1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r0[d][j], af[i], x1[i][j], 0, 0, 0);
            if (ks + R0 < KS1)
#pragma unroll
                for (int j = 0; j < NT2; ++j) r0[d][j] = w0[(j * KS1 + ks + R0) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 g0[NT2], be0[NT2];   // the LayerNorm's weights: a round trip that overlaps the row statistics
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const int col = wave * 16 * NT2 + j * 16 + cq;
            g0[j] = *reinterpret_cast<const f32x4*>(ln0w + col);
            be0[j] = *reinterpret_cast<const f32x4*>(ln0b + col);
        }
        float ps[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < NT2; ++j) {
                const int col = wave * 16 * NT2 + j * 16 + cq;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(b0 + col);
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x_f32 + (size_t)grow[i] * H + col);
                const f32x4 y = (x1[i][j] + bv) + xv;
                x1[i][j] = y;
                sm += (y[0] + y[1]) + (y[2] + y[3]);
            }
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            ps[i] = sm;
        }
        if (lane < 16)
#pragma unroll
            for (int i = 0; i < RT; ++i) red[(i * 16 + lane) * NW + wave] = ps[i];
        __syncthreads();
        float mu[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const float* p = red + (i * 16 + fr) * NW;
            mu[i] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) / (float)H;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            float qs = 0.f;
#pragma unroll
            for (int j = 0; j < NT2; ++j) {
                const f32x4 d = x1[i][j] - mu[i];
                qs += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
            qs += __shfl_xor(qs, 16);
            qs += __shfl_xor(qs, 32);
            if (lane < 16) red[(i * 16 + lane) * NW + wave] = qs;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const float* p = red + (i * 16 + fr) * NW;
            const float var = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) / (float)H;
            const float inv = 1.0f / sqrtf(var + eps);
            const bool real = bm0 + i * 16 + fr < M;
#pragma unroll
            for (int j = 0; j < NT2; ++j) {
                const int col = wave * 16 * NT2 + j * 16 + cq;
                const f32x4 g = g0[j];
                const f32x4 b = be0[j];
                const f32x4 y = (x1[i][j] - mu[i]) * inv * g + b;
                if (real) *reinterpret_cast<f32x4*>(x_f32 + (size_t)grow[i] * H + col) = y;   // the FFN's residual: read back by this lane
                half4 h;
                h[0] = (_Float16)y[0];
                h[1] = (_Float16)y[1];
                h[2] = (_Float16)y[2];
                h[3] = (_Float16)y[3];
                *reinterpret_cast<half4*>(&Xh[(i * 16 + fr) * HP + col]) = h;
            }
        }
    }
    FFN_STAMP(1);
    // the down-projection's accumulators, over both halves
    f32x4 acc2[RT][NT2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NT2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const half8* w2 = W2p + (size_t)(wave * NT2) * ksteps2 * 64 + lane;   // W2 fragment (tile j, k-step ks) at w2[(j * ksteps2 + ks) * 64]
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        // ---- phase 1: GELU(x1 W1^T + b1) over intermediate columns [IH h, IH (h + 1)) -> the intermediate tile ---------------------
        // W1 stream of this wave: fragment (chunk c, tile j, k-step ks) at w1[((CH c + j) * KS1 + ks) * 64]
        const half8* w1 = W1p + (size_t)(h * (IH / 16) + wave * tpw) * KS1 * 64 + lane;
        half8 r1[D1][CH];
#pragma unroll
        for (int s0 = 0; s0 < D1; ++s0)
#pragma unroll
            for (int j = 0; j < CH; ++j) r1[s0][j] = w1[((CH * (s0 / KS1) + j) * KS1 + s0 % KS1) * 64];
        __syncthreads();   // h = 0: the x tile is complete, the borrowed LDS is free; h = 1: every wave is done with the first half's tile
#pragma unroll
        for (int c = 0; c < nchunks; ++c) {
            f32x4 acc[RT][CH];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 bvs[CH];   // requested before the ring's refills (loads return in order)
#pragma unroll
            for (int j = 0; j < CH; ++j) bvs[j] = *reinterpret_cast<const f32x4*>(b1 + h * IH + (wave * tpw + CH * c + j) * 16 + cq);
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int st = c * KS1 + ks, slot = st % D1;
                half8 af[RT];
#pragma unroll
                for (int i = 0; i < RT; ++i) af[i] = *reinterpret_cast<const half8*>(&Xh[(i * 16 + fr) * HP + ks * 32 + q * 8]);
#pragma unroll
                for (int j = 0; j < CH; ++j)
#pragma unroll
                    for (int i = 0; i < RT; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r1[slot][j], af[i], acc[i][j], 0, 0, 0);
                if (st + D1 < NSTEP) {
                    const int s2 = st + D1;
#pragma unroll
                    for (int j = 0; j < CH; ++j) r1[slot][j] = w1[((CH * (s2 / KS1) + j) * KS1 + s2 % KS1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);   // (the refill stays a ring's depth ahead of its use)
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int lcol = (wave * tpw + CH * c + j) * 16 + cq;   // within the half
                const f32x4 bv = bvs[j];
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    const f32x4 y = acc[i][j] + bv;
                    half4 hh;
                    hh[0] = (_Float16)gelu_as_w(y[0]);
                    hh[1] = (_Float16)gelu_as_w(y[1]);
                    hh[2] = (_Float16)gelu_as_w(y[2]);
                    hh[3] = (_Float16)gelu_as_w(y[3]);
                    *reinterpret_cast<half4*>(&Is[(i * 16 + fr) * ip + lcol]) = hh;
                }
            }
        }
        FFN_STAMP(2 + 2 * h);
        // ---- phase 2: the down-projection's partial sums over this half ----------------------------------------------------------
        half8 r2[R2][NT2];
#pragma unroll
        for (int d = 0; d < R2; ++d)
            if (d < ksh)
#pragma unroll
                for (int j = 0; j < NT2; ++j) r2[d][j] = w2[((size_t)j * ksteps2 + h * ksh + d) * 64];
        __syncthreads();   // the half's intermediate tile is complete (an LDS-only barrier that lets the requested fragments fly on: measured the same)
        auto ring_round = [&](int ks0) {
#pragma unroll
            for (int d = 0; d < R2; ++d) {
                const int ks = ks0 + d;
                if (ks < ksh) {
                    half8 af[RT];
#pragma unroll
                    for (int i = 0; i < RT; ++i) af[i] = *reinterpret_cast<const half8*>(&Is[(i * 16 + fr) * ip + ks * 32 + q * 8]);
#pragma unroll
                    for (int j = 0; j < NT2; ++j)
#pragma unroll
                        for (int i = 0; i < RT; ++i)
                            acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r2[d][j], af[i], acc2[i][j], 0, 0, 0);
                    if (ks + R2 < ksh)
#pragma unroll
                        for (int j = 0; j < NT2; ++j) r2[d][j] = w2[((size_t)j * ksteps2 + h * ksh + ks + R2) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
#pragma unroll
        for (int ks0 = 0; ks0 < ksh; ks0 += R2) ring_round(ks0);
        FFN_STAMP(3 + 2 * h);
    }
    // ---- epilogue: x = LayerNorm(x1 + acc2 + b2); rows i * 16 + fr, columns wave * 16 NT2 + j * 16 + cq .. + 3 ---------------------
    f32x4 g2[NT2], be2[NT2];
#pragma unroll
    for (int j = 0; j < NT2; ++j) {   // (the LayerNorm's weights: a round trip that overlaps the row statistics)
        const int col = wave * 16 * NT2 + j * 16 + cq;
        g2[j] = *reinterpret_cast<const f32x4*>(lnw + col);
        be2[j] = *reinterpret_cast<const f32x4*>(lnb + col);
    }
    float psum[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const int col = wave * 16 * NT2 + j * 16 + cq;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + col);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x_f32 + (size_t)grow[i] * H + col);   // what this lane stored in phase A
            const f32x4 y = (acc2[i][j] + bv) + xv;
            acc2[i][j] = y;
            s += (y[0] + y[1]) + (y[2] + y[3]);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        psum[i] = s;
    }
    if (lane < 16)
#pragma unroll
        for (int i = 0; i < RT; ++i) red[(i * 16 + lane) * NW + wave] = psum[i];
    __syncthreads();   // (also: every wave is done with the last intermediate tile — its space becomes the output tile)
    float mean[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const float* p = red + (i * 16 + fr) * NW;
        mean[i] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) / (float)H;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        float qs = 0.f;
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const f32x4 d = acc2[i][j] - mean[i];
            qs += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        qs += __shfl_xor(qs, 16);
        qs += __shfl_xor(qs, 32);
        if (lane < 16) red[(i * 16 + lane) * NW + wave] = qs;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const float* p = red + (i * 16 + fr) * NW;
        const float var = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) / (float)H;
        const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const int col = wave * 16 * NT2 + j * 16 + cq;
            const f32x4 g = g2[j];
            const f32x4 b = be2[j];
            *reinterpret_cast<f32x4*>(&Xs[(i * 16 + fr) * XP + col]) = (acc2[i][j] - mean[i]) * inv * g + b;   // own element
        }
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < XL; ++x) {
        const int pp = tid + 512 * x, r = pp / (H / 4), c4 = pp % (H / 4);
        const int row = bm0 + r;
        if (row >= M) continue;
        const f32x4 y = *reinterpret_cast<const f32x4*>(&Xs[r * XP + c4 * 4]);
        *(reinterpret_cast<f32x4*>(x_f32 + (size_t)row * H) + c4) = y;
        half4 hh;
        hh[0] = (_Float16)y[0];
        hh[1] = (_Float16)y[1];
        hh[2] = (_Float16)y[2];
        hh[3] = (_Float16)y[3];
        *(reinterpret_cast<half4*>(x_h + (size_t)row * H) + c4) = hh;
    }
    FFN_STAMP(6);
}

// ---- launchers ------------------------------------------------------------------------------------------

hipError_t launch_bert_pack_w(const void* w_h, void* packed_h, int N, int K, hipStream_t stream) {
    if (N % 16 != 0 || K % 32 != 0) return hipErrorInvalidValue;
    const size_t pieces = (size_t)(N / 16) * (K / 32) * 64;
    hipLaunchKernelGGL(bert_pack_w_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream,
                       static_cast<const _Float16*>(w_h), static_cast<_Float16*>(packed_h), N, K);
    return hipGetLastError();
}

// K = hidden in {128, 256, 384}, N % 128 == 0
bool bert_gemm_w_supported(int N, int K) { return (K == 128 || K == 256 || K == 384) && N % 128 == 0; }

template <int EPI, int KS>
static void launch_gemm_w_t(const void* a_h, const void* wp, const float* bias, float* out_f32, void* out_h, int M, int N,
                            hipStream_t stream) {
    // large M: the weight-stationary form — about three blocks per CU, each walking its share of the row tiles
    const int tiles = (M + 63) / 64, cols = N / 128;
    if (tiles >= FSGPU_GEMM_WQ_MIN_TILES && cols >= 1) {
        // thousands of rows: 64 x 128 block tiles with 64 x 64... per wave 32 x 64, two blocks per CU, each walking its share of the row tiles
        int device = 0, cus = 256;
        if (hipGetDevice(&device) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        constexpr int CT = KS >= 12 ? 3 : 4;          // column tiles per wave (see the kernel)
        constexpr int BN = 32 * CT;
        if (N % BN == 0) {
            const int qcols = N / BN;
            // per XCD: qcols column blocks x walkers row walkers on its cus / 8 CUs, two blocks per CU
            const int per_xcd = std::max(2, 2 * cus / 8);
            int walkers = std::max(1, per_xcd / qcols);
            walkers = std::min(walkers, std::max(1, (tiles + 7) / 8));
            hipLaunchKernelGGL((bert_gemm_wq_kernel<EPI, KS, CT>), dim3(8 * qcols * walkers), dim3(256), 0, stream,
                               static_cast<const _Float16*>(a_h), static_cast<const half8*>(wp), bias, out_f32,
                               static_cast<_Float16*>(out_h), M, N, qcols, walkers);
            return;
        }
    }
    if (tiles >= FSGPU_GEMM_WP_MIN_TILES && cols >= 1) {
        int device = 0, cus = 256;
        if (hipGetDevice(&device) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        int rows_per_col = std::max(1, (2 * cus + cols - 1) / cols);   // 2 blocks per CU: 68 KB of LDS and ~200 VGPRs each
        rows_per_col = std::min(rows_per_col, tiles);
        hipLaunchKernelGGL((bert_gemm_wp_kernel<EPI, KS>), dim3(cols, rows_per_col), dim3(256), 0, stream,
                           static_cast<const _Float16*>(a_h), static_cast<const half8*>(wp), bias, out_f32,
                           static_cast<_Float16*>(out_h), M, N);
        return;
    }
    hipLaunchKernelGGL((bert_gemm_w_kernel<EPI, KS>), dim3(N / 128, (M + 63) / 64), dim3(256), 0, stream,
                       static_cast<const _Float16*>(a_h), static_cast<const half8*>(wp), bias, out_f32,
                       static_cast<_Float16*>(out_h), M, N);
}

hipError_t launch_bert_gemm_w(const void* a_h, const void* wp, const float* bias, float* out_f32, void* out_h, int M, int N,
                              int K, int epilogue, hipStream_t stream) {
    if (!bert_gemm_w_supported(N, K) || epilogue < 0 || epilogue > 2) return hipErrorInvalidValue;
#define FSGPU_GW(E)                                                                            \
    do {                                                                                       \
        if (K == 384) launch_gemm_w_t<E, 12>(a_h, wp, bias, out_f32, out_h, M, N, stream);     \
        else if (K == 256) launch_gemm_w_t<E, 8>(a_h, wp, bias, out_f32, out_h, M, N, stream); \
        else launch_gemm_w_t<E, 4>(a_h, wp, bias, out_f32, out_h, M, N, stream);               \
    } while (0)
    if (epilogue == 0) FSGPU_GW(0);
    else if (epilogue == 1) FSGPU_GW(1);
    else FSGPU_GW(2);
#undef FSGPU_GW
    return hipGetLastError();
}

static size_t gemm_ln_w_lds(int hidden, int K) {
    return (size_t)32 * 4 * 4 + (size_t)32 * (K + 16) * 2 + (size_t)32 * (hidden + 4) * 4;
}

// hidden in {128, 256, 384}; the 32 x K activation tile and the 32 x hidden residual tile must fit the 160 KB of LDS
bool bert_gemm_ln_w_supported(int hidden, int K) {
    return (hidden == 384 || hidden == 256 || hidden == 128) && K % 32 == 0 && K >= 32 &&
           gemm_ln_w_lds(hidden, K) <= (size_t)160 * 1024;
}

template <int CT>
static hipError_t launch_gemm_ln_w_t(const void* a_h, const void* wp, const float* bias, float* x_f32, void* x_h,
                                     const float* lnw, const float* lnb, int M, int K, float eps, hipStream_t stream) {
    const size_t lds = gemm_ln_w_lds(64 * CT, K);
    auto kern = bert_gemm_ln_w_kernel<CT, FSGPU_LN_RING>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((M + 31) / 32), dim3(256), lds, stream, static_cast<const _Float16*>(a_h),
                       static_cast<const half8*>(wp), bias, x_f32, static_cast<_Float16*>(x_h), lnw, lnb, M, K, eps);
    return hipGetLastError();
}

hipError_t launch_bert_gemm_ln_w(const void* a_h, const void* wp, const float* bias, float* x_f32, void* x_h,
                                 const float* lnw, const float* lnb, int M, int hidden, int K, float eps, hipStream_t stream) {
    if (!bert_gemm_ln_w_supported(hidden, K)) return hipErrorInvalidValue;
    switch (hidden) {
        case 384: return launch_gemm_ln_w_t<6>(a_h, wp, bias, x_f32, x_h, lnw, lnb, M, K, eps, stream);
        case 256: return launch_gemm_ln_w_t<4>(a_h, wp, bias, x_f32, x_h, lnw, lnb, M, K, eps, stream);
        default: return launch_gemm_ln_w_t<2>(a_h, wp, bias, x_f32, x_h, lnw, lnb, M, K, eps, stream);
    }
}

static size_t ffn_w_lds(int hidden, int inter) {
    return (size_t)32 * 8 * 4 + (size_t)32 * (hidden + 16) * 2 + (size_t)32 * (inter + 16) * 2;
}

// hidden in {128, 256, 384}; inter a multiple of 256 (8 waves x chunks of 2 tiles; of 384 for the 3-tile chunks) and at least
// as wide as the f32 residual tile that later reuses its LDS; f16 x tile + intermediate tile within 160 KB
bool bert_ffn_w_supported(int hidden, int inter) {
    return (hidden == 384 || hidden == 256 || hidden == 128) && inter >= 256 && inter % 256 == 0 &&
           (size_t)(inter + 16) * 2 >= (size_t)(hidden + 4) * 4 && ffn_w_lds(hidden, inter) <= (size_t)160 * 1024;
}

template <int CT, bool AO>
static hipError_t launch_ffn_w_t(const void* ctx_h, const void* w0p, const float* b0, const float* ln0w, const float* ln0b,
                                 const void* w1p, const float* b1, const void* w2p, const float* b2, float* x_f32, void* x_h,
                                 const float* lnw, const float* lnb, int M, int I, float eps, hipStream_t stream) {
    const size_t lds = ffn_w_lds(64 * CT, I);
    // three 16-column tiles per chunk when the wave's share divides, else two; the AO form (which also carries the rows
    // between the LayerNorms in registers) always two: with three, hidden = 384 spills
    auto kern = (!AO && (I / 16 / 8) % 3 == 0) ? bert_ffn_w_kernel<CT, 3, AO> : bert_ffn_w_kernel<CT, 2, AO>;
    if (AO && I == 4 * 64 * CT) kern = bert_ffn_w_kernel<CT, 2, true, 4 * 64 * CT>;   // the usual inter = 4 hidden: compile-time trip counts (see the kernel)
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((M + 31) / 32), dim3(512), lds, stream, static_cast<const _Float16*>(ctx_h),
                       static_cast<const half8*>(w0p), b0, ln0w, ln0b, static_cast<const half8*>(w1p), b1,
                       static_cast<const half8*>(w2p), b2, x_f32, static_cast<_Float16*>(x_h), lnw, lnb, M, I, eps);
    return hipGetLastError();
}

hipError_t launch_bert_ffn_w(const void* w1p, const float* b1, const void* w2p, const float* b2, float* x_f32, void* x_h,
                             const float* lnw, const float* lnb, int M, int hidden, int inter, float eps, hipStream_t stream) {
    if (!bert_ffn_w_supported(hidden, inter)) return hipErrorInvalidValue;
    switch (hidden) {
        case 384: return launch_ffn_w_t<6, false>(nullptr, nullptr, nullptr, nullptr, nullptr, w1p, b1, w2p, b2, x_f32, x_h, lnw, lnb, M, inter, eps, stream);
        case 256: return launch_ffn_w_t<4, false>(nullptr, nullptr, nullptr, nullptr, nullptr, w1p, b1, w2p, b2, x_f32, x_h, lnw, lnb, M, inter, eps, stream);
        default: return launch_ffn_w_t<2, false>(nullptr, nullptr, nullptr, nullptr, nullptr, w1p, b1, w2p, b2, x_f32, x_h, lnw, lnb, M, inter, eps, stream);
    }
}

// the borrowed LDS of phase A (context tile + f32 residual tile) must fit inside the intermediate tile
bool bert_post_attn_w_supported(int hidden, int inter) {
    return bert_ffn_w_supported(hidden, inter) &&
           (size_t)(inter + 16) * 2 >= (size_t)(hidden + 16) * 2 + (size_t)(hidden + 4) * 4;
}

// the 64-row form (bert_ffn_w64_kernel): halves of the intermediate in chunks of two tiles over 8 waves; the f32 output tile and the
// borrowed context tile inside the half intermediate tile; everything within 160 KB
static size_t ffn_w64_lds(int hidden, int inter) {
    return (size_t)64 * 8 * 4 + (size_t)64 * (hidden + 16) * 2 + (size_t)64 * (inter / 2 + 16) * 2;
}
static bool ffn_w64_supported(int hidden, int inter) {
    return (hidden == 384 || hidden == 256 || hidden == 128) && inter == 4 * hidden &&   // (built for the usual shape: compile-time trip counts)
           (size_t)(inter / 2 + 16) * 2 >= (size_t)(hidden + 4) * 4 && ffn_w64_lds(hidden, inter) <= (size_t)160 * 1024;
}

template <int CT>
static hipError_t launch_ffn_w64_t(const void* ctx_h, const void* w0p, const float* b0, const float* ln0w, const float* ln0b,
                                   const void* w1p, const float* b1, const void* w2p, const float* b2, float* x_f32, void* x_h,
                                   const float* lnw, const float* lnb, int M, int I, float eps, hipStream_t stream) {
    const size_t lds = ffn_w64_lds(64 * CT, I);
    auto kern = bert_ffn_w64_kernel<CT, 4 * 64 * CT>;   // (ffn_w64_supported: inter = 4 hidden)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)160 * 1024));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((M + 63) / 64), dim3(512), lds, stream, static_cast<const _Float16*>(ctx_h),
                       static_cast<const half8*>(w0p), b0, ln0w, ln0b, static_cast<const half8*>(w1p), b1,
                       static_cast<const half8*>(w2p), b2, x_f32, static_cast<_Float16*>(x_h), lnw, lnb, M, I, eps);
    return hipGetLastError();
}

#ifndef FSGPU_FFN_W64_MIN_ROWS
#define FSGPU_FFN_W64_MIN_ROWS 8193   // more 32-row blocks than one round of the chip's 256 CUs: from there 64-row blocks halve the weight stream
#endif

hipError_t launch_bert_post_attn_w(const void* ctx_h, const void* w0p, const float* b0, const float* ln0w, const float* ln0b,
                                   const void* w1p, const float* b1, const void* w2p, const float* b2, float* x_f32, void* x_h,
                                   const float* lnw, const float* lnb, int M, int hidden, int inter, float eps,
                                   hipStream_t stream) {
    if (!bert_post_attn_w_supported(hidden, inter)) return hipErrorInvalidValue;
    if (M >= FSGPU_FFN_W64_MIN_ROWS && ffn_w64_supported(hidden, inter)) {
        switch (hidden) {
            case 384: return launch_ffn_w64_t<6>(ctx_h, w0p, b0, ln0w, ln0b, w1p, b1, w2p, b2, x_f32, x_h, lnw, lnb, M, inter, eps, stream);
            case 256: return launch_ffn_w64_t<4>(ctx_h, w0p, b0, ln0w, ln0b, w1p, b1, w2p, b2, x_f32, x_h, lnw, lnb, M, inter, eps, stream);
            default: return launch_ffn_w64_t<2>(ctx_h, w0p, b0, ln0w, ln0b, w1p, b1, w2p, b2, x_f32, x_h, lnw, lnb, M, inter, eps, stream);
        }
    }
    switch (hidden) {
        case 384: return launch_ffn_w_t<6, true>(ctx_h, w0p, b0, ln0w, ln0b, w1p, b1, w2p, b2, x_f32, x_h, lnw, lnb, M, inter, eps, stream);
        case 256: return launch_ffn_w_t<4, true>(ctx_h, w0p, b0, ln0w, ln0b, w1p, b1, w2p, b2, x_f32, x_h, lnw, lnb, M, inter, eps, stream);
        default: return launch_ffn_w_t<2, true>(ctx_h, w0p, b0, ln0w, ln0b, w1p, b1, w2p, b2, x_f32, x_h, lnw, lnb, M, inter, eps, stream);
    }
}

#ifdef FSGPU_FFN_STAMPS
extern "C" int fsgpu_lab_ffn_stamps(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_ffn_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    unsigned long long zero[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ffn_stamps), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif

}  // namespace fsgpu
