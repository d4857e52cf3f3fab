// f32_kernels.hip — Quantization::F32 slabs (FSVI quantization byte 0, crates/frankensearch-index/src/lib.rs:203-208):
// rows are raw little-endian f32.  F16 is the reference's default and the format every BASELINE config uses; F32 files
// have one fused scan + top-k kernel (scan_topk_f32_kernel, 1 / 2 / 4 queries per pass, k <= 256, dim % 8 == 0) and the general
// path (score every row -> radix sort -> top k) for everything else.
//
// dot_product_f32_bytes_f32 (simd.rs:581-702): four 8-lane accumulators over groups of 32 elements, separate multiply
// and add (this file is compiled with -ffp-contract=off like the others), (acc0+acc1)+(acc2+acc3), THEN the leftover
// 8-element chunks are added to that sum (the f16 kernel adds them to acc0 before the combine), reduce_add, and a
// fused multiply-add for the last dim % 8 elements.
//
// Mapping: a quad of lanes per row; lane a owns accumulator a, i.e. elements [32 g + 8 a, 32 g + 8 a + 8) of every
// group — the four lanes of a quad read 128 contiguous bytes.  The combine is two DPP quad butterflies (IEEE add is
// commutative, so all four lanes end with the same bits); leftovers and tail are computed redundantly by the quad.
#include "kernels.hpp"
#include "scan_common.hpp"

namespace fsgpu {

using namespace scan_detail;

// GATHER = false: item i is row i of the slab; out_packed[i] = (score, row_base + i) or kEmpty when the row is
//   tombstoned / filtered.  GATHER = true: item i is rows[i] (a global id; ids outside the shard are skipped),
//   out_scores[i] = the dot.
template <bool GATHER>
__global__ __launch_bounds__(256) void dot_rows_f32_kernel(ScanArgs args, const uint32_t* __restrict__ rows, uint32_t n,
                                                           u64* __restrict__ out_packed, float* __restrict__ out_scores,
                                                           int q_index) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* qs = reinterpret_cast<float*>(smem);
    const int dim = (int)args.dim;
    const int tid = threadIdx.x, a = tid & 3;
    for (int i = tid; i < dim; i += 256) qs[i] = args.queries[(size_t)q_index * dim + i];
    __syncthreads();
    const uint32_t item = (blockIdx.x * 256u + (uint32_t)tid) >> 2;
    const bool in_range = item < n;
    uint32_t row = in_range ? (GATHER ? rows[item] - args.row_base : item) : 0;  // wraps for ids below the shard base
    const bool mine = in_range && row < args.nrows;
    if (!mine) row = 0;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(args.slab) + (size_t)row * args.row_stride;
    const float* w = reinterpret_cast<const float*>(base);
    const bool vec = (args.row_stride & 15u) == 0;  // 16-byte aligned rows: dwordx4 loads
    const int chunks = dim >> 3, groups = chunks >> 2;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int g = 0; g < groups; ++g) {
        const int e = 32 * g + 8 * a;
        float x[8];
        if (vec) {
            const float4 lo = *reinterpret_cast<const float4*>(w + e), hi = *reinterpret_cast<const float4*>(w + e + 4);
            x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w;
            x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = w[e + j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = x[j] * qs[e + j];
            acc[j] = acc[j] + p;
        }
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float u = acc[j] + quad_xor1(acc[j]);  // lanes 0,1: acc0+acc1   lanes 2,3: acc2+acc3
        v[j] = u + quad_xor2(u);                     // (acc0+acc1)+(acc2+acc3)
    }
    for (int c = 4 * groups; c < chunks; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = w[8 * c + j] * qs[8 * c + j];
            v[j] = v[j] + p;
        }
    float s = hreduce8(v, args.hreduce);
    for (int i = chunks * 8; i < dim; ++i) s = __builtin_fmaf(w[i], qs[i], s);
    if (a != 0 || !in_range) return;
    if (GATHER) {
        if (mine) out_scores[item] = s;
    } else {
        bool valid = true;
        if (args.live) valid = valid && ((args.live[row >> 6] >> (row & 63)) & 1ull);
        if (args.allow) valid = valid && ((args.allow[row >> 6] >> (row & 63)) & 1ull);
        out_packed[item] = valid ? pack(s, args.row_base + row) : kEmpty;
    }
}

// Fused scan + top-k over an F32 slab, NQ queries per pass over the rows (1, 2 or 4): the f32 counterpart of scan_topk_kernel's
// runtime-dimension body (scan_kernels.hip) — same lane mapping (a quad per row, 16 rows per wave tile, grid-stride tiles), same
// threshold-gated wave candidate buffers (one per query) and block fold, the arithmetic of dot_rows_f32_kernel above: a row's
// elements are loaded ONCE and multiplied against every query of the pass (round 5; the r04 kernel was one query per pass — a batch
// of B queries streamed the slab B times).  Queries at args.queries + q * dim; one best-first list of k entries per (query, block) at
// args.partial + (q * gridDim.x + block) * k; merge_topk_kernel finishes.  dim % 8 == 0.
template <int KCAP, int NQ>
__global__ __launch_bounds__(256) void scan_topk_f32_kernel(ScanArgs args) {
    constexpr int CAP = 2 * KCAP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int dim = (int)args.dim;
    float* qs = reinterpret_cast<float*>(smem);                                                      // [NQ][dim]
    u64* bufs = reinterpret_cast<u64*>(smem + (((size_t)NQ * dim * 4 + 15) & ~(size_t)15));         // [wave][NQ][CAP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = lane & 3, r = lane >> 2;
    for (int i = tid; i < NQ * dim; i += 256) qs[i] = args.queries[i];
    __syncthreads();
    WaveTopK<CAP> tk[NQ];
    u64 thr[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        tk[q].init(bufs + ((size_t)wave * NQ + q) * CAP);
        thr[q] = 0;
    }
    const uint32_t nrows = args.nrows;
    const uint32_t ntiles = (nrows + kRowsPerTile - 1) / kRowsPerTile;
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    const unsigned char* slab = reinterpret_cast<const unsigned char*>(args.slab);
    const size_t row_bytes = args.row_stride;
    const bool vec = (args.row_stride & 15u) == 0;
    const int k = (int)args.k;
    const int chunks = dim >> 3, groups = chunks >> 2;
    for (uint32_t tile = blockIdx.x * kWavesPerBlock + wave; tile < ntiles; tile += nwaves) {
        const uint32_t row = tile * kRowsPerTile + r;
        const uint32_t rowc = row < nrows ? row : nrows - 1;
        const float* w = reinterpret_cast<const float*>(slab + (size_t)rowc * row_bytes);
        const uint32_t w64 = (tile * kRowsPerTile) >> 6;
        const u64 live_word = args.live ? args.live[w64] : ~0ull;
        const u64 allow_word = args.allow ? args.allow[w64] : ~0ull;
        float acc[NQ][8];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
#pragma unroll 2
        for (int g = 0; g < groups; ++g) {
            const int e = 32 * g + 8 * a;
            float x[8];
            if (vec) {
                const float4 lo = *reinterpret_cast<const float4*>(w + e), hi = *reinterpret_cast<const float4*>(w + e + 4);
                x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w;
                x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = w[e + j];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float p = x[j] * qs[q * dim + e + j];
                    acc[q][j] = acc[q][j] + p;
                }
        }
        const bool valid = row < nrows && ((live_word >> (row & 63)) & 1ull) && ((allow_word >> (row & 63)) & 1ull);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float u = acc[q][j] + quad_xor1(acc[q][j]);
                v[j] = u + quad_xor2(u);
            }
            for (int c = 4 * groups; c < chunks; ++c)   // leftover chunks join AFTER the combine (simd.rs:581-702)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float p = w[8 * c + j] * qs[q * dim + 8 * c + j];
                    v[j] = v[j] + p;
                }
            const float score = hreduce8(v, args.hreduce);
            const u64 packed = pack(score, args.row_base + row);
            bool cand = valid && a == 0 && sortkey(packed) > thr[q];
            u64 m = __ballot(cand);
            if (m == 0) continue;
            if (tk[q].count + (int)__popcll(m) > CAP) {
                thr[q] = tk[q].compact(k, lane);
                cand = cand && sortkey(packed) > thr[q];
                m = __ballot(cand);
            }
            if (cand) tk[q].buf[tk[q].count + (int)__popcll(m & ((1ull << lane) - 1ull))] = packed;
            tk[q].count += (int)__popcll(m);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) (void)tk[q].compact(k, lane);
    __syncthreads();
    // fold the four waves' best-first lists of a query: elementwise max of A[i] and B[KCAP-1-i], re-sort.  Wave q folds query q
    // (NQ <= 4 = the block's waves); its destination is wave 0's buffer of that query.
    if (wave < NQ) {
        const int q = wave;
        u64* dst = bufs + (size_t)q * CAP;
        for (int w = 1; w < kWavesPerBlock; ++w) {
            const u64* src = bufs + ((size_t)w * NQ + q) * CAP;
            for (int i = lane; i < KCAP; i += 64) {
                const u64 xx = dst[i], yy = src[KCAP - 1 - i];
                dst[i] = sortkey(xx) >= sortkey(yy) ? xx : yy;
            }
            for (int i = KCAP + lane; i < CAP; i += 64) dst[i] = kEmpty;
            wave_sort_desc<CAP>(dst, lane);
        }
        u64* out = args.partial + ((size_t)q * gridDim.x + blockIdx.x) * k;
        for (int i = lane; i < k; i += 64) out[i] = dst[i];
    }
}

template <int KCAP, int NQ>
static hipError_t launch_f32_t(const ScanArgs& args, int grid, hipStream_t stream, int* occupancy) {
    static_assert(NQ <= kWavesPerBlock, "one wave folds one query's lists");
    const size_t lds = (((size_t)NQ * args.dim * 4 + 15) & ~(size_t)15) + (size_t)kWavesPerBlock * NQ * 2 * KCAP * 8;
    auto kern = scan_topk_f32_kernel<KCAP, NQ>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (occupancy) {
        int blocks = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kern, 256, lds) != hipSuccess || blocks < 1) blocks = 1;
        *occupancy = blocks;
        return hipSuccess;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, args);
    return hipGetLastError();
}

// largest number of queries (4, 2 or 1) one pass of the fused F32 kernel takes for this shape: the queries and the per-wave
// candidate buffers must fit the LDS
int scan_f32_queries_per_pass(int dim, int kcap, int want) {
    for (int nq = 4; nq >= 1; nq >>= 1) {
        if (nq > want) continue;
        const size_t lds = (((size_t)nq * dim * 4 + 15) & ~(size_t)15) + (size_t)kWavesPerBlock * nq * 2 * (kcap <= 64 ? 64 : 256) * 8;
        if (lds <= (size_t)144 * 1024) return nq;
    }
    return 1;
}

// kcap 64 or 256 (k <= kcap); nq queries per pass (1, 2 or 4: scan_f32_queries_per_pass); occupancy != null: only reports the
// resident blocks per CU
hipError_t launch_scan_topk_f32(const ScanArgs& args, int kcap, int nq, int grid, hipStream_t stream, int* occupancy) {
    if ((args.dim & 7) || (size_t)nq * args.dim * 4 > (size_t)96 * 1024) return hipErrorInvalidValue;
    if (kcap <= 64) {
        if (nq == 4) return launch_f32_t<64, 4>(args, grid, stream, occupancy);
        if (nq == 2) return launch_f32_t<64, 2>(args, grid, stream, occupancy);
        return nq == 1 ? launch_f32_t<64, 1>(args, grid, stream, occupancy) : hipErrorInvalidValue;
    }
    if (nq == 4) return launch_f32_t<256, 4>(args, grid, stream, occupancy);
    if (nq == 2) return launch_f32_t<256, 2>(args, grid, stream, occupancy);
    return nq == 1 ? launch_f32_t<256, 1>(args, grid, stream, occupancy) : hipErrorInvalidValue;
}

hipError_t launch_score_rows_f32(const ScanArgs& args, u64* out_packed, int q_index, hipStream_t stream) {
    const unsigned blocks = (unsigned)(((uint64_t)args.nrows * 4 + 255) / 256);
    hipLaunchKernelGGL((dot_rows_f32_kernel<false>), dim3(blocks ? blocks : 1), dim3(256), (size_t)args.dim * 4, stream,
                       args, nullptr, args.nrows, out_packed, nullptr, q_index);
    return hipGetLastError();
}

hipError_t launch_gather_dot_f32(const ScanArgs& args, const uint32_t* rows, uint32_t n, float* out, hipStream_t stream) {
    const unsigned blocks = (unsigned)(((uint64_t)n * 4 + 255) / 256);
    hipLaunchKernelGGL((dot_rows_f32_kernel<true>), dim3(blocks ? blocks : 1), dim3(256), (size_t)args.dim * 4, stream,
                       args, rows, n, nullptr, out, 0);
    return hipGetLastError();
}

}  // namespace fsgpu
