// scan_kernels.hip — the f16 cosine brute-force scan fused with a wave-parallel top-k (gfx950).
//
// Replaces, for a device-resident slab, the reference's hot loop
//   scan_parallel -> scan_range_chunk -> dot_product_f16_bytes_f32 -> bounded BinaryHeap ->
//   merge_partial_heaps -> resolve_hits sort
// (crates/frankensearch-index/src/search.rs:1013-1036,1257-1327,1704-1720,1493-1501;
//  crates/frankensearch-index/src/simd.rs:398-446).
//
// Arithmetic contract: the score of a row is computed in the reference's exact operation order —
// 8-lane chunks, chunk c of every group of four accumulated into accumulator (c mod 4) with a
// separate IEEE multiply and add (never an FMA), leftover chunks into accumulator 0, the lane-wise
// (s0+s1)+(s2+s3), the 8-lane horizontal add, and a fused scalar tail for dim % 8 — so GPU scores
// are bit-identical to the CPU path and the ranking needs no re-scoring.
//
// Mapping to CDNA4: FOUR LANES PER ROW.  Lane a (0..3) of a quad owns accumulator s_a, i.e. the
// 16-byte chunks c = 4g+a of its row, so every load is a dwordx4 and the four lanes of a quad
// fetch 64 contiguous bytes; a wave covers a 16-row tile (16 x dim x 2 bytes, contiguous in HBM).
// The (s0+s1)+(s2+s3) step is two DPP quad_perm butterflies (IEEE add is commutative, so every
// lane of the quad ends with the same bits).  Queries sit in LDS (broadcast ds_read_b128).
// Top-k: each wave keeps a threshold-gated candidate buffer in LDS (ballot + mbcnt compaction, no
// atomics), re-sorted by a wave-local bitonic network only when it fills; waves of a block merge
// at the end and one sorted list per block goes to HBM for the final merge kernel.
#pragma clang fp contract(off)

#include <cstring>

#include "scan_common.hpp"

namespace fsgpu {

using namespace scan_detail;

// DIM_CT: compile-time dimension (multiple of 32, <= 512: query and double-buffered tiles live in
//         registers, no LDS traffic in the hot loop) or 0 for the runtime-dimension body (any
//         dim % 8 == 0, incl. leftover chunks; query read from LDS).
// NQ:     queries scored per pass (1, 2 or 4).  A lane is (row r, accumulator a, query b):
//         lane = r*4*NQ + b*4 + a, so a wave covers 16/NQ rows and the NQ lanes that share (r, a)
//         load the same 16 bytes (coalesced by the TA into one fetch) but multiply by different
//         queries.  Per-lane arithmetic is identical for every NQ.
// KCAP:   capacity tier of the per-wave / per-block lists (k <= KCAP); CAP = 2*KCAP.
// qsrc: where the block reads its NQ queries from — args.queries (device memory), or for the one-query latency path the
// kernel's own argument block (scan_topk_kq_kernel below)
template <int DIM_CT, int NQ, int KCAP, bool NT_LOADS>
__device__ __forceinline__ void scan_topk_body(const ScanArgs& args, const float* __restrict__ qsrc) {
    constexpr int CAP = 2 * KCAP;
    constexpr int kRows = kRowsPerTile / NQ;  // rows per wave tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int dim = DIM_CT ? DIM_CT : (int)args.dim;
    float* qs = reinterpret_cast<float*>(smem);                                   // [NQ][dim]
    u64* bufs = reinterpret_cast<u64*>(smem + (((size_t)NQ * dim * 4 + 15) & ~(size_t)15));  // [wave][NQ][CAP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int a = lane & 3;
    const int b = (lane >> 2) & (NQ - 1);
    const int r = lane / (4 * NQ);

    for (int i = tid; i < NQ * dim; i += 256) qs[i] = qsrc[i];
    __syncthreads();

    WaveTopK<CAP> tk[NQ];
#pragma unroll
    for (int x = 0; x < NQ; ++x) tk[x].init(bufs + ((size_t)wave * NQ + x) * CAP);
    u64 thr = 0;  // this lane's query's threshold

    const uint32_t nrows = args.nrows;
    const uint32_t ntiles = (nrows + kRows - 1) / kRows;
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    const uint32_t wave_gid = blockIdx.x * kWavesPerBlock + wave;
    const unsigned char* slab = reinterpret_cast<const unsigned char*>(args.slab);
    const size_t row_bytes = args.row_stride;
    const int k = (int)args.k;
    const int hreduce = args.hreduce;
    // lanes with a == 0 carry the candidates; those of query x sit at bit positions 4*NQ*r + 4*x
    constexpr u64 kOwnerMask = NQ == 1 ? 0x1111111111111111ull : (NQ == 2 ? 0x0101010101010101ull : 0x0001000100010001ull);
    const float* qlane = qs + b * dim;

    // Turns the lane's 8 partial sums into the row score and pushes candidates.
    auto finish_tile = [&](uint32_t tile, float (&acc)[8], u64 live_word, u64 allow_word) {
        const uint32_t row = tile * kRows + r;
        bool valid = row < nrows;
        valid = valid && ((live_word >> (row & 63)) & 1ull) && ((allow_word >> (row & 63)) & 1ull);
        float score = quad_finish(acc, hreduce);
        if (DIM_CT == 0 && (dim & 7)) {  // scalar tail: fused mul_add, in index order (unused: host routes
                                         // unaligned dims to the general path)
            const uint32_t rowc = row < nrows ? row : nrows - 1;
            const _Float16* hp = reinterpret_cast<const _Float16*>(slab + (size_t)rowc * row_bytes);
            for (int i = dim & ~7; i < dim; ++i) score = __builtin_fmaf((float)hp[i], qlane[i], score);
        }
        const u64 packed = pack(score, args.row_base + row);
        bool cand = valid && (a == 0) && sortkey(packed) > thr;
        u64 m = __ballot(cand);
        if (m == 0) return;
#pragma unroll
        for (int x = 0; x < NQ; ++x) {
            const u64 xmask = kOwnerMask << (4 * x);
            u64 mx = m & xmask;
            if (mx == 0) continue;
            if (tk[x].count + (int)__popcll(mx) > CAP) {
                const u64 t = tk[x].compact(k, lane);
                if (b == x) thr = t;
                cand = cand && sortkey(packed) > thr;
                mx = __ballot(cand) & xmask;
            }
            if (cand && b == x) {
                const int pos = tk[x].count + (int)__popcll(mx & ((1ull << lane) - 1ull));
                tk[x].buf[pos] = packed;
            }
            tk[x].count += (int)__popcll(mx);
        }
    };

    auto tile_words = [&](uint32_t tile, u64& live_word, u64& allow_word) {
        const uint32_t w64 = (tile * kRows) >> 6;
        live_word = args.live ? args.live[w64] : ~0ull;
        allow_word = args.allow ? args.allow[w64] : ~0ull;
    };

    if constexpr (DIM_CT != 0) {
        constexpr int G = DIM_CT / 32;
        // The lane's slice of its query lives in registers for the whole kernel (G*8 VGPRs).
        float4 q0[G], q1[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4* qp = reinterpret_cast<const float4*>(qlane + 32 * g + 8 * a);
            q0[g] = qp[0];
            q1[g] = qp[1];
        }
        auto load_tile = [&](uint32_t tile, u32x4 (&w)[G]) {
            uint32_t row = tile * kRows + r;
            row = row < nrows ? row : nrows - 1;
            const u32x4* p = reinterpret_cast<const u32x4*>(slab + (size_t)row * row_bytes) + a;
#pragma unroll
            for (int g = 0; g < G; ++g) w[g] = NT_LOADS ? load_nt16(p + 4 * g) : p[4 * g];
        };
        auto compute_tile = [&](uint32_t tile, const u32x4 (&w)[G], u64 live_word, u64 allow_word) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) chunk_mac(acc, w[g], q0[g], q1[g]);
            finish_tile(tile, acc, live_word, allow_word);
        };
        constexpr bool kDoubleBuffer = G <= 12;
        if constexpr (kDoubleBuffer) {
            u32x4 wa[G], wb[G];
            u64 la = ~0ull, aa = ~0ull, lb = ~0ull, ab = ~0ull;
            uint32_t tile = wave_gid;
            if (tile < ntiles) {
                load_tile(tile, wa);
                tile_words(tile, la, aa);
            }
            while (tile < ntiles) {
                uint32_t next = tile + nwaves;
                if (next < ntiles) {
                    load_tile(next, wb);
                    tile_words(next, lb, ab);
                }
                compute_tile(tile, wa, la, aa);
                tile = next;
                if (tile >= ntiles) break;
                next = tile + nwaves;
                if (next < ntiles) {
                    load_tile(next, wa);
                    tile_words(next, la, aa);
                }
                compute_tile(tile, wb, lb, ab);
                tile = next;
            }
        } else {
            for (uint32_t tile = wave_gid; tile < ntiles; tile += nwaves) {
                u32x4 w[G];
                u64 lw, aw;
                load_tile(tile, w);
                tile_words(tile, lw, aw);
                compute_tile(tile, w, lw, aw);
            }
        }
    } else {
        // Runtime dimension (dim % 8 == 0 for the 16-byte loads).
        const int chunks = dim >> 3;
        const int groups = chunks >> 2;
        const int leftover = chunks & 3;
        for (uint32_t tile = wave_gid; tile < ntiles; tile += nwaves) {
            uint32_t row = tile * kRows + r;
            row = row < nrows ? row : nrows - 1;
            const u32x4* p = reinterpret_cast<const u32x4*>(slab + (size_t)row * row_bytes);
            u64 lw, aw;
            tile_words(tile, lw, aw);
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 4
            for (int g = 0; g < groups; ++g) {
                const u32x4 w = load_nt16(p + 4 * g + a);
                const float4* qp = reinterpret_cast<const float4*>(qlane + 32 * g + 8 * a);
                chunk_mac(acc, w, qp[0], qp[1]);
            }
            if (a == 0) {  // leftover chunks all accumulate into s0, in order
                for (int c = 4 * groups; c < 4 * groups + leftover; ++c) {
                    const u32x4 w = load_nt16(p + c);
                    const float4* qp = reinterpret_cast<const float4*>(qlane + 8 * c);
                    chunk_mac(acc, w, qp[0], qp[1]);
                }
            }
            finish_tile(tile, acc, lw, aw);
        }
    }

    // ---- block merge: every wave sorts its lists, then wave x folds the four lists of query x ----
#pragma unroll
    for (int x = 0; x < NQ; ++x) (void)tk[x].compact(k, lane);
    __syncthreads();
    for (int x = wave; x < NQ; x += kWavesPerBlock) {
        u64* dst = bufs + ((size_t)0 * NQ + x) * CAP;
        if (kWavesPerBlock * k <= CAP) {
            // small ranks (k <= CAP / 4): the four sorted lists' first k entries side by side in wave 0's buffer and ONE sort of
            // the next power of two (k = 10: 64 entries, 21 compare-exchange steps) instead of three folds of 28 steps each over
            // CAP entries — this tail is on the critical path of every block, and every block ends at the same time
            const int n = kWavesPerBlock * k;
            int np2 = 64;
            while (np2 < n) np2 <<= 1;
            for (int i = k + lane; i < np2; i += 64) {
                u64 v = kEmpty;
                if (i < n) {
                    const int w = i / k;
                    v = bufs[((size_t)w * NQ + x) * CAP + (i - w * k)];
                }
                dst[i] = v;
            }
            wave_sort_desc_rt(dst, np2, lane);
            u64* out = args.partial + ((size_t)x * gridDim.x + blockIdx.x) * k;
            for (int i = lane; i < k; i += 64) out[i] = dst[i];
            continue;
        }
        for (int w = 1; w < kWavesPerBlock; ++w) {
            const u64* src = bufs + ((size_t)w * NQ + x) * CAP;
            // top-KCAP of two best-first lists: elementwise max of A[i] and B[KCAP-1-i], then re-sort
            for (int i = lane; i < KCAP; i += 64) {
                const u64 xx = dst[i], yy = src[KCAP - 1 - i];
                dst[i] = sortkey(xx) >= sortkey(yy) ? xx : yy;
            }
            for (int i = KCAP + lane; i < CAP; i += 64) dst[i] = kEmpty;
            wave_sort_desc<CAP>(dst, lane);
        }
        // one best-first list of k entries per (query, block)
        u64* out = args.partial + ((size_t)x * gridDim.x + blockIdx.x) * k;
        for (int i = lane; i < k; i += 64) out[i] = dst[i];
    }
}

template <int DIM_CT, int NQ, int KCAP, bool NT_LOADS = false>
__global__ __launch_bounds__(256) void scan_topk_kernel(ScanArgs args) {
    scan_topk_body<DIM_CT, NQ, KCAP, NT_LOADS>(args, args.queries);
}

// One query, carried in the kernel's argument block (the dispatch packet's kernarg segment): the latency path of a lone caller
// needs no H2D copy in front of the scan — the launch itself delivers the 4 dim bytes.  dim <= kKernargQueryDims.
struct KernargQuery {
    float q[kKernargQueryDims];
};
template <int DIM_CT, int KCAP>
__global__ __launch_bounds__(256) void scan_topk_kq_kernel(ScanArgs args, KernargQuery kq) {
    scan_topk_body<DIM_CT, 1, KCAP, false>(args, kq.q);
}

// Thread-per-row fallback for dimensions that are not a multiple of 8 (rows are not 16-byte aligned).
// Same arithmetic order; writes one packed entry per row (kEmpty for dead rows) for the general path.
__global__ __launch_bounds__(256) void score_rows_generic_kernel(ScanArgs args, u64* out_packed, int q_index) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= args.nrows) return;
    const int dim = (int)args.dim;
    bool valid = true;
    if (args.live) valid = valid && ((args.live[row >> 6] >> (row & 63)) & 1ull);
    if (args.allow) valid = valid && ((args.allow[row >> 6] >> (row & 63)) & 1ull);
    if (!valid) {
        out_packed[row] = kEmpty;
        return;
    }
    const _Float16* hp = reinterpret_cast<const _Float16*>(reinterpret_cast<const unsigned char*>(args.slab) + (size_t)row * args.row_stride);
    const float* q = args.queries + (size_t)q_index * dim;
    float s[4][8];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[x][j] = 0.f;
    const int chunks = dim >> 3;
    int c = 0;
    for (; c + 4 <= chunks; c += 4) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float p = (float)hp[(c + x) * 8 + j] * q[(c + x) * 8 + j];
                s[x][j] = s[x][j] + p;
            }
    }
    for (; c < chunks; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = (float)hp[c * 8 + j] * q[c * 8 + j];
            s[0][j] = s[0][j] + p;
        }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (s[0][j] + s[1][j]) + (s[2][j] + s[3][j]);
    float result = hreduce8(v, args.hreduce);
    for (int i = chunks * 8; i < dim; ++i) result = __builtin_fmaf((float)hp[i], q[i], result);
    out_packed[row] = pack(result, args.row_base + row);
}

// Four-lanes-per-row scorer that writes every row's packed entry (general path: k beyond the fused
// tiers, collect-all).  dim % 8 == 0.
__global__ __launch_bounds__(256) void score_rows_kernel(ScanArgs args, u64* out_packed, int q_index) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* qs = reinterpret_cast<float*>(smem);
    const int dim = (int)args.dim;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, a = lane & 3, r = lane >> 2;
    for (int i = tid; i < dim; i += 256) qs[i] = args.queries[(size_t)q_index * dim + i];
    __syncthreads();
    const uint32_t nrows = args.nrows;
    const uint32_t ntiles = (nrows + kRowsPerTile - 1) / kRowsPerTile;
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    const unsigned char* slab = reinterpret_cast<const unsigned char*>(args.slab);
    const size_t row_bytes = args.row_stride;
    const int chunks = dim >> 3, groups = chunks >> 2, leftover = chunks & 3;
    for (uint32_t tile = blockIdx.x * kWavesPerBlock + wave; tile < ntiles; tile += nwaves) {
        const uint32_t row = tile * kRowsPerTile + r;
        const uint32_t rowc = row < nrows ? row : nrows - 1;
        const u32x4* p = reinterpret_cast<const u32x4*>(slab + (size_t)rowc * row_bytes);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 4
        for (int g = 0; g < groups; ++g) {
            const u32x4 w = load_nt16(p + 4 * g + a);
            const float4* qp = reinterpret_cast<const float4*>(qs + 32 * g + 8 * a);
            chunk_mac(acc, w, qp[0], qp[1]);
        }
        if (a == 0) {
            for (int c = 4 * groups; c < 4 * groups + leftover; ++c) {
                const u32x4 w = load_nt16(p + c);
                const float4* qp = reinterpret_cast<const float4*>(qs + 8 * c);
                chunk_mac(acc, w, qp[0], qp[1]);
            }
        }
        const float s = quad_finish(acc, args.hreduce);
        if (a == 0 && row < nrows) {
            bool valid = true;
            if (args.live) valid = valid && ((args.live[row >> 6] >> (row & 63)) & 1ull);
            if (args.allow) valid = valid && ((args.allow[row >> 6] >> (row & 63)) & 1ull);
            out_packed[row] = valid ? pack(s, args.row_base + row) : kEmpty;
        }
    }
}

// Final merge: one block per query folds nlists best-first lists of list_len entries into the top-k and
// emits (row, score) arrays, best first (merge_partial_heaps + the resolve_hits sort, search.rs:1704-1720,
// 1493-1501).  Selection is by unique integer sortkeys, so the LDS append order cannot change the result.
// Latency-shaped for the usual case (nlists*list_len <= 8192 entries, e.g. 512 block lists x k=10):
//   1. every thread fetches its <= 8 entries with independent loads (ONE memory round trip);
//   2. two lower bounds on the global k-th best prune them: the best k-th entry of any list ("tails") and
//      the k-th largest list head ("heads": the k largest heads are k distinct entries);
//   3. survivors (typically a few multiples of k) are appended wave-aggregated and sorted by one wave.
// Larger inputs take a threshold-tightening streaming pass with whole-block sorts.
template <int MCAP, int NT>
__global__ __launch_bounds__(NT) void merge_topk_kernel(MergeArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char merge_smem[];
    u64* buf = reinterpret_cast<u64*>(merge_smem);
    __shared__ int s_count;
    __shared__ u64 s_thr;
    __shared__ int s_rank[2048];
    __shared__ u64 s_gmax[64];
    constexpr int HCAP = 2048;
    constexpr int PER = MCAP / NT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = blockIdx.x;
    const u64* in = args.lists + (size_t)q * args.q_stride;
    const uint32_t list_len = args.list_len;
    const uint32_t nlists = args.nlists;
    const size_t total = (size_t)nlists * list_len;
    const uint32_t total32 = (uint32_t)total;
    const int k = (int)args.k;
    if (tid == 0) {
        s_count = 0;
        s_thr = 0;
    }
    int cnt;
    if (total <= (size_t)MCAP) {
        u64* hkeys = buf + (MCAP - HCAP);  // aliased: dead before survivors are appended
        const bool sorted_lists = args.lists_sorted != 0;
        const bool use_heads = sorted_lists && nlists >= (uint32_t)k && nlists <= (uint32_t)HCAP;
        for (int j = tid; j < HCAP; j += NT) s_rank[j] = 0;
        u64 e[PER];
        uint32_t pos[PER], lst[PER];
#pragma unroll
        for (int x = 0; x < PER; ++x) {
            const uint32_t i = tid + x * NT;
            e[x] = kEmpty;
            lst[x] = i / list_len;
            pos[x] = i - lst[x] * list_len;
            if (i < total32) e[x] = in[(size_t)lst[x] * args.l_stride + pos[x]];
        }
        __syncthreads();
        u64 best_tail = 0;
#pragma unroll
        for (int x = 0; x < PER; ++x) {
            const bool real = e[x] != kEmpty;
            const u64 key = real ? sortkey(e[x]) : 0ull;
            if (sorted_lists && real && pos[x] == (uint32_t)(k - 1)) best_tail = key > best_tail ? key : best_tail;
            if (use_heads && pos[x] == 0 && tid + x * NT < total32) hkeys[lst[x]] = key;
        }
        // wave max of best_tail, then one LDS atomic per wave
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u64 other = __shfl_xor(best_tail, off);
            best_tail = other > best_tail ? other : best_tail;
        }
        if (lane == 0 && best_tail) atomicMax(&s_thr, best_tail);
        __syncthreads();
        // Small ranks over at most NT lists (a lone query's exact scan: ~1,000 block lists x k = 10): the EXACT k-th largest head
        // below costs every thread a pass over all heads (~7 us of this kernel's 25); any k distinct entries bound the k-th best
        // from below, so the k-th largest of the maxima of the heads' 16-lane groups serves — a dozen more survivors for the final
        // one-wave sort, three barriers instead of the ranking.
        const uint32_t ngroups = (nlists + 15) / 16;
        const bool quick_heads = use_heads && nlists <= (uint32_t)NT && ngroups <= 64 && (uint32_t)k <= ngroups;
        if (quick_heads) {
            u64 h = (uint32_t)tid < nlists ? hkeys[tid] : 0ull;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const u64 o = __shfl_xor(h, off);
                h = o > h ? o : h;
            }
            if ((tid & 15) == 0 && (uint32_t)(tid >> 4) < ngroups) s_gmax[tid >> 4] = h;
            __syncthreads();
            if (tid < 64) {
                const u64 mine = (uint32_t)tid < ngroups ? s_gmax[tid] : 0ull;
                int greater = 0;
                for (uint32_t j = 0; j < ngroups; ++j) greater += s_gmax[j] > mine ? 1 : 0;
                if (mine != 0 && greater == k - 1) atomicMax(&s_thr, mine);   // keys are unique: at most one group has this rank
            }
            __syncthreads();
        } else if (use_heads) {
            // rank heads: P threads share one head, each scanning a slice of the head array (broadcast reads)
            uint32_t np2 = 1;
            while (np2 < nlists) np2 <<= 1;
            const uint32_t P = NT / np2 > 0 ? NT / np2 : 1;           // threads per head (NT, np2 powers of two)
            const uint32_t heads_per_round = NT / P;
            for (uint32_t h0 = 0; h0 < nlists; h0 += heads_per_round) {
                const uint32_t h = h0 + tid / P;
                const uint32_t part = tid % P;
                if (h < nlists) {
                    const u64 mine = hkeys[h];
                    const uint32_t span = (nlists + P - 1) / P;
                    const uint32_t j0 = part * span, j1 = j0 + span < nlists ? j0 + span : nlists;
                    int greater = 0;
#pragma unroll 4
                    for (uint32_t j = j0; j < j1; ++j) greater += hkeys[j] > mine ? 1 : 0;
                    if (greater) atomicAdd(&s_rank[h], greater);
                }
            }
            __syncthreads();
            for (uint32_t h = tid; h < nlists; h += NT)
                if (hkeys[h] != 0 && s_rank[h] == k - 1) atomicMax(&s_thr, hkeys[h]);  // k-th largest head
            __syncthreads();
        }
        const u64 thr0 = s_thr;
#pragma unroll
        for (int x = 0; x < PER; ++x) {
            const bool ok = e[x] != kEmpty && sortkey(e[x]) >= thr0;
            const u64 m = __ballot(ok);
            if (m) {
                int wbase = 0;
                if (lane == 0) wbase = atomicAdd(&s_count, (int)__popcll(m));
                wbase = __shfl(wbase, 0);
                if (ok) buf[wbase + (int)__popcll(m & ((1ull << lane) - 1ull))] = e[x];
            }
        }
        __syncthreads();
        cnt = s_count;
    } else {
        // More entries than one register pass holds (many block lists x a large k).  For best-first lists the best k-th
        // entry of any list is a lower bound of the global k-th best, which prunes almost everything up front; the
        // stream then appends the survivors, four entries per thread between barriers, and only re-sorts (tightening
        // the threshold) if the buffer nearly fills.
        __syncthreads();
        if (args.lists_sorted) {
            // the same two bounds as the one-pass path, from a pre-pass over list heads and k-th entries only
            u64* hkeys = buf + (MCAP - HCAP);  // dead before the survivors reach that part of the buffer
            const bool use_heads = nlists >= (uint32_t)k && nlists <= (uint32_t)HCAP;
            for (int j = tid; j < HCAP; j += NT) s_rank[j] = 0;
            u64 best = 0;
            for (uint32_t l = tid; l < nlists; l += NT) {
                if (list_len >= (uint32_t)k) {
                    const u64 c = in[(size_t)l * args.l_stride + (k - 1)];
                    if (c != kEmpty) {
                        const u64 key = sortkey(c);
                        best = key > best ? key : best;
                    }
                }
                if (use_heads) {
                    const u64 c0 = in[(size_t)l * args.l_stride];
                    hkeys[l] = c0 != kEmpty ? sortkey(c0) : 0ull;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const u64 other = __shfl_xor(best, off);
                best = other > best ? other : best;
            }
            if (lane == 0 && best) atomicMax(&s_thr, best);
            __syncthreads();
            if (use_heads) {
                uint32_t np2 = 1;
                while (np2 < nlists) np2 <<= 1;
                const uint32_t P = NT / np2 > 0 ? NT / np2 : 1;
                const uint32_t heads_per_round = NT / P;
                for (uint32_t h0 = 0; h0 < nlists; h0 += heads_per_round) {
                    const uint32_t h = h0 + tid / P;
                    const uint32_t part = tid % P;
                    if (h < nlists) {
                        const u64 mine = hkeys[h];
                        const uint32_t span = (nlists + P - 1) / P;
                        const uint32_t j0 = part * span, j1 = j0 + span < nlists ? j0 + span : nlists;
                        int greater = 0;
#pragma unroll 4
                        for (uint32_t j = j0; j < j1; ++j) greater += hkeys[j] > mine ? 1 : 0;
                        if (greater) atomicAdd(&s_rank[h], greater);
                    }
                }
                __syncthreads();
                for (uint32_t h = tid; h < nlists; h += NT)
                    if (hkeys[h] != 0 && s_rank[h] == k - 1) atomicMax(&s_thr, hkeys[h]);  // k-th largest head
                __syncthreads();
            }
            if (tid == 0 && s_thr) s_thr -= 1;  // the loop keeps keys > s_thr: the bound entry itself stays
            __syncthreads();
        }
        constexpr int U = 4;
        for (uint32_t base = 0; base < total32; base += NT * U) {
            u64 c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = base + u * NT + tid;
                c[u] = kEmpty;
                if (i < total32) {
                    const uint32_t l = i / list_len;
                    c[u] = in[(size_t)l * args.l_stride + (i - l * list_len)];
                }
            }
            const u64 thr = s_thr;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c[u] != kEmpty && sortkey(c[u]) > thr) {
                    const int pos1 = atomicAdd(&s_count, 1);
                    if (pos1 < MCAP) buf[pos1] = c[u];
                }
            __syncthreads();
            if (s_count > MCAP - NT * U) {  // block-uniform
                const int have = s_count < MCAP ? s_count : MCAP;
                for (int j = have + tid; j < MCAP; j += NT) buf[j] = kEmpty;
                block_sort_desc<MCAP, NT>(buf, tid);
                if (tid == 0) {
                    s_count = have < k ? have : k;
                    if (have >= k) s_thr = sortkey(buf[k - 1]);
                }
                __syncthreads();
            }
        }
        cnt = s_count;
    }
    // sort the survivors
    int np2s = 64;
    while (np2s < cnt) np2s <<= 1;
    for (int j = cnt + tid; j < np2s; j += NT) buf[j] = kEmpty;
    __syncthreads();
    if (np2s <= 512) {
        if (tid < 64) wave_sort_desc_rt(buf, np2s, lane);
        __syncthreads();
    } else {
        block_sort_desc_rt<NT>(buf, np2s, tid);
    }
    const int n = cnt < k ? cnt : k;  // every survivor is a real entry
    for (int j = tid; j < (int)args.out_stride; j += NT) {
        const u64 c = j < n ? buf[j] : kEmpty;
        if (args.out_rows) args.out_rows[(size_t)q * args.out_stride + j] = (uint32_t)c;
        if (args.out_scores) args.out_scores[(size_t)q * args.out_stride + j] = __uint_as_float((uint32_t)(c >> 32));
        if (args.out_packed) args.out_packed[(size_t)q * args.out_stride + j] = c;
    }
    if (tid == 0 && args.out_counts) args.out_counts[q] = (uint32_t)n;
}

// packed -> sortkey (in place), for the general path's radix sort; and the inverse for rows.
__global__ void packed_to_sortkey_kernel(u64* data, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const u64 c = data[i];
        data[i] = c == kEmpty ? 0ull : sortkey(c);
    }
}

// General-path epilogue: first k sorted sortkeys -> rows (+count of real entries) (+ the packed entries themselves: the
// inverse of sortkey() — exact for every non-NaN score, which is all an integer pass-1 score can be).
__global__ void sorted_keys_to_rows_kernel(const u64* keys, uint32_t k, uint32_t* out_rows, uint32_t* out_count, u64* out_packed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) {
        const u64 key = keys[i];
        out_rows[i] = key ? ~(uint32_t)key : 0xffffffffu;
        if (out_packed) {
            const uint32_t ord = (uint32_t)(key >> 32);
            const uint32_t bits = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
            out_packed[i] = key ? (((u64)bits << 32) | (uint32_t)~(uint32_t)key) : kEmpty;
        }
        // count = number of non-zero keys among the first k (zeros sort last)
        if (key != 0 && (i + 1 == k || keys[i + 1] == 0)) *out_count = i + 1;
        if (i == 0 && key == 0) *out_count = 0;
    }
}

// gather-dot: VectorIndex::dot_query_at (lib.rs:3229-3239) over a row list; one quad per row.
// rows are global ids; out-of-shard rows are skipped (left untouched).
__global__ __launch_bounds__(256) void gather_dot_kernel(ScanArgs args, const uint32_t* rows, uint32_t n, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* qs = reinterpret_cast<float*>(smem);
    const int dim = (int)args.dim;
    const int tid = threadIdx.x, lane = tid & 63, a = lane & 3;
    for (int i = tid; i < dim; i += 256) qs[i] = args.queries[i];
    __syncthreads();
    const uint32_t item = (blockIdx.x * 256 + tid) >> 2;
    const bool in_range = item < n;
    uint32_t row = in_range ? rows[item] - args.row_base : 0;  // wraps for rows below the shard base
    const bool mine = in_range && row < args.nrows;
    if (!mine) row = 0;
    const unsigned char* slab = reinterpret_cast<const unsigned char*>(args.slab);
    const int chunks = dim >> 3, groups = chunks >> 2, leftover = chunks & 3;
    float s;
    if ((dim & 7) == 0) {
        const u32x4* p = reinterpret_cast<const u32x4*>(slab + (size_t)row * args.row_stride);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int g = 0; g < groups; ++g) {
            const u32x4 w = p[4 * g + a];
            const float4* qp = reinterpret_cast<const float4*>(qs + 32 * g + 8 * a);
            chunk_mac(acc, w, qp[0], qp[1]);
        }
        if (a == 0) {
            for (int c = 4 * groups; c < 4 * groups + leftover; ++c) {
                const u32x4 w = p[c];
                const float4* qp = reinterpret_cast<const float4*>(qs + 8 * c);
                chunk_mac(acc, w, qp[0], qp[1]);
            }
        }
        s = quad_finish(acc, args.hreduce);
    } else {
        // unaligned rows: lane a still owns accumulator a, element loads
        const _Float16* hp = reinterpret_cast<const _Float16*>(slab + (size_t)row * args.row_stride);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int g = 0; g < groups; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 32 * g + 8 * a + j;
                const float p = (float)hp[e] * qs[e];
                acc[j] = acc[j] + p;
            }
        if (a == 0)
            for (int c = 4 * groups; c < 4 * groups + leftover; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float p = (float)hp[8 * c + j] * qs[8 * c + j];
                    acc[j] = acc[j] + p;
                }
        s = quad_finish(acc, args.hreduce);
        for (int i = chunks * 8; i < dim; ++i) s = __builtin_fmaf((float)hp[i], qs[i], s);
    }
    if (mine && a == 0) out[item] = s;
}

// gather-dot for MANY queries in one launch: item i = (row rows[i], query qidx[i]) — quality_scores_for_hits of a whole chunk of the
// two-tier flow (two_tier.rs:1566-1631 per query; a chunk is ~1,024 x 30 items).  A quad per item, the operation order of
// gather_dot_kernel (same bits); the quad reads its query's chunks from global memory (a query's 1.5 KB is shared by its ~30 items
// and stays in the L1 / L2), so consecutive items need not share a query.  dim % 8 == 0, f16 rows.
__global__ __launch_bounds__(256) void gather_dot_mq_kernel(ScanArgs args, const uint32_t* __restrict__ rows, const uint32_t* __restrict__ qidx,
                                                            uint32_t n, float* __restrict__ out) {
    const int dim = (int)args.dim;
    const int tid = threadIdx.x, a = tid & 3;
    const uint32_t item = (blockIdx.x * 256 + tid) >> 2;
    const bool in_range = item < n;
    uint32_t row = in_range ? rows[item] - args.row_base : 0;
    const bool mine = in_range && row < args.nrows;
    if (!mine) row = 0;
    const float* qs = args.queries + (size_t)(in_range ? qidx[item] : 0) * dim;
    const unsigned char* slab = reinterpret_cast<const unsigned char*>(args.slab);
    const int chunks = dim >> 3, groups = chunks >> 2, leftover = chunks & 3;
    const u32x4* p = reinterpret_cast<const u32x4*>(slab + (size_t)row * args.row_stride);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int g = 0; g < groups; ++g) {
        const u32x4 w = p[4 * g + a];
        const float4* qp = reinterpret_cast<const float4*>(qs + 32 * g + 8 * a);
        chunk_mac(acc, w, qp[0], qp[1]);
    }
    if (a == 0) {
        for (int c = 4 * groups; c < 4 * groups + leftover; ++c) {
            const u32x4 w = p[c];
            const float4* qp = reinterpret_cast<const float4*>(qs + 8 * c);
            chunk_mac(acc, w, qp[0], qp[1]);
        }
    }
    const float s = quad_finish(acc, args.hreduce);
    if (mine && a == 0) out[item] = s;
}

// f32 -> f16 round-to-nearest-even (encode_f32_to_f16_extend, simd.rs:2245-2305): v_cvt_f16_f32 is RNE.
__global__ void encode_f16_kernel(const float* src, size_t n, unsigned short* dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const _Float16 h = (_Float16)src[i];
        dst[i] = __builtin_bit_cast(unsigned short, h);
    }
}

// Same conversion for whole rows taken in a given order (FSVI writer: rows land in (hash, doc_id) order).
__global__ void encode_rows_f16_kernel(const float* __restrict__ src, const uint32_t* __restrict__ perm, uint64_t n, uint32_t dim,
                                       unsigned short* __restrict__ dst) {
    const uint64_t row = blockIdx.x;
    if (row >= n) return;
    const float* s = src + (size_t)perm[row] * dim;
    for (uint32_t d = threadIdx.x; d < dim; d += blockDim.x) {
        const _Float16 h = (_Float16)s[d];
        dst[(size_t)row * dim + d] = __builtin_bit_cast(unsigned short, h);
    }
}

// f16 -> f32 widen (widen8_f16_lanes, simd.rs:63-82): exact.
__global__ void widen_f16_kernel(const unsigned short* src, size_t n, float* dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)__builtin_bit_cast(_Float16, src[i]);
}

// ---- host-side launchers ---------------------------------------------------------------------------

size_t scan_lds_bytes(int dim, int nq, int kcap) {
    return (((size_t)nq * dim * 4 + 15) & ~(size_t)15) + (size_t)kWavesPerBlock * nq * (2 * kcap) * 8;
}

template <int DIM_CT, int NQ, int KCAP, bool NT_LOADS = false>
static hipError_t launch_scan_t(const ScanArgs& args, int grid, hipStream_t stream) {
    const size_t lds = scan_lds_bytes(DIM_CT ? DIM_CT : (int)args.dim, NQ, KCAP);
    auto kern = scan_topk_kernel<DIM_CT, NQ, KCAP, NT_LOADS>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, args);
    return hipGetLastError();
}

template <int NQ, int KCAP>
static hipError_t launch_scan_dim(const ScanArgs& args, int grid, hipStream_t stream, bool force_runtime_dim,
                                  bool plain_loads /* variant 2: nontemporal loads (A/B only) */) {
    if (plain_loads && args.dim == 384) return launch_scan_t<384, NQ, KCAP, true>(args, grid, stream);  // A/B: nt loads
    if (!force_runtime_dim) {
        switch (args.dim) {
            case 128: return launch_scan_t<128, NQ, KCAP>(args, grid, stream);
            case 256: return launch_scan_t<256, NQ, KCAP>(args, grid, stream);
            case 384: return launch_scan_t<384, NQ, KCAP>(args, grid, stream);
            case 512: return launch_scan_t<512, NQ, KCAP>(args, grid, stream);
            default: break;
        }
    }
    return launch_scan_t<0, NQ, KCAP>(args, grid, stream);
}

hipError_t launch_scan_topk(const ScanArgs& args, int nq, int kcap, int grid, hipStream_t stream,
                            bool force_runtime_dim, bool plain_loads) {
#define FSGPU_DISPATCH(NQ_, KCAP_) \
    if (nq == NQ_ && kcap == KCAP_)  \
        return launch_scan_dim<NQ_, KCAP_>(args, grid, stream, force_runtime_dim, plain_loads);
    FSGPU_DISPATCH(1, 64)
    FSGPU_DISPATCH(2, 64)
    FSGPU_DISPATCH(4, 64)
    FSGPU_DISPATCH(1, 256)
    FSGPU_DISPATCH(2, 256)
    FSGPU_DISPATCH(4, 256)
#undef FSGPU_DISPATCH
    return hipErrorInvalidValue;
}

template <int DIM_CT, int KCAP>
static hipError_t launch_scan_kq_t(const ScanArgs& args, const KernargQuery& kq, int grid, hipStream_t stream) {
    const size_t lds = scan_lds_bytes(DIM_CT ? DIM_CT : (int)args.dim, 1, KCAP);
    auto kern = scan_topk_kq_kernel<DIM_CT, KCAP>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, args, kq);
    return hipGetLastError();
}

bool scan_kernarg_query_supported(int dim, int kcap) { return dim % 8 == 0 && dim <= kKernargQueryDims && (kcap == 64 || kcap == 256); }

// one query from HOST memory (copied into the launch's argument block); same kernel body, grid and lists as launch_scan_topk(nq = 1)
hipError_t launch_scan_topk_host_query(const ScanArgs& args, const float* query_host, int kcap, int grid, hipStream_t stream) {
    if (!scan_kernarg_query_supported((int)args.dim, kcap)) return hipErrorInvalidValue;
    KernargQuery kq;
    std::memcpy(kq.q, query_host, (size_t)args.dim * 4);
#define FSGPU_KQ(KCAP_)                                                                      \
    if (kcap == KCAP_) {                                                                     \
        switch (args.dim) {                                                                  \
            case 128: return launch_scan_kq_t<128, KCAP_>(args, kq, grid, stream);           \
            case 256: return launch_scan_kq_t<256, KCAP_>(args, kq, grid, stream);           \
            case 384: return launch_scan_kq_t<384, KCAP_>(args, kq, grid, stream);           \
            case 512: return launch_scan_kq_t<512, KCAP_>(args, kq, grid, stream);           \
            default: return launch_scan_kq_t<0, KCAP_>(args, kq, grid, stream);              \
        }                                                                                    \
    }
    FSGPU_KQ(64)
    FSGPU_KQ(256)
#undef FSGPU_KQ
    return hipErrorInvalidValue;
}

template <int DIM_CT, int NQ, int KCAP>
static int occupancy_t(int dim) {
    int blocks = 0;
    const size_t lds = scan_lds_bytes(DIM_CT ? DIM_CT : dim, NQ, KCAP);
    auto kern = scan_topk_kernel<DIM_CT, NQ, KCAP>;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kern, 256, lds) != hipSuccess || blocks < 1) blocks = 1;
    return blocks;
}

template <int NQ, int KCAP>
static int occupancy_dim(int dim, bool force_runtime_dim) {
    if (!force_runtime_dim) {
        switch (dim) {
            case 128: return occupancy_t<128, NQ, KCAP>(dim);
            case 256: return occupancy_t<256, NQ, KCAP>(dim);
            case 384: return occupancy_t<384, NQ, KCAP>(dim);
            case 512: return occupancy_t<512, NQ, KCAP>(dim);
            default: break;
        }
    }
    return occupancy_t<0, NQ, KCAP>(dim);
}

// Resident 256-thread blocks per CU for the instantiation launch_scan_topk would pick: the grid is
// sized to exactly one resident wave of blocks so the strided tile walk has no tail.
int scan_occupancy_blocks_per_cu(int dim, int nq, int kcap, bool force_runtime_dim) {
#define FSGPU_OCC(NQ_, KCAP_) \
    if (nq == NQ_ && kcap == KCAP_) return occupancy_dim<NQ_, KCAP_>(dim, force_runtime_dim);
    FSGPU_OCC(1, 64)
    FSGPU_OCC(2, 64)
    FSGPU_OCC(4, 64)
    FSGPU_OCC(1, 256)
    FSGPU_OCC(2, 256)
    FSGPU_OCC(4, 256)
#undef FSGPU_OCC
    return 1;
}

template <int MCAP>
static hipError_t launch_merge_t(const MergeArgs& args, int nq, hipStream_t stream) {
    constexpr int NT = 1024;
    auto kern = merge_topk_kernel<MCAP, NT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, MCAP * 8);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nq), dim3(NT), MCAP * 8, stream, args);
    return hipGetLastError();
}

hipError_t launch_merge_topk(const MergeArgs& args, int nq, hipStream_t stream) {
    // The exact scan's grid of one resident wave of blocks (1,024 on 256 CUs) x k = 10 is 10,240 entries: just past the
    // 8,192-entry one-pass form, whose streaming fallback took 25 us of a 1M-row query's 185.  Up to 16,384 entries the one-pass
    // form runs with 16 entries per thread (128 KB of LDS for the worst case of nothing pruned).
    const size_t total = (size_t)args.nlists * args.list_len;
    if (total > 8192 && total <= 16384) return launch_merge_t<16384>(args, nq, stream);
    return launch_merge_t<8192>(args, nq, stream);
}

hipError_t launch_score_rows(const ScanArgs& args, u64* out_packed, int q_index, int grid, hipStream_t stream) {
    if (args.dim % 8 != 0) {
        const int blocks = (int)((args.nrows + 255) / 256);
        hipLaunchKernelGGL(score_rows_generic_kernel, dim3(blocks), dim3(256), 0, stream, args, out_packed, q_index);
    } else {
        hipLaunchKernelGGL(score_rows_kernel, dim3(grid), dim3(256), (size_t)args.dim * 4, stream, args, out_packed,
                           q_index);
    }
    return hipGetLastError();
}

hipError_t launch_packed_to_sortkey(u64* data, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(packed_to_sortkey_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, data, n);
    return hipGetLastError();
}

hipError_t launch_sorted_keys_to_rows(const u64* keys, uint32_t k, uint32_t* out_rows, uint32_t* out_count,
                                      hipStream_t stream, u64* out_packed) {
    hipLaunchKernelGGL(sorted_keys_to_rows_kernel, dim3((k + 255) / 256), dim3(256), 0, stream, keys, k, out_rows,
                       out_count, out_packed);
    return hipGetLastError();
}

hipError_t launch_gather_dot(const ScanArgs& args, const uint32_t* rows, uint32_t n, float* out, hipStream_t stream) {
    const unsigned blocks = (unsigned)(((size_t)n * 4 + 255) / 256);
    hipLaunchKernelGGL(gather_dot_kernel, dim3(blocks ? blocks : 1), dim3(256), (size_t)args.dim * 4, stream, args,
                       rows, n, out);
    return hipGetLastError();
}

hipError_t launch_gather_dot_mq(const ScanArgs& args, const uint32_t* rows, const uint32_t* qidx, uint32_t n, float* out, hipStream_t stream) {
    if ((args.dim & 7) != 0) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)(((size_t)n * 4 + 255) / 256);
    hipLaunchKernelGGL(gather_dot_mq_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, args, rows, qidx, n, out);
    return hipGetLastError();
}

hipError_t launch_encode_f16(const float* src, size_t n, unsigned short* dst, hipStream_t stream) {
    hipLaunchKernelGGL(encode_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, n, dst);
    return hipGetLastError();
}

hipError_t launch_encode_rows_f16(const float* src, const uint32_t* perm, uint64_t n, uint32_t dim, unsigned short* dst,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(encode_rows_f16_kernel, dim3((unsigned)n), dim3(128), 0, stream, src, perm, n, dim, dst);
    return hipGetLastError();
}

// Batched-path fallbacks: the queries the matrix-core path could not certify are compacted, answered together by
// the exact kernels (8 per pass) and written back to their slots.
__global__ void gather_queries_kernel(const float* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t dim,
                                      uint32_t src_stride, float* __restrict__ dst) {
    const uint32_t j = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) dst[(size_t)j * dim + i] = src[(size_t)idx[j] * src_stride + i];
}

__global__ void scatter_hits_kernel(const uint32_t* __restrict__ idx, uint32_t k, const uint32_t* __restrict__ src_rows,
                                    const float* __restrict__ src_scores, const uint32_t* __restrict__ src_counts,
                                    uint32_t* __restrict__ dst_rows, float* __restrict__ dst_scores,
                                    uint32_t* __restrict__ dst_counts, u64* __restrict__ dst_packed) {
    const uint32_t j = blockIdx.x, q = idx[j];
    const uint32_t n = src_counts[j];
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        const uint32_t r = src_rows[(size_t)j * k + i];
        const float sc = src_scores[(size_t)j * k + i];
        if (dst_rows) dst_rows[(size_t)q * k + i] = i < n ? r : 0xffffffffu;
        if (dst_scores) dst_scores[(size_t)q * k + i] = sc;
        if (dst_packed) dst_packed[(size_t)q * k + i] = i < n ? pack(sc, r) : kEmpty;
    }
    if (threadIdx.x == 0 && dst_counts) dst_counts[q] = n;
}

hipError_t launch_gather_queries(const float* src, const uint32_t* idx, uint32_t n, uint32_t dim, uint32_t src_stride, float* dst,
                                 hipStream_t stream) {
    hipLaunchKernelGGL(gather_queries_kernel, dim3(n), dim3(128), 0, stream, src, idx, dim, src_stride ? src_stride : dim, dst);
    return hipGetLastError();
}

hipError_t launch_scatter_hits(const uint32_t* idx, uint32_t n, uint32_t k, const uint32_t* src_rows,
                               const float* src_scores, const uint32_t* src_counts, uint32_t* dst_rows,
                               float* dst_scores, uint32_t* dst_counts, u64* dst_packed, hipStream_t stream) {
    hipLaunchKernelGGL(scatter_hits_kernel, dim3(n), dim3(64), 0, stream, idx, k, src_rows, src_scores, src_counts, dst_rows,
                       dst_scores, dst_counts, dst_packed);
    return hipGetLastError();
}

hipError_t launch_widen_f16(const unsigned short* src, size_t n, float* dst, hipStream_t stream) {
    hipLaunchKernelGGL(widen_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, n, dst);
    return hipGetLastError();
}

}  // namespace fsgpu
