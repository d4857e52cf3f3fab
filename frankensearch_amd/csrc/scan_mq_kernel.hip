// scan_mq_kernel.hip — multi-query (4 or 8 queries per pass) form of the fused f16 scan + top-k.
//
// Same arithmetic contract as scan_kernels.hip (reference order of dot_product_f16_bytes_f32,
// crates/frankensearch-index/src/simd.rs:398-446; ordering of search.rs:1655-1686) — scores are bit-identical.
//
// Why a second kernel: with more than two queries per pass the lane-replicated mapping of scan_kernels.hip runs
// out of bytes in flight (each wave tile shrinks to 16/NQ rows while the per-lane query slice pins 96 VGPRs), and
// becomes latency-bound (measured 0.51 of HBM peak at NQ=4).  Here a lane is (row r, accumulator a) as in the
// single-query kernel — a wave still streams a full 16-row tile (12 KB at dim 384) with double-buffered register
// tiles — and carries NQ x 8 accumulators.  The queries live in LDS as [NQ][DIM] f32; lane a fetches its 8-float
// slice of chunk 4g+a for each query with two ds_read_b128 (16 rows broadcast, 4 distinct addresses: conflict
// free).  Those reads are issued through inline asm one chunk at a time: left to hipcc, all G*NQ*2 reads are
// hoisted to the top of the tile (up to 768 VGPRs) and spill.
#pragma clang fp contract(off)

#include <utility>

#include "scan_common.hpp"

namespace fsgpu {

using namespace scan_detail;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Two ds_read_b128 at a compile-time byte offset from the lane's LDS base; the destination registers are only
// valid after the matching lds_wait.
template <int OFF>
__device__ __forceinline__ void lds_read_q(f32x4& q0, f32x4& q1, uint32_t lds_addr) {
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                 : "=&v"(q0), "=&v"(q1)
                 : "v"(lds_addr), "i"(OFF), "i"(OFF + 16));
}

// Wait until at most PENDING LDS reads are outstanding (the 8 of the next step, or none); names every
// destination of the CURRENT step so no consumer is scheduled above the wait (cdna_hip_programming.md §5.7 ii).
template <int PENDING>
__device__ __forceinline__ void lds_wait4(f32x4 (&q0)[4], f32x4 (&q1)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(q0[0]), "+v"(q1[0]), "+v"(q0[1]), "+v"(q1[1]), "+v"(q0[2]), "+v"(q1[2]), "+v"(q0[3]), "+v"(q1[3])
                 : "i"(PENDING));
}

// Empty volatile statement naming every accumulator: the following volatile LDS reads cannot be hoisted above
// the arithmetic that produced these values.  Without it hipcc issues the whole (independent) chain of reads
// first and needs every chunk's query slices live at once (512 registers, spills).
template <int NQ>
__device__ __forceinline__ void acc_barrier(f32x4 (&acc)[NQ][2]) {
    if constexpr (NQ == 4) {
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]),
                          "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
    } else {
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]),
                          "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[4][0]), "+v"(acc[4][1]),
                          "+v"(acc[5][0]), "+v"(acc[5][1]), "+v"(acc[6][0]), "+v"(acc[6][1]), "+v"(acc[7][0]),
                          "+v"(acc[7][1]));
    }
}

__device__ __forceinline__ void chunk_mac_v(f32x4 (&acc)[2], const u32x4& w, const f32x4& q0, const f32x4& q1) {
    const half8 h = __builtin_bit_cast(half8, w);
    float p;
    p = (float)h[0] * q0[0]; acc[0][0] = acc[0][0] + p;
    p = (float)h[1] * q0[1]; acc[0][1] = acc[0][1] + p;
    p = (float)h[2] * q0[2]; acc[0][2] = acc[0][2] + p;
    p = (float)h[3] * q0[3]; acc[0][3] = acc[0][3] + p;
    p = (float)h[4] * q1[0]; acc[1][0] = acc[1][0] + p;
    p = (float)h[5] * q1[1]; acc[1][1] = acc[1][1] + p;
    p = (float)h[6] * q1[2]; acc[1][2] = acc[1][2] + p;
    p = (float)h[7] * q1[3]; acc[1][3] = acc[1][3] + p;
}

// A "step" is (chunk g, query group h) = 4 queries x one 16-byte chunk: 8 ds_read_b128 into one staging set.
template <int DIM, int NQ, int STEP, int... B>
__device__ __forceinline__ void issue_step(f32x4 (&q0)[4], f32x4 (&q1)[4], uint32_t lds_addr,
                                           std::integer_sequence<int, B...>) {
    constexpr int HPC = NQ / 4;  // steps per chunk
    constexpr int G = STEP / HPC, H = STEP % HPC;
    (lds_read_q<((4 * H + B) * DIM + 32 * G) * 4>(q0[B], q1[B], lds_addr), ...);
}

// One tile: acc[b] += w[g] (x) q[b][chunk 4g+a].  Step s+1's reads are in flight while step s multiplies.
template <int DIM, int NQ, int... Ss>
__device__ __forceinline__ void mac_tile(f32x4 (&acc)[NQ][2], const u32x4 (&w)[DIM / 32], uint32_t lds_addr,
                                         std::integer_sequence<int, Ss...>) {
    constexpr int HPC = NQ / 4;
    constexpr int S = (DIM / 32) * HPC;
    f32x4 q0[2][4], q1[2][4];
    issue_step<DIM, NQ, 0>(q0[0], q1[0], lds_addr, std::make_integer_sequence<int, 4>{});
    (
        [&] {
            constexpr int cur = Ss & 1;
            if constexpr (Ss + 1 < S) {
                issue_step<DIM, NQ, (Ss + 1 < S ? Ss + 1 : 0)>(q0[cur ^ 1], q1[cur ^ 1], lds_addr,
                                                               std::make_integer_sequence<int, 4>{});
                lds_wait4<8>(q0[cur], q1[cur]);
            } else {
                lds_wait4<0>(q0[cur], q1[cur]);
            }
            constexpr int g = Ss / HPC, h = Ss % HPC;
#pragma unroll
            for (int b = 0; b < 4; ++b) chunk_mac_v(acc[4 * h + b], w[g], q0[cur][b], q1[cur][b]);
            acc_barrier<NQ>(acc);
        }(),
        ...);
}

}  // namespace

template <int DIM, int NQ, int KCAP>
__global__ __launch_bounds__(256) void scan_mq_topk_kernel(ScanArgs args) {
    static_assert(NQ == 4 || NQ == 8, "multi-query kernel serves 4 or 8 queries per pass");
    constexpr int CAP = 2 * KCAP;
    constexpr int G = DIM / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* qs = reinterpret_cast<float*>(smem);                                            // [NQ][DIM]
    u64* bufs = reinterpret_cast<u64*>(smem + (((size_t)NQ * DIM * 4 + 15) & ~(size_t)15));  // [wave][NQ][CAP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int a = lane & 3;
    const int r = lane >> 2;

    for (int i = tid; i < NQ * DIM; i += 256) qs[i] = args.queries[i];
    __syncthreads();
    const uint32_t lds_addr = (uint32_t)(uintptr_t)qs + (uint32_t)a * 32u;  // LDS byte offset of the lane's slice

    WaveTopK<CAP> tk[NQ];
#pragma unroll
    for (int x = 0; x < NQ; ++x) tk[x].init(bufs + ((size_t)wave * NQ + x) * CAP);
    u64 thr[NQ / 4];  // lane a owns queries a, a+4
#pragma unroll
    for (int h = 0; h < NQ / 4; ++h) thr[h] = 0;

    const uint32_t nrows = args.nrows;
    const uint32_t ntiles = (nrows + kRowsPerTile - 1) / kRowsPerTile;
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    const uint32_t wave_gid = blockIdx.x * kWavesPerBlock + wave;
    const unsigned char* slab = reinterpret_cast<const unsigned char*>(args.slab);
    constexpr size_t row_bytes = (size_t)DIM * 2;
    const int k = (int)args.k;
    const int hreduce = args.hreduce;

    auto load_tile = [&](uint32_t tile, u32x4 (&w)[G]) {
        uint32_t row = tile * kRowsPerTile + r;
        row = row < nrows ? row : nrows - 1;
        const u32x4* p = reinterpret_cast<const u32x4*>(slab + (size_t)row * row_bytes) + a;
#pragma unroll
        for (int g = 0; g < G; ++g) w[g] = p[4 * g];
    };
    auto tile_words = [&](uint32_t tile, u64& live_word, u64& allow_word) {
        const uint32_t w64 = (tile * kRowsPerTile) >> 6;
        live_word = args.live ? args.live[w64] : ~0ull;
        allow_word = args.allow ? args.allow[w64] : ~0ull;
    };
    auto compute_tile = [&](uint32_t tile, const u32x4 (&w)[G], u64 live_word, u64 allow_word) {
        f32x4 acc[NQ][2];
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
            acc[b][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[b][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        mac_tile<DIM, NQ>(acc, w, lds_addr, std::make_integer_sequence<int, G * (NQ / 4)>{});
        const uint32_t row = tile * kRowsPerTile + r;
        bool valid = row < nrows;
        valid = valid && ((live_word >> (row & 63)) & 1ull) && ((allow_word >> (row & 63)) & 1ull);
        // every lane of the quad gets all NQ scores; lane a keeps those of queries a (and a+4)
        float mine[NQ / 4];
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
            const float lanes[8] = {acc[b][0][0], acc[b][0][1], acc[b][0][2], acc[b][0][3],
                                    acc[b][1][0], acc[b][1][1], acc[b][1][2], acc[b][1][3]};
            const float s = quad_finish(lanes, hreduce);
            if ((b & 3) == a) mine[b >> 2] = s;
        }
#pragma unroll
        for (int h = 0; h < NQ / 4; ++h) {
            const u64 packed = pack(mine[h], args.row_base + row);
            bool cand = valid && sortkey(packed) > thr[h];
            const u64 m = __ballot(cand);
            if (m == 0) continue;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int qx = h * 4 + x;
                const u64 xmask = 0x1111111111111111ull << x;
                u64 mx = m & xmask;
                if (mx == 0) continue;
                if (tk[qx].count + (int)__popcll(mx) > CAP) {
                    const u64 t = tk[qx].compact(k, lane);
                    if (a == x) thr[h] = t;
                    cand = cand && sortkey(packed) > thr[h];
                    mx = __ballot(cand) & xmask;
                }
                if (cand && a == x) {
                    const int pos = tk[qx].count + (int)__popcll(mx & ((1ull << lane) - 1ull));
                    tk[qx].buf[pos] = packed;
                }
                tk[qx].count += (int)__popcll(mx);
            }
        }
    };

    {
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
    }

    // ---- block merge (same as scan_topk_kernel) ----
#pragma unroll
    for (int x = 0; x < NQ; ++x) (void)tk[x].compact(k, lane);
    __syncthreads();
    for (int x = wave; x < NQ; x += kWavesPerBlock) {
        u64* dst = bufs + ((size_t)0 * NQ + x) * CAP;
        for (int w = 1; w < kWavesPerBlock; ++w) {
            const u64* src = bufs + ((size_t)w * NQ + x) * CAP;
            for (int i = lane; i < KCAP; i += 64) {
                const u64 xx = dst[i], yy = src[KCAP - 1 - i];
                dst[i] = sortkey(xx) >= sortkey(yy) ? xx : yy;
            }
            for (int i = KCAP + lane; i < CAP; i += 64) dst[i] = kEmpty;
            wave_sort_desc<CAP>(dst, lane);
        }
        u64* out = args.partial + ((size_t)x * gridDim.x + blockIdx.x) * k;
        for (int i = lane; i < k; i += 64) out[i] = dst[i];
    }
}

// ---- host side ----------------------------------------------------------------------------------------------

bool scan_mq_supported(int dim, int nq, int kcap) {
    if (!(nq == 4 || nq == 8)) return false;
    if (!(dim == 128 || dim == 256 || dim == 384)) return false;
    return scan_lds_bytes(dim, nq, kcap) <= 160 * 1024;
}

template <int DIM, int NQ, int KCAP>
static hipError_t launch_mq_t(const ScanArgs& args, int grid, hipStream_t stream, int* occupancy) {
    const size_t lds = scan_lds_bytes(DIM, NQ, KCAP);
    auto kern = scan_mq_topk_kernel<DIM, NQ, KCAP>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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

template <int NQ, int KCAP>
static hipError_t launch_mq_dim(const ScanArgs& args, int grid, hipStream_t stream, int* occupancy) {
    switch (args.dim) {
        case 128: return launch_mq_t<128, NQ, KCAP>(args, grid, stream, occupancy);
        case 256: return launch_mq_t<256, NQ, KCAP>(args, grid, stream, occupancy);
        case 384: return launch_mq_t<384, NQ, KCAP>(args, grid, stream, occupancy);
        default: return hipErrorInvalidValue;
    }
}

// occupancy != nullptr: only query the resident blocks per CU of the instantiation.
hipError_t launch_scan_mq(const ScanArgs& args, int nq, int kcap, int grid, hipStream_t stream, int* occupancy) {
    if (nq == 4 && kcap == 64) return launch_mq_dim<4, 64>(args, grid, stream, occupancy);
    if (nq == 8 && kcap == 64) return launch_mq_dim<8, 64>(args, grid, stream, occupancy);
    if (nq == 4 && kcap == 256) return launch_mq_dim<4, 256>(args, grid, stream, occupancy);
    if (nq == 8 && kcap == 256) return launch_mq_dim<8, 256>(args, grid, stream, occupancy);
    return hipErrorInvalidValue;
}

}  // namespace fsgpu
