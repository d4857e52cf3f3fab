// sort_general.hip — descending radix sort of 64-bit sortkeys for the large-k / collect-all path (k beyond the fused top-k tiers, or
// k >= nrows: the reference's collect-all branch, crates/frankensearch-index/src/search.rs:449-473, and the candidate selection of a
// quantised pass 1 whose candidate count exceeds the fused tiers, search.rs:603-661).  The hot fused paths never come here.
//
// Least-significant-digit radix sort, 8 bits per pass, 8 passes, three launches per pass:
//   count   : every block histograms the digit of its tile of 8,192 keys (LDS atomics)            -> hist[digit][block]
//   scan    : one block per digit turns its row of block counts into exclusive prefixes, in place -> + totals[digit]
//   scatter : every block re-reads its tile; wave w owns a contiguous quarter of it, walks it 64 keys at a time IN ORDER, ranks each
//             key among the lanes of its chunk that hold the same digit (8 ballots) and writes it to
//             base[digit] + prefix[digit][block] + (keys of that digit in earlier waves / chunks) + rank — a stable pass.
// Descending order = ascending order of the inverted digit.  Wave-64 throughout: one ballot is one 64-key chunk.
// The scatter stages the tile in LDS in digit order first (64 KB), so a block's keys of one digit leave as one run instead of one
// 8-byte store per key wherever it lands.  Collect-all search at 10M x 384 (score every row + sort + re-score, p50): 3.02 ms with the
// direct stores, 2.36 ms staged and with the constant digit skipped; rounds 1-5 called rocprim::radix_sort_keys_desc here: 2.53 ms
// (1M rows: 0.46 against 0.41 ms — 21 launches).  profiles/r06/sort_ab.txt
// Buffers: passes ping-pong between a scratch copy and keys_out (the last pass lands in keys_out); keys_in is never written.
// `varying_bits`: the caller's promise about which key bits can differ at all — a digit without one is skipped.
#include <cstring>

#include "kernels.hpp"

namespace fsgpu {

namespace {

constexpr int kSortThreads = 256;                          // 4 waves
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortChunks = 32;                            // 64-key chunks per wave
constexpr uint32_t kSortTile = kSortThreads * kSortChunks; // 8,192 keys per block
constexpr uint32_t kSortWaveKeys = 64 * kSortChunks;

__device__ __forceinline__ uint32_t sort_digit(u64 key, int shift) { return 255u - (uint32_t)((key >> shift) & 255ull); }

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)v, o, 64);
        if (lane >= o) v += y;
    }
    return v;
}

__global__ __launch_bounds__(kSortThreads) void radix_count_kernel(const u64* __restrict__ keys, size_t n, int shift,
                                                                    uint32_t* __restrict__ hist, uint32_t nblocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kSortTile;
#pragma unroll 4
    for (int c = 0; c < kSortChunks; ++c) {
        const size_t i = base + (size_t)c * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[sort_digit(keys[i], shift)], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(kSortThreads) void radix_scan_kernel(uint32_t* __restrict__ hist, uint32_t nblocks, uint32_t* __restrict__ totals) {
    __shared__ uint32_t wsum[kSortWaves];
    uint32_t* h = hist + (size_t)blockIdx.x * nblocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += kSortThreads) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? h[i] : 0u;
        const uint32_t x = wave_inclusive_sum(v, lane);
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            const uint32_t s = wsum[w];
            if (w < wave) before += s;
            total += s;
        }
        if (i < nblocks) h[i] = carry + before + x - v;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(const u64* __restrict__ in, u64* __restrict__ out, size_t n, int shift,
                                                                      const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals,
                                                                      uint32_t nblocks) {
    __shared__ u64 sorted[kSortTile];            // the tile in digit order (64 KB): global writes then go out in runs, not key by key
    __shared__ uint32_t next[kSortWaves][256];   // first the per-wave digit counts, then the running positions inside `sorted`
    __shared__ uint32_t local_start[256];        // where a digit's keys start inside the tile ...
    __shared__ uint32_t global_start[256];       // ... and in the output
    __shared__ uint32_t wsum[2][kSortWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // where each digit's keys start in the output: exclusive prefix of the 256 totals (thread t = digit t)
    const uint32_t tot = totals[tid];
    const uint32_t inc = wave_inclusive_sum(tot, lane);
    if (lane == 63) wsum[0][wave] = inc;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) next[w][tid] = 0;
    __syncthreads();
    uint32_t digit_base = inc - tot;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w)
        if (w < wave) digit_base += wsum[0][w];

    const size_t tile0 = (size_t)blockIdx.x * kSortTile;
    const size_t sub = tile0 + (size_t)wave * kSortWaveKeys;
    u64 key[kSortChunks];
#pragma unroll
    for (int c = 0; c < kSortChunks; ++c) {
        const size_t i = sub + (size_t)c * 64 + lane;
        key[c] = i < n ? in[i] : 0ull;
    }
#pragma unroll
    for (int c = 0; c < kSortChunks; ++c)
        if (sub + (size_t)c * 64 + lane < n) atomicAdd(&next[wave][sort_digit(key[c], shift)], 1u);
    __syncthreads();
    {   // thread t = digit t: its keys' first position inside the tile (block-wide exclusive prefix), per wave, and in the output
        uint32_t cnt[kSortWaves], mine_total = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            cnt[w] = next[w][tid];
            mine_total += cnt[w];
        }
        const uint32_t linc = wave_inclusive_sum(mine_total, lane);
        if (lane == 63) wsum[1][wave] = linc;
        __syncthreads();
        uint32_t run = linc - mine_total;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w)
            if (w < wave) run += wsum[1][w];
        local_start[tid] = run;
        global_start[tid] = digit_base + hist[(size_t)tid * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            next[w][tid] = run;
            run += cnt[w];
        }
    }
    __syncthreads();
    volatile uint32_t* mine = next[wave];
    const u64 below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll   // (fully: key[] stays in registers)
    for (int c = 0; c < kSortChunks; ++c) {
        const bool valid = sub + (size_t)c * 64 + lane < n;
        const uint32_t d = sort_digit(key[c], shift);
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const u64 m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & below);
        const uint32_t at = mine[d];              // every lane reads before the chunk's leaders advance the positions
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            sorted[at + rank] = key[c];
            if (rank == 0) mine[d] = at + (uint32_t)__popcll(peers);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const uint32_t have = (uint32_t)(n - tile0 < kSortTile ? n - tile0 : kSortTile);
#pragma unroll 4
    for (uint32_t i = tid; i < have; i += kSortThreads) {
        const u64 kv = sorted[i];
        const uint32_t d = sort_digit(kv, shift);
        out[global_start[d] + (i - local_start[d])] = kv;
    }
}

inline uint32_t sort_blocks(size_t n) { return (uint32_t)((n + kSortTile - 1) / kSortTile); }
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

hipError_t sort_keys_desc_temp_bytes(size_t n, size_t* temp_bytes) {
    const size_t nblocks = sort_blocks(n ? n : 1);
    *temp_bytes = align256(n * 8) + align256(nblocks * 256 * 4) + 256 * 4;
    return hipSuccess;
}

hipError_t sort_keys_desc(void* temp, size_t temp_bytes, const u64* keys_in, u64* keys_out, size_t n, hipStream_t stream, u64 varying_bits) {
    if (n == 0) return hipSuccess;
    if (n > 0xffffffffull) return hipErrorInvalidValue;
    size_t need = 0;
    (void)sort_keys_desc_temp_bytes(n, &need);
    if (!temp || temp_bytes < need) return hipErrorInvalidValue;
    const uint32_t nblocks = sort_blocks(n);
    u64* scratch = static_cast<u64*>(temp);
    uint32_t* hist = reinterpret_cast<uint32_t*>(static_cast<char*>(temp) + align256(n * 8));
    uint32_t* totals = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(hist) + align256((size_t)nblocks * 256 * 4));
    // a digit none of whose bits varies over the keys (the high byte of the row half on slabs below 16.7M rows) needs no pass
    int shifts[8], passes = 0;
    for (int d = 0; d < 8; ++d)
        if ((varying_bits >> (8 * d)) & 255ull) shifts[passes++] = 8 * d;
    if (passes == 0) return hipMemcpyAsync(keys_out, keys_in, n * 8, hipMemcpyDeviceToDevice, stream);
    const u64* src = keys_in;
    for (int pass = 0; pass < passes; ++pass) {
        u64* dst = ((passes - 1 - pass) & 1) ? scratch : keys_out;   // the last pass lands in keys_out
        hipLaunchKernelGGL(radix_count_kernel, dim3(nblocks), dim3(kSortThreads), 0, stream, src, n, shifts[pass], hist, nblocks);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(256), dim3(kSortThreads), 0, stream, hist, nblocks, totals);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblocks), dim3(kSortThreads), 0, stream, src, dst, n, shifts[pass], hist, totals, nblocks);
        src = dst;
    }
    return hipGetLastError();
}

}  // namespace fsgpu
