// fp6_skeleton.hip — r05 verdict item 4: "decide the 6-bit filter with a skeleton before building it".
//
// A TIMING SKELETON of the batched search's main pass (mfma_wide.hip, scan_wide_body) in two element formats, identical in structure:
//   FMT 0  int8 rows of 384 bytes, v_mfma_i32_16x16x64_i8            (6 k-steps per row, 16 bytes per lane and k-step)   = what ships
//   FMT 1  fp6 (E2M3) rows of 288 bytes, v_mfma_scale_f32_16x16x128_f8f6f4 (3 k-steps, 24 bytes per lane and k-step)     = the candidate
// Per block (512 threads, one per CU, persistent over its share of the row tiles): queries resident in registers (QT query tiles of
// 16 per wave, 8 waves: 128 x QT queries per pass), row tiles of 128 rows through a three-slot LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4, hand-counted s_waitcnt vmcnt, one s_barrier per tile), every wave reading every row fragment of the tile
// from LDS (ds_read_b128 / 3 x ds_read_b64: conflict-free linear images — the numbers are meaningless, the bytes and the LDS cycles are
// the real kernel's), the accumulators started at -tau and reduced to one sign test per sub-tile pair (the neg-tau test of the real
// loop), a store behind a wave vote that never fires.  No bound math, no appends, random data.  What it answers: the ratio of the two
// formats' times for one 512-query pass over 10M rows, and how close the int8 skeleton comes to the shipped launch (2 passes in
// ~2.67 ms = 1.33 ms per pass) — i.e. how much of the real kernel the skeleton models.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/fp6_skeleton.hip -o scripts/ubench/_build/fp6_skeleton
//   scripts/ubench/_build/fp6_skeleton [rows=10000000]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int FMT, int QT>
__global__ __launch_bounds__(512, 1) void skel(const unsigned char* __restrict__ slab, uint32_t ntiles, const i32x8* __restrict__ qfrag, float tau,
                                                float* __restrict__ out) {
    using qfrag_t = std::conditional_t<FMT == 0, i32x4, i32x8>;
    constexpr int ROWB = FMT ? 288 : 384, KS = FMT ? 3 : 6, TR = 128, TILE_B = TR * ROWB, NI = TILE_B / 1024, NSLOT = 3;
    constexpr int FB = FMT ? 24 : 16;   // fragment bytes per lane and k-step
    static_assert(TILE_B % 1024 == 0, "whole DMA instructions per tile");
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ring_addr = (uint32_t)(uintptr_t)ring;   // LDS byte address
    // resident query fragments: QT tiles x KS k-steps
    qfrag_t bq[QT][KS];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const i32x8 v = qfrag[((wave * QT + t) * KS + k) * 64 + lane];
            if constexpr (FMT == 0) bq[t][k] = i32x4{v[0], v[1], v[2], v[3]};
            else bq[t][k] = i32x8{v[0], v[1], v[2], v[3], v[4], v[5], 0, 0};
        }
    // this block's tiles: b, b + grid, ...
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    const uint32_t my_tiles = b < ntiles ? (ntiles - b + nb - 1) / nb : 0;
    // DMA share of a tile: pieces wave, wave + 8, ... (NI = 36: waves 0-3 issue 5, waves 4-7 issue 4; NI = 48: 6 each)
    constexpr int PW_MAX = (NI + 7) / 8;
    const int my_pieces = (NI - wave + 7) / 8;
    auto issue_tile = [&](uint32_t i, uint32_t slot) {
        const unsigned char* g = slab + (size_t)(b + i * nb) * TILE_B + lane * 16;
#pragma unroll
        for (int x = 0; x < PW_MAX; ++x) {
            const int p = wave + 8 * x;
            if (p < NI) glds16(g + (size_t)p * 1024, __builtin_amdgcn_readfirstlane(ring_addr + slot * TILE_B + p * 1024));
        }
    };
    float keep = 0.f;
    if (my_tiles) {
        issue_tile(0, 0);
        if (my_tiles > 1) issue_tile(1, 1);
        for (uint32_t i = 0; i < my_tiles; ++i) {
            // tile i has landed for this wave once only tile i + 1's pieces are outstanding
            if (i + 1 < my_tiles) {
                if (my_pieces == PW_MAX) wait_vmcnt<PW_MAX>();
                else wait_vmcnt<PW_MAX - 1>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();   // ... and for everyone; tile i - 1's slot is free
            asm volatile("" ::: "memory");
            if (i + 2 < my_tiles) issue_tile(i + 2, (i + 2) % NSLOT);
            const unsigned char* base = ring + (size_t)(i % NSLOT) * TILE_B;
#pragma unroll
            for (int sp = 0; sp < TR / 32; ++sp) {   // sub-tile pairs
                if constexpr (FMT == 0) {
                    i32x4 acc[2][QT];
                    const int nt = -(int)tau;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < QT; ++t) acc[h][t] = i32x4{nt, nt, nt, nt};
                    i32x4 f[2][KS];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int k = 0; k < KS; ++k)
                            f[h][k] = *reinterpret_cast<const i32x4*>(base + ((sp * 2 + h) * KS + k) * 1024 + lane * 16);
#pragma unroll
                    for (int k = 0; k < KS; ++k)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int t = 0; t < QT; ++t) {
                                if constexpr (FMT == 0) acc[h][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f[h][k], bq[t][k], acc[h][t], 0, 0, 0);
                            }
                    int m = acc[0][0][0];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < QT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) m = acc[h][t][r] > m ? acc[h][t][r] : m;
                    if (__ballot(m >= 0)) keep += (float)m;   // (never: tau is far above every score)
                } else {
                    f32x4 acc[2][QT];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < QT; ++t) acc[h][t] = f32x4{-tau, -tau, -tau, -tau};
                    i32x8 f[2][KS];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int k = 0; k < KS; ++k) {
                            // 24 bytes per lane: three 8-byte reads of a linear image (16 rows x 96 bytes of a k-step = 1,536 bytes)
                            const unsigned char* p = base + (size_t)(sp * 2 + h) * (16 * ROWB) + k * (64 * FB) + lane * FB;
                            const i32x2 a0 = *reinterpret_cast<const i32x2*>(p), a1 = *reinterpret_cast<const i32x2*>(p + 8),
                                        a2 = *reinterpret_cast<const i32x2*>(p + 16);
                            f[h][k] = i32x8{a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], 0, 0};
                        }
#pragma unroll
                    for (int k = 0; k < KS; ++k)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int t = 0; t < QT; ++t)
                                if constexpr (FMT != 0)
                                    acc[h][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(f[h][k], bq[t][k], acc[h][t], 2, 2, 0, 127, 0, 127);
                    float m = acc[0][0][0];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < QT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[h][t][r]);
                    if (__ballot(m >= 0.f)) keep += m;
                }
            }
        }
    }
    wait_vmcnt<0>();
    if (keep != 0.f) out[blockIdx.x * 512 + tid] = keep;
}

template <int FMT, int QT>
static double run(const unsigned char* slab, uint64_t rows, const i32x8* qfrag, float* out, int reps) {
    constexpr int ROWB = FMT ? 288 : 384;
    const uint32_t ntiles = (uint32_t)(rows / 128);
    const size_t lds = (size_t)3 * 128 * ROWB;
    hipFuncSetAttribute(reinterpret_cast<const void*>(skel<FMT, QT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((skel<FMT, QT>), dim3(256), dim3(512), lds, 0, slab, ntiles, qfrag, 1.0e9f, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((skel<FMT, QT>), dim3(256), dim3(512), lds, 0, slab, ntiles, qfrag, 1.0e9f, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
    return ms / reps;
}

int main(int argc, char** argv) {
    const uint64_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10'000'000ull;
    unsigned char* slab;
    i32x8* qfrag;
    float* out;
    const size_t bytes = (size_t)rows * 384;
    hipMalloc(&slab, bytes);
    hipMalloc(&qfrag, (size_t)8 * 8 * 6 * 64 * sizeof(i32x8));
    hipMalloc(&out, 256 * 512 * 4);
    {
        // random bytes with the exponent bits of every fp6 / int8 code kept small (finite, mid-range values: the chip clocks to its power budget)
        std::vector<uint32_t> h(1 << 24);
        for (auto& v : h) v = (((uint32_t)rand() << 16) ^ (uint32_t)rand()) & 0x37373737u;
        for (size_t off = 0; off < bytes; off += h.size() * 4) hipMemcpy(slab + off, h.data(), std::min(bytes - off, h.size() * 4), hipMemcpyHostToDevice);
        hipMemcpy(qfrag, h.data(), (size_t)8 * 8 * 6 * 64 * sizeof(i32x8), hipMemcpyHostToDevice);
    }
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device %s, %d CUs; %llu rows; one pass = 128 x QT queries against every row, 256 blocks x 512 threads\n", prop.name, prop.multiProcessorCount,
           (unsigned long long)rows);
    for (int round = 0; round < 2; ++round) {
        const double i4 = run<0, 4>(slab, rows, qfrag, out, 10), f4 = run<1, 4>(slab, rows, qfrag, out, 10);
        const double f5 = run<1, 5>(slab, rows, qfrag, out, 10), f6 = run<1, 6>(slab, rows, qfrag, out, 10);
        auto rate = [&](double ms, int qt, int dim) { return 2.0 * rows * dim * 128.0 * qt / (ms * 1e-3) / 1e15; };
        printf("int8 384 B rows, QT 4 (512 queries per pass): %.3f ms  %.2f POP/s   | HBM %.2f TB/s\n", i4, rate(i4, 4, 384), rows * 384.0 / (i4 * 1e-3) / 1e12);
        printf("fp6  288 B rows, QT 4 (512 queries per pass): %.3f ms  %.2f PFLOP/s | HBM %.2f TB/s   int8 / fp6 = %.3f\n", f4, rate(f4, 4, 384),
               rows * 288.0 / (f4 * 1e-3) / 1e12, i4 / f4);
        printf("fp6  288 B rows, QT 5 (640 queries per pass): %.3f ms  %.2f PFLOP/s   per 512 queries %.3f ms   int8 / fp6 = %.3f\n", f5, rate(f5, 5, 384),
               f5 * 512 / 640, i4 / (f5 * 512 / 640));
        printf("fp6  288 B rows, QT 6 (768 queries per pass): %.3f ms  %.2f PFLOP/s   per 512 queries %.3f ms   int8 / fp6 = %.3f\n", f6, rate(f6, 6, 384),
               f6 * 512 / 768, i4 / (f6 * 512 / 768));
    }
    return 0;
}
