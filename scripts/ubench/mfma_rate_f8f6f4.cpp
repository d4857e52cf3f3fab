// What the gfx950 matrix pipe sustains on the block-scaled 16x16x128 instruction with fp8 / fp6 / fp4 operands (unit scales), next to
// the int8 16x16x64 instruction the batched search's filter runs on — zero and random operands (the chip clocks to its power budget).
// A filter on 6-bit rows would stream 3/4 of the int8 copy's bytes and, if the pipe really runs it at twice the int8 rate, halve the
// main pass's matrix time (what comes next).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_rate_f8f6f4.cpp -o /tmp/mfma_rate_f8f6f4 && /tmp/mfma_rate_f8f6f4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FMT>   // 0: fp8 e4m3, 2: fp6 e2m3, 4: fp4 e2m1, -1: int8 16x16x64
__global__ __launch_bounds__(512) void k(const i32x8* in, float* out, int iters) {
    i32x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x + 512 * i) & 2047]; b[i] = in[(threadIdx.x * 7 + 512 * i + 3) & 2047]; }
    if constexpr (FMT < 0) {
        i32x4 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                const i32x4 x = {a[u & 3][0], a[u & 3][1], a[u & 3][2], a[u & 3][3]}, y = {b[(u >> 2) & 3][0], b[(u >> 2) & 3][1], b[(u >> 2) & 3][2], b[(u >> 2) & 3][3]};
                acc[u & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, acc[u & 3], 0, 0, 0);
            }
        out[blockIdx.x * 512 + threadIdx.x] = (float)(acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]);
    } else {
        f32x4 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 24; ++u)
                acc[u & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[u & 3], b[(u >> 2) & 3], acc[u & 3], FMT, FMT, 0, 127, 0, 127);
        out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    }
}

int main() {
    const int iters = 20000, grid = 256;
    i32x8* in; float* out;
    hipMalloc(&in, 2048 * sizeof(i32x8)); hipMalloc(&out, grid * 512 * 4);
    for (int data = 0; data < 2; ++data) {
        std::vector<int> h(2048 * 8);
        for (auto& v : h) v = data ? (int)(((unsigned)rand() << 16) ^ (unsigned)rand()) & 0x37373737 : 0;   // (exponent bits kept small: finite values in every format)
        hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int kind = 0; kind < 4; ++kind) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&] {
                if (kind == 0) hipLaunchKernelGGL(k<-1>, dim3(grid), dim3(512), 0, 0, in, out, iters);
                else if (kind == 1) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, in, out, iters);
                else if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 0, 0, in, out, iters);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per_wave = 24.0 * iters;
            const double ops = per_wave * (kind == 0 ? 32768.0 : 65536.0) * grid * 8;   // 2 * 16 * 16 * K
            printf("%s %-22s %.3f ms  %.0f T(FL)OP/s  %.2f ns per MFMA per SIMD\n", data ? "random" : "zeros ",
                   kind == 0 ? "i8 16x16x64" : kind == 1 ? "fp8 e4m3 16x16x128" : kind == 2 ? "fp6 e2m3 16x16x128" : "fp4 e2m1 16x16x128", ms,
                   ops / (ms * 1e-3) / 1e12, ms * 1e6 / (per_wave * 2));
        }
    }
    return 0;
}
