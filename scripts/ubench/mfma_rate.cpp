// What the matrix pipe sustains on gfx950 with nothing else in the loop: f16 16x16x32, f16 32x32x16 and i8 16x16x64, with
// zero operands and with random operands (the chip clocks to its power budget: MI355X_MICROARCH.md, DVFS give-back).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_rate.cpp -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void k(const half8* in, float* out, int iters) {
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x + 512 * i) & 4095]; b[i] = in[(threadIdx.x * 7 + 512 * i + 3) & 4095]; }
    if (KIND == 0) {
        f32x4 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 24; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 3], b[(u >> 2) & 3], acc[u & 3], 0, 0, 0);
        out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else if (KIND == 1) {
        f32x16 acc[2] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 12; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u >> 2) & 3], acc[u & 1], 0, 0, 0);
        out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][5];
    } else {
        i32x4 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 24; ++u)
                acc[u & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a[u & 3]), __builtin_bit_cast(i32x4, b[(u >> 2) & 3]), acc[u & 3], 0, 0, 0);
        out[blockIdx.x * 512 + threadIdx.x] = (float)(acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]);
    }
}

int main() {
    const int iters = 20000, grid = 256;
    half8* in; float* out;
    hipMalloc(&in, 4096 * sizeof(half8)); hipMalloc(&out, grid * 512 * 4);
    for (int data = 0; data < 2; ++data) {
        std::vector<_Float16> h(4096 * 8);
        for (auto& v : h) v = data ? (_Float16)((rand() % 2001 - 1000) / 16000.0f) : (_Float16)0.f;
        hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int kind = 0; kind < 3; ++kind) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&] {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, in, out, iters);
                else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, in, out, iters);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per_wave = (kind == 1 ? 12.0 : 24.0) * iters;          // MFMAs per wave
            const double ops = per_wave * (kind == 2 ? 32768.0 : (kind == 1 ? 32768.0 : 16384.0)) * grid * 8;
            const double ns_per = ms * 1e6 / (per_wave * 2);                      // 2 waves per SIMD share the pipe
            printf("%s %-14s %.3f ms  %.0f T(FL)OP/s  %.2f ns per MFMA per SIMD\n", data ? "random" : "zeros ",
                   kind == 0 ? "f16 16x16x32" : kind == 1 ? "f16 32x32x16" : "i8 16x16x64", ms, ops / (ms * 1e-3) / 1e12, ns_per);
        }
    }
    return 0;
}
