// scan_common.hpp — device helpers shared by the scan kernels (scan_kernels.hip, scan_mq_kernel.hip).
#pragma once

#include "device_util.hpp"
#include "kernels.hpp"

namespace fsgpu {
namespace scan_detail {

constexpr int kWavesPerBlock = 4;
constexpr int kRowsPerTile = 16;

__device__ __forceinline__ u32x4 load_nt16(const u32x4* p) { return __builtin_nontemporal_load(p); }

// ---- per-wave top-k state over an LDS buffer of CAP packed entries ---------------------------------
template <int CAP>
struct WaveTopK {
    u64* buf;   // CAP entries (LDS)
    int count;  // wave-uniform
    __device__ __forceinline__ void init(u64* b) {
        buf = b;
        count = 0;
    }
    // Sort, trim to k, return the new threshold sortkey (0 while fewer than k entries are held).
    __device__ __forceinline__ u64 compact(int k, int lane) {
        for (int i = count + lane; i < CAP; i += 64) buf[i] = kEmpty;
        wave_sort_desc<CAP>(buf, lane);
        if (count > k) count = k;
        u64 thr = 0;
        if (count == k) thr = sortkey(buf[k - 1]);
        return thr;
    }
};

// One chunk (8 f16 x 8 f32) into the lane's 8 accumulators: separate multiply and add.
__device__ __forceinline__ void chunk_mac(float (&acc)[8], const u32x4& w, const float4& q0, const float4& q1) {
    const half8 h = __builtin_bit_cast(half8, w);
    float p;
    p = (float)h[0] * q0.x; acc[0] = acc[0] + p;
    p = (float)h[1] * q0.y; acc[1] = acc[1] + p;
    p = (float)h[2] * q0.z; acc[2] = acc[2] + p;
    p = (float)h[3] * q0.w; acc[3] = acc[3] + p;
    p = (float)h[4] * q1.x; acc[4] = acc[4] + p;
    p = (float)h[5] * q1.y; acc[5] = acc[5] + p;
    p = (float)h[6] * q1.z; acc[6] = acc[6] + p;
    p = (float)h[7] * q1.w; acc[7] = acc[7] + p;
}

// (s0+s1)+(s2+s3) across the quad, then the horizontal add; every lane of the quad gets the result.
__device__ __forceinline__ float quad_finish(const float (&acc)[8], int hreduce) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float u = acc[j] + quad_xor1(acc[j]);  // lanes 0,1: s0+s1   lanes 2,3: s2+s3
        v[j] = u + quad_xor2(u);                     // (s0+s1)+(s2+s3)
    }
    return hreduce8(v, hreduce);
}


}  // namespace scan_detail
}  // namespace fsgpu
