// device_util.hpp — gfx950 device helpers shared by the fsgpu kernels.
//
// Ordering contract (crates/frankensearch-index/src/search.rs:91-126,1655-1686 in the reference):
// best-first = higher score_key first where score_key maps NaN to -inf and compares with f32
// total_cmp (-0.0 < +0.0); ties go to the LOWER row.  A candidate is carried as one 64-bit word
//   packed  = raw f32 score bits << 32 | global row id
// and compared through
//   sortkey = ord(score bits) << 32 | ~row
// where ord() is the monotone u32 image of that total order.  Larger sortkey == better, all
// sortkeys of real rows are distinct and > 0, so top-k selection is an order-independent integer
// problem: any parallel schedule yields the reference's exact ranking.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fsgpu {

typedef unsigned long long u64;

// Padding value: sorts below every real entry (row 0xFFFFFFFF never exists: row < 2^32-1).
static constexpr u64 kEmpty = ~0ull;

__device__ __forceinline__ uint32_t score_ord(uint32_t bits) {
    if ((bits & 0x7fffffffu) > 0x7f800000u) bits = 0xff800000u;  // NaN ranks as -inf (score_key)
    return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);  // f32::total_cmp order
}

__device__ __forceinline__ u64 sortkey(u64 packed) {
    return ((u64)score_ord((uint32_t)(packed >> 32)) << 32) | (uint32_t)(~(uint32_t)packed);
}

// the f32 score bits a sortkey was made from (a NaN score comes back as -inf: that is how it ranks)
__device__ __forceinline__ uint32_t score_from_sortkey(u64 key) {
    const uint32_t ord = (uint32_t)(key >> 32);
    return (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
}

__device__ __forceinline__ u64 pack(float score, uint32_t row) {
    return ((u64)__float_as_uint(score) << 32) | row;
}

// Compiler-level ordering of LDS traffic inside ONE wave (DS ops of a wave execute in order in
// hardware; this only stops hipcc from moving/caching accesses across the point).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Quad (4-lane) butterfly through DPP quad_perm — no LDS, no bpermute.
__device__ __forceinline__ float quad_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));  // [1,0,3,2]
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));  // [2,3,0,1]
}

// Wave-wide unsigned max through DPP (row_shr 1/2/4/8 fold each row of 16 into its last lane, row_bcast15/31 carry
// rows 0->1, 2->3 and 1->2,3), read back from lane 63: no LDS traffic, ~10 VALU ops.  Returned value is wave-uniform.
__device__ __forceinline__ uint32_t wave_umax32(uint32_t v) {
    int x = (int)v;
#define FSGPU_DPP_MAX(ctrl, row_mask)                                                                \
    {                                                                                                \
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, x, ctrl, row_mask, 0xF, false);  \
        x = (int)(t > (uint32_t)x ? t : (uint32_t)x);                                                \
    }
    FSGPU_DPP_MAX(0x111, 0xF)  // row_shr:1
    FSGPU_DPP_MAX(0x112, 0xF)  // row_shr:2
    FSGPU_DPP_MAX(0x114, 0xF)  // row_shr:4
    FSGPU_DPP_MAX(0x118, 0xF)  // row_shr:8
    FSGPU_DPP_MAX(0x142, 0xA)  // row_bcast:15 into rows 1 and 3
    FSGPU_DPP_MAX(0x143, 0xC)  // row_bcast:31 into rows 2 and 3
#undef FSGPU_DPP_MAX
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

// Wave-wide max of 64-bit sort keys (0 = none): high words first, then the low words of the lanes that tie on it.
__device__ __forceinline__ u64 wave_max_key(u64 m) {
    const uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
    const uint32_t mh = wave_umax32(hi);
    const uint32_t ml = wave_umax32(hi == mh ? lo : 0u);
    return ((u64)mh << 32) | ml;
}

// The wave's k best of PER entries per lane, best first, into out[0..k) (pre-filled with kEmpty by the caller).
// key[] = sortkey or 0 for holes; keys are unique, so each round retires exactly one entry.
template <int PER>
__device__ __forceinline__ void wave_extract_topk(u64 (&key)[PER], const u64 (&e)[PER], int k, u64* out) {
    for (int r = 0; r < k; ++r) {
        u64 m = key[0];
#pragma unroll
        for (int x = 1; x < PER; ++x) m = key[x] > m ? key[x] : m;
        const u64 wm = wave_max_key(m);
        if (wm == 0ull) break;  // wave-uniform: nothing left
        if (m == wm) {
#pragma unroll
            for (int x = 0; x < PER; ++x)
                if (key[x] == wm) {
                    out[r] = e[x];
                    key[x] = 0ull;
                }
        }
    }
}

// wide::f32x8::reduce_add (third-party; simd.rs:439,563).  mode 0 = SSE2 build order, 1 = AVX order, 2 = two sequential
// 4-lane sums (f32x8 = a.reduce_add() + b.reduce_add() with a left-to-right f32x4 sum).
__device__ __forceinline__ float hreduce8(const float (&v)[8], int mode) {
    if (mode == 2) {
        float a = ((v[0] + v[1]) + v[2]) + v[3];
        float b = ((v[4] + v[5]) + v[6]) + v[7];
        return a + b;
    }
    if (mode == 1) {
        float a = v[0] + v[4], b = v[1] + v[5], c = v[2] + v[6], d = v[3] + v[7];
        float lo = a + c, hi = b + d;
        return lo + hi;
    }
    float a = (v[0] + v[2]) + (v[1] + v[3]);
    float b = (v[4] + v[6]) + (v[5] + v[7]);
    return a + b;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// One wave sorts CAP (power of two, >= 128) packed entries in LDS, best first.
template <int CAP>
__device__ __forceinline__ void wave_sort_desc(u64* buf, int lane) {
#pragma unroll 1
    for (int size = 2; size <= CAP; size <<= 1) {
#pragma unroll 1
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            wave_lds_fence();
#pragma unroll
            for (int t = lane; t < CAP / 2; t += 64) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const u64 x = buf[lo], y = buf[hi];
                const bool lt = sortkey(x) < sortkey(y);
                if (lt == desc) {
                    buf[lo] = y;
                    buf[hi] = x;
                }
            }
        }
    }
    wave_lds_fence();
}

// One wave sorts n (power of two, >= 2, runtime) packed entries in LDS, best first.
__device__ __forceinline__ void wave_sort_desc_rt(u64* buf, int n, int lane) {
#pragma unroll 1
    for (int size = 2; size <= n; size <<= 1) {
#pragma unroll 1
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            wave_lds_fence();
#pragma unroll 1
            for (int t = lane; t < n / 2; t += 64) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const u64 x = buf[lo], y = buf[hi];
                const bool lt = sortkey(x) < sortkey(y);
                if (lt == desc) {
                    buf[lo] = y;
                    buf[hi] = x;
                }
            }
        }
    }
    wave_lds_fence();
}

// Whole block sorts n (power of two, runtime) packed entries in LDS, best first; one barrier per step.
template <int NT>
__device__ __forceinline__ void block_sort_desc_rt(u64* buf, int n, int tid) {
#pragma unroll 1
    for (int size = 2; size <= n; size <<= 1) {
#pragma unroll 1
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
#pragma unroll 1
            for (int t = tid; t < n / 2; t += NT) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const u64 x = buf[lo], y = buf[hi];
                const bool lt = sortkey(x) < sortkey(y);
                if (lt == desc) {
                    buf[lo] = y;
                    buf[hi] = x;
                }
            }
        }
    }
    __syncthreads();
}

// Whole block (NT threads) sorts CAP packed entries in LDS, best first.
// Compare-exchange steps whose stride stays inside one wave's CAP/(NT/64)-entry chunk run wave-locally
// (no block barrier, only compiler-level LDS ordering); only the log-many wide strides synchronise the
// block: 14 barriers instead of 91 for 8192 entries on 16 waves.
template <int CAP, int NT>
__device__ __forceinline__ void block_sort_desc(u64* buf, int tid) {
    constexpr int NW = NT / 64;
    constexpr int CH = CAP / NW;  // entries per wave chunk
    static_assert(CH >= 128 && (CH & (CH - 1)) == 0, "chunk must be a power of two >= 128");
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int base = wave * CH;
    bool need_block_sync = true;  // entries were written by arbitrary threads before the call
#pragma unroll 1
    for (int size = 2; size <= CAP; size <<= 1) {
#pragma unroll 1
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= CH) {
                __syncthreads();
                for (int t = tid; t < CAP / 2; t += NT) {
                    const int lo = 2 * t - (t & (stride - 1));
                    const int hi = lo + stride;
                    const bool desc = (lo & size) == 0;
                    const u64 x = buf[lo], y = buf[hi];
                    const bool lt = sortkey(x) < sortkey(y);
                    if (lt == desc) {
                        buf[lo] = y;
                        buf[hi] = x;
                    }
                }
                need_block_sync = true;
            } else {
                if (need_block_sync) {
                    __syncthreads();
                    need_block_sync = false;
                } else {
                    wave_lds_fence();
                }
                for (int t = lane; t < CH / 2; t += 64) {
                    const int lo = base + 2 * t - (t & (stride - 1));
                    const int hi = lo + stride;
                    const bool desc = (lo & size) == 0;
                    const u64 x = buf[lo], y = buf[hi];
                    const bool lt = sortkey(x) < sortkey(y);
                    if (lt == desc) {
                        buf[lo] = y;
                        buf[hi] = x;
                    }
                }
            }
        }
    }
    __syncthreads();
}

}  // namespace fsgpu
