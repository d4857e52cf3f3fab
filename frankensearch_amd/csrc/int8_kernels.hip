// int8_kernels.hip — the reference's production fast-tier path: int8 pass-1 scan + exact f16 rescore
// (VectorIndex::search_top_k_int8_two_pass, crates/frankensearch-index/src/search.rs:514-661).
//
//   slab quantiser   quantize_f16_le_bytes_to_i8_generic  simd.rs:1865-1886  (ONE corpus-wide max-abs scale,
//                    round() half away from zero, clamp +-127, NaN -> 0)
//   pass-1 dot       dot_i8_i8                            simd.rs:757,1240-1286 (exact i32)
//   pass-1 order     int8_heap_key / int8_heap_key_from_f32  search.rs:141-156 (score desc, row asc; the score is
//                    compared as f32 once dim > 1040)
// The int8 dot is integer-exact, so ANY summation order gives the reference's bits: v_dot4_i32_i8 over 16-byte
// lane chunks, quad-reduced with DPP adds.  Converting the i32 dot to f32 is exact up to dim 1040 and is exactly
// the reference's `score as f32` key beyond, so the same packed-f32 top-k machinery as the f16 scan serves both
// branches.  HBM traffic is N*dim bytes per pass — half of the f16 scan.
//
// The 4-bit two-pass (search_top_k_4bit_two_pass, search.rs:860-1000; simd.rs:1286-1556, 1886-1900, 2153-2215) is the
// same shape on a quarter of the bytes: packed signed nibbles (corpus-wide scale 7/max_abs), v_dot8_i32_i4, then the
// exact f16 rescore.  The kernels below are templated on the element width.
#include "scan_common.hpp"

namespace fsgpu {

using namespace scan_detail;

namespace {

__device__ __forceinline__ int quad_sum_i32(int v) {
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);
    return v;
}

__device__ __forceinline__ signed char quant_i8(float x, float scale) {
    float v = roundf(x * scale);  // half away from zero, like f32::round
    if (v != v) return 0;         // Rust `as i8` maps NaN to 0
    v = fminf(fmaxf(v, -127.0f), 127.0f);
    return (signed char)(int)v;
}

// nibble_of_4bit (simd.rs:1892-1896): round half away from zero, clamp +-7, NaN -> 0, two's complement low nibble
__device__ __forceinline__ uint32_t quant_nibble(float x, float scale) {
    float v = roundf(x * scale);
    if (v != v) return 0u;
    v = fminf(fmaxf(v, -7.0f), 7.0f);
    return (uint32_t)(int)v & 0xFu;
}

}  // namespace

// max |f16| over the slab (f32::max ignores NaN, so does fmaxf); result accumulated with an integer atomic max
// on the non-negative float's bits.
__global__ __launch_bounds__(256) void slab_maxabs_kernel(const unsigned short* __restrict__ slab, size_t n_values,
                                                          unsigned int* __restrict__ out_bits) {
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nvec = n_values / 8;
    const u32x4* v = reinterpret_cast<const u32x4*>(slab);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const half8 h = __builtin_bit_cast(half8, v[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf((float)h[j]));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = nvec * 8; i < n_values; ++i)
            m = fmaxf(m, fabsf((float)__builtin_bit_cast(_Float16, slab[i])));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void quantize_slab_i8_kernel(const unsigned short* __restrict__ slab, size_t n_values,
                                                               const unsigned int* __restrict__ max_bits,
                                                               signed char* __restrict__ out) {
    const float max_abs = __uint_as_float(*max_bits);
    const bool zero = !(max_abs > 0.0f);
    const float scale = zero ? 0.f : 127.0f / max_abs;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nvec = n_values / 8;
    const u32x4* v = reinterpret_cast<const u32x4*>(slab);
    typedef signed char i8x8 __attribute__((ext_vector_type(8)));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const half8 h = __builtin_bit_cast(half8, v[i]);
        i8x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = zero ? (signed char)0 : quant_i8((float)h[j], scale);
        reinterpret_cast<i8x8*>(out)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = nvec * 8; i < n_values; ++i)
            out[i] = zero ? (signed char)0 : quant_i8((float)__builtin_bit_cast(_Float16, slab[i]), scale);
}

// The 4-bit levels of pack_f16_le_bytes_to_4bit (simd.rs:2153-2215: scale 7/max_abs, 0 when max_abs <= 1e-9; round half away
// from zero, clamp +-7, NaN -> 0) kept one per BYTE: the batched 4-bit pass 1 runs them through the int8 matrix-core kernels
// (the nibble dot is the same integer either way, and that pass is bound by matrix instructions, not by bytes).
__global__ __launch_bounds__(256) void quantize_slab_4bit_levels_kernel(const unsigned short* __restrict__ slab, size_t n_values,
                                                                        const unsigned int* __restrict__ max_bits,
                                                                        signed char* __restrict__ out) {
    const float max_abs = __uint_as_float(*max_bits);
    const float scale = max_abs > 1e-9f ? 7.0f / max_abs : 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nvec = n_values / 8;
    const u32x4* v = reinterpret_cast<const u32x4*>(slab);
    typedef signed char i8x8 __attribute__((ext_vector_type(8)));
    auto level = [&](float x) { return (signed char)((int)(quant_nibble(x, scale) << 28) >> 28); };   // sign-extended nibble
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const half8 h = __builtin_bit_cast(half8, v[i]);
        i8x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = level((float)h[j]);
        reinterpret_cast<i8x8*>(out)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = nvec * 8; i < n_values; ++i) out[i] = level((float)__builtin_bit_cast(_Float16, slab[i]));
}

// What the int8 slab misses of the f16 slab, measured once per slab so that integer scores can serve as a PROVABLE
// filter for the exact search (mfma_scan.hip, "int8 filter").  With c = fl(127 / max_abs) — the f32 scale the quantiser
// used — every element is x = (r + eps) / c; this kernel bounds, over all rows, the quantisation error and the size of
// the integer rows:
//   out[0]  float bits of max_row sum_i (|fl(x c) - r| + 8e-6)^2   (8e-6 covers the rounding of the product x c)
//   out[1]  max_row sum_i |r_i|
//   out[2]  max_row sum_i r_i^2
//   out[3]  != 0 when the slab holds a NaN or an infinity (no bound exists then)
// One wave per row, grid-stride; upper bounds only (order-free), accumulated with integer atomic max.
__global__ __launch_bounds__(256) void i8_slab_stats_kernel(const unsigned short* __restrict__ slab,
                                                            const signed char* __restrict__ slab_i8, uint32_t nrows,
                                                            uint32_t dim, const unsigned int* __restrict__ max_bits,
                                                            unsigned int* __restrict__ out) {
    const float max_abs = __uint_as_float(*max_bits);
    const float scale = max_abs > 0.0f ? 127.0f / max_abs : 0.f;
    const int lane = threadIdx.x & 63;
    const uint32_t wave_gid = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * 256) >> 6;
    float e2max = 0.f;
    unsigned int r1max = 0, r2max = 0, bad = 0;
    for (uint32_t row = wave_gid; row < nrows; row += nwaves) {
        const _Float16* p = reinterpret_cast<const _Float16*>(slab) + (size_t)row * dim;
        const signed char* pi = slab_i8 + (size_t)row * dim;
        float e2 = 0.f;
        unsigned int r1 = 0, r2 = 0;
        for (uint32_t i = lane; i < dim; i += 64) {
            const float x = (float)p[i];
            const int r = (int)pi[i];
            if (!(fabsf(x) <= 65504.0f)) bad = 1;
            const float e = fabsf(x * scale - (float)r) + 8e-6f;
            e2 += e * e;
            r1 += (unsigned int)(r < 0 ? -r : r);
            r2 += (unsigned int)(r * r);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            e2 += __shfl_xor(e2, off);
            r1 += __shfl_xor(r1, off);
            r2 += __shfl_xor(r2, off);
        }
        e2max = fmaxf(e2max, e2);
        r1max = r1 > r1max ? r1 : r1max;
        r2max = r2 > r2max ? r2 : r2max;
    }
    bad = __any(bad) ? 1u : 0u;
    if (lane == 0) {
        if (e2max == e2max) atomicMax(&out[0], __float_as_uint(e2max));
        else bad = 1;
        atomicMax(&out[1], r1max);
        atomicMax(&out[2], r2max);
        if (bad) atomicOr(&out[3], 1u);
    }
}

// ---- the int8 FILTER's rotated copy (round 5) ---------------------------------------------------------------------------------
// One corpus-wide int8 scale is 127 / max|x|: a corpus with outlier channels (a few dimensions carrying most of every row's norm, as
// trained embedding models have) quantises its other dimensions to a handful of levels, and the filter's proven margin — fixed in
// integer units — grows in cosine units with 1 / (c_s c_q): 0.063 against 0.011 on the bench's two corpora, 1,000 rows within the
// margin of the k-th best against 40 (scripts/r05/rotation_bound_study.py).  Dot products are invariant under an orthogonal map,
// outlier channels are not: the filter's copy may hold quantised rows of R x, its queries R q — R a fixed random orthogonal matrix —
// and x . q = (R x) . (R q).  The map is applied in f64 and rounded ONCE to f32, so what it adds to the bound is 2^-24-sized
// (prepare_queries_i8_filter_kernel's extra_coeff); everything downstream — statistics, margin, thresholds — is the unrotated code on
// the rotated numbers, and the exact re-score that decides rows and score bits never sees the rotation.
//
// out[r][d] = fl32( sum_j R[d][j] in[r][j] ) with the sum in f64; rt = R transposed ([j][d]: a thread's loads are coalesced).
// One thread per output dimension, RB rows per block (their inputs in LDS as f64, read as broadcasts).  bad_to_nan (queries): a row
// with an element that is not finite or above 65,504 comes out as NaN, so that the filter marks it uncertifiable as it does unrotated.
template <typename IN, int RB>
__global__ void rotate_rows_kernel(const IN* __restrict__ in, uint32_t nrows, uint32_t in_stride, uint32_t dim,
                                   const double* __restrict__ rt, float* __restrict__ out, int bad_to_nan) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rot_smem[];
    double* xs = reinterpret_cast<double*>(rot_smem);   // [RB][dim]
    __shared__ int bad[RB];
    const uint32_t row0 = blockIdx.x * RB;
    const int d = threadIdx.x;
    for (int i = threadIdx.x; i < RB; i += blockDim.x) bad[i] = 0;   // (blockDim = dim may be smaller than RB)
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < RB * dim; i += blockDim.x) {
        const uint32_t rr = i / dim, j = i - rr * dim;
        const uint32_t row = row0 + rr < nrows ? row0 + rr : nrows - 1;
        const float v = (float)in[(size_t)row * in_stride + j];
        if (bad_to_nan && !(fabsf(v) <= 65504.0f)) bad[rr] = 1;
        xs[i] = (double)v;
    }
    __syncthreads();
    double acc[RB];
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) acc[rr] = 0.0;
    for (uint32_t j = 0; j < dim; ++j) {
        const double r = rt[(size_t)j * dim + d];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) acc[rr] = fma(r, xs[rr * dim + j], acc[rr]);
    }
#pragma unroll
    for (int rr = 0; rr < RB; ++rr)
        if (row0 + rr < nrows) out[(size_t)(row0 + rr) * dim + d] = bad[rr] ? __builtin_nanf("") : (float)acc[rr];
}

// max |v| over f32 values, ACCUMULATED into *out_bits (the caller zeroes it once): fmaxf ignores NaN like f32::max
__global__ __launch_bounds__(256) void maxabs_f32_kernel(const float* __restrict__ v, size_t n, unsigned int* __restrict__ out_bits) {
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(v[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));
}

// quantize_f16_le_bytes_to_i8_generic's rule (simd.rs:1865-1886: x * (127 / max), round half away, clamp, NaN -> 0) on f32 values
__global__ __launch_bounds__(256) void quantize_f32_i8_kernel(const float* __restrict__ v, size_t n, const unsigned int* __restrict__ max_bits,
                                                              signed char* __restrict__ out) {
    const float max_abs = __uint_as_float(*max_bits);
    const bool zero = !(max_abs > 0.0f);
    const float scale = zero ? 0.f : 127.0f / max_abs;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = zero ? (signed char)0 : quant_i8(v[i], scale);
}

// i8_slab_stats_kernel on f32 rows, ACCUMULATED into out (the caller zeroes it once)
__global__ __launch_bounds__(256) void i8_stats_f32_kernel(const float* __restrict__ rows, const signed char* __restrict__ rows_i8,
                                                           uint32_t nrows, uint32_t dim, const unsigned int* __restrict__ max_bits,
                                                           unsigned int* __restrict__ out) {
    const float max_abs = __uint_as_float(*max_bits);
    const float scale = max_abs > 0.0f ? 127.0f / max_abs : 0.f;
    const int lane = threadIdx.x & 63;
    const uint32_t wave_gid = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * 256) >> 6;
    float e2max = 0.f;
    unsigned int r1max = 0, r2max = 0, bad = 0;
    for (uint32_t row = wave_gid; row < nrows; row += nwaves) {
        const float* p = rows + (size_t)row * dim;
        const signed char* pi = rows_i8 + (size_t)row * dim;
        float e2 = 0.f;
        unsigned int r1 = 0, r2 = 0;
        for (uint32_t i = lane; i < dim; i += 64) {
            const float x = p[i];
            const int r = (int)pi[i];
            if (!(fabsf(x) <= 65504.0f)) bad = 1;
            const float e = fabsf(x * scale - (float)r) + 8e-6f;
            e2 += e * e;
            r1 += (unsigned int)(r < 0 ? -r : r);
            r2 += (unsigned int)(r * r);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            e2 += __shfl_xor(e2, off);
            r1 += __shfl_xor(r1, off);
            r2 += __shfl_xor(r2, off);
        }
        e2max = fmaxf(e2max, e2);
        r1max = r1 > r1max ? r1 : r1max;
        r2max = r2 > r2max ? r2 : r2max;
    }
    bad = __any(bad) ? 1u : 0u;
    if (lane == 0) {
        if (e2max == e2max) atomicMax(&out[0], __float_as_uint(e2max));
        else bad = 1;
        atomicMax(&out[1], r1max);
        atomicMax(&out[2], r2max);
        if (bad) atomicOr(&out[3], 1u);
    }
}

// pack_f16_le_bytes_to_4bit (simd.rs:2153-2215): scale 7/max_abs (0 when max_abs <= 1e-9), low nibble = even dim.
// Even dims: 8 values -> one 32-bit word anywhere in the slab (rows are whole bytes).
__global__ __launch_bounds__(256) void pack_slab_4bit_kernel(const unsigned short* __restrict__ slab, uint64_t count,
                                                             uint32_t dim, const unsigned int* __restrict__ max_bits,
                                                             unsigned char* __restrict__ out) {
    const float max_abs = __uint_as_float(*max_bits);
    const float scale = max_abs > 1e-9f ? 7.0f / max_abs : 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if ((dim & 7) == 0) {
        const size_t nvec = (size_t)count * dim / 8;
        const u32x4* v = reinterpret_cast<const u32x4*>(slab);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
            const half8 h = __builtin_bit_cast(half8, v[i]);
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) w |= quant_nibble((float)h[j], scale) << (4 * j);
            reinterpret_cast<uint32_t*>(out)[i] = w;
        }
    } else {  // any dim: one thread per output byte, rows padded to ceil(dim/2) bytes
        const uint32_t bpv = (dim + 1) / 2;
        const size_t nbytes = (size_t)count * bpv;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += stride) {
            const size_t row = i / bpv;
            const uint32_t d = (uint32_t)(i - row * bpv) * 2;
            const unsigned short* p = slab + row * dim + d;
            uint32_t b = quant_nibble((float)__builtin_bit_cast(_Float16, p[0]), scale);
            if (d + 1 < dim) b |= quant_nibble((float)__builtin_bit_cast(_Float16, p[1]), scale) << 4;
            out[i] = (unsigned char)b;
        }
    }
}

// Pass 1, fused with the wave top-k (row bytes % 64 == 0).  A lane is (row r, quarter a): chunk c = 4g+a is 16 bytes
// = 16 int8 (BITS 8, v_dot4_i32_i8) or 32 signed nibbles (BITS 4, v_dot8_i32_i4; both operands use the same
// low-nibble-first packing, so nibble j of a word meets nibble j of the query word).
template <int DIM, int KCAP, int BITS = 8>
__global__ __launch_bounds__(256) void scan_i8_topk_kernel(ScanArgs args, const signed char* __restrict__ slab_i8,
                                                           const signed char* __restrict__ query_i8) {
    constexpr int CAP = 2 * KCAP;
    constexpr int ROW_BYTES = DIM * BITS / 8;
    constexpr int G = ROW_BYTES / 64;
    static_assert(ROW_BYTES % 64 == 0, "row must be a whole number of 64-byte quad steps");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* bufs = reinterpret_cast<u64*>(smem);  // [wave][CAP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, a = lane & 3, r = lane >> 2;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 q[G];
#pragma unroll
    for (int g = 0; g < G; ++g) q[g] = reinterpret_cast<const i32x4*>(query_i8)[4 * g + a];
    WaveTopK<CAP> tk;
    tk.init(bufs + (size_t)wave * CAP);
    u64 thr = 0;
    const uint32_t nrows = args.nrows;
    const uint32_t ntiles = (nrows + kRowsPerTile - 1) / kRowsPerTile;
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    const int k = (int)args.k;
    auto load_tile = [&](uint32_t tile, i32x4 (&w)[G]) {
        uint32_t row = tile * kRowsPerTile + r;
        row = row < nrows ? row : nrows - 1;
        const i32x4* p = reinterpret_cast<const i32x4*>(slab_i8 + (size_t)row * ROW_BYTES) + a;
#pragma unroll
        for (int g = 0; g < G; ++g) w[g] = p[4 * g];
    };
    auto tile_words = [&](uint32_t tile, u64& live_word, u64& allow_word) {
        const uint32_t w64 = (tile * kRowsPerTile) >> 6;
        live_word = args.live ? args.live[w64] : ~0ull;
        allow_word = args.allow ? args.allow[w64] : ~0ull;
    };
    auto compute_tile = [&](uint32_t tile, const i32x4 (&w)[G], u64 live_word, u64 allow_word) {
        int acc = 0;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc = BITS == 8 ? __builtin_amdgcn_sdot4(w[g][j], q[g][j], acc, false)
                                : __builtin_amdgcn_sdot8(w[g][j], q[g][j], acc, false);
        const int dot = quad_sum_i32(acc);
        const uint32_t row = tile * kRowsPerTile + r;
        bool valid = row < nrows && a == 0;
        valid = valid && ((live_word >> (row & 63)) & 1ull) && ((allow_word >> (row & 63)) & 1ull);
        const u64 packed = pack((float)dot, args.row_base + row);
        bool cand = valid && sortkey(packed) > thr;
        u64 m = __ballot(cand);
        if (m == 0) return;
        if (tk.count + (int)__popcll(m) > CAP) {
            thr = tk.compact(k, lane);
            cand = cand && sortkey(packed) > thr;
            m = __ballot(cand);
        }
        if (cand) tk.buf[tk.count + (int)__popcll(m & ((1ull << lane) - 1ull))] = packed;
        tk.count += (int)__popcll(m);
    };
    {
        i32x4 wa[G], wb[G];
        u64 la = ~0ull, aa = ~0ull, lb = ~0ull, ab = ~0ull;
        uint32_t tile = blockIdx.x * kWavesPerBlock + wave;
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
    (void)tk.compact(k, lane);
    __syncthreads();
    if (wave == 0) {
        u64* dst = bufs;
        for (int w = 1; w < kWavesPerBlock; ++w) {
            const u64* src = bufs + (size_t)w * CAP;
            for (int i = lane; i < KCAP; i += 64) {
                const u64 xx = dst[i], yy = src[KCAP - 1 - i];
                dst[i] = sortkey(xx) >= sortkey(yy) ? xx : yy;
            }
            for (int i = KCAP + lane; i < CAP; i += 64) dst[i] = kEmpty;
            wave_sort_desc<CAP>(dst, lane);
        }
        u64* out = args.partial + (size_t)blockIdx.x * k;
        for (int i = lane; i < k; i += 64) out[i] = dst[i];
    }
}

// General pass 1: one packed entry per row (any dim; candidate counts beyond the fused tiers).
__global__ __launch_bounds__(256) void score_rows_i8_kernel(ScanArgs args, const signed char* __restrict__ slab_i8,
                                                            const signed char* __restrict__ query_i8,
                                                            u64* __restrict__ out_packed) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= args.nrows) return;
    bool valid = true;
    if (args.live) valid = valid && ((args.live[row >> 6] >> (row & 63)) & 1ull);
    if (args.allow) valid = valid && ((args.allow[row >> 6] >> (row & 63)) & 1ull);
    if (!valid) {
        out_packed[row] = kEmpty;
        return;
    }
    const signed char* p = slab_i8 + (size_t)row * args.dim;
    int dot = 0;
    for (uint32_t i = 0; i < args.dim; ++i) dot += (int)p[i] * (int)query_i8[i];
    out_packed[row] = pack((float)dot, args.row_base + row);
}

// General 4-bit pass 1: one packed entry per row (any dim).  dot_4bit_prepared_generic semantics (simd.rs:1519-1546).
__global__ __launch_bounds__(256) void score_rows_4bit_kernel(ScanArgs args, const unsigned char* __restrict__ slab_4bit,
                                                              const unsigned char* __restrict__ query_4bit,
                                                              u64* __restrict__ out_packed) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= args.nrows) return;
    bool valid = true;
    if (args.live) valid = valid && ((args.live[row >> 6] >> (row & 63)) & 1ull);
    if (args.allow) valid = valid && ((args.allow[row >> 6] >> (row & 63)) & 1ull);
    if (!valid) {
        out_packed[row] = kEmpty;
        return;
    }
    const uint32_t bpv = (args.dim + 1) / 2;
    const unsigned char* p = slab_4bit + (size_t)row * bpv;
    int dot = 0;
    for (uint32_t i = 0; i < bpv; ++i) {
        const int sl = (int)(((p[i] & 0xF) ^ 8) - 8), sh = (int)(((p[i] >> 4) ^ 8) - 8);
        const int ql = (int)(((query_4bit[i] & 0xF) ^ 8) - 8), qh = (int)(((query_4bit[i] >> 4) ^ 8) - 8);
        dot += sl * ql + sh * qh;
    }
    out_packed[row] = pack((float)dot, args.row_base + row);
}

// rows of packed entries -> u32 row ids (kEmpty -> 0xffffffff)
__global__ void packed_rows_kernel(const u64* __restrict__ packed, uint32_t n, uint32_t* __restrict__ rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rows[i] = (uint32_t)packed[i];
}

// (rows, exact scores) -> packed entries for the final merge; rows 0xffffffff stay kEmpty
__global__ void pack_hits_kernel(const uint32_t* __restrict__ rows, const float* __restrict__ scores, uint32_t n,
                                 u64* __restrict__ packed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) packed[i] = rows[i] == 0xffffffffu ? kEmpty : pack(scores[i], rows[i]);
}

// ---- launchers ----------------------------------------------------------------------------------------------

// One corpus-wide max-abs (simd.rs:1865-1886) into max_bits_dev (f32 bits).  A sharded index reduces the shards' values to the
// corpus-wide one (ncclAllReduce(max)) and then quantises with max_ready = true.
hipError_t launch_slab_maxabs(const void* slab_f16, size_t n_values, unsigned int* max_bits_dev, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(max_bits_dev, 0, 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(slab_maxabs_kernel, dim3(2048), dim3(256), 0, stream, static_cast<const unsigned short*>(slab_f16), n_values,
                       max_bits_dev);
    return hipGetLastError();
}

hipError_t launch_quantize_slab_i8(const void* slab_f16, size_t n_values, unsigned int* max_bits_dev, void* out_i8,
                                   hipStream_t stream, bool max_ready) {
    const int grid = 2048;
    if (!max_ready) {
        hipError_t e = launch_slab_maxabs(slab_f16, n_values, max_bits_dev, stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(quantize_slab_i8_kernel, dim3(grid), dim3(256), 0, stream,
                       static_cast<const unsigned short*>(slab_f16), n_values, max_bits_dev,
                       static_cast<signed char*>(out_i8));
    return hipGetLastError();
}

hipError_t launch_quantize_slab_4bit_levels(const void* slab_f16, size_t n_values, unsigned int* max_bits_dev, void* out_i8,
                                            hipStream_t stream, bool max_ready) {
    const int grid = 2048;
    if (!max_ready) {
        hipError_t e = launch_slab_maxabs(slab_f16, n_values, max_bits_dev, stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(quantize_slab_4bit_levels_kernel, dim3(grid), dim3(256), 0, stream,
                       static_cast<const unsigned short*>(slab_f16), n_values, max_bits_dev, static_cast<signed char*>(out_i8));
    return hipGetLastError();
}

hipError_t launch_i8_slab_stats(const void* slab_f16, const void* slab_i8, uint32_t nrows, uint32_t dim,
                                const unsigned int* max_bits_dev, unsigned int* stats_dev, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(stats_dev, 0, 16, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(i8_slab_stats_kernel, dim3(2048), dim3(256), 0, stream, static_cast<const unsigned short*>(slab_f16),
                       static_cast<const signed char*>(slab_i8), nrows, dim, max_bits_dev, stats_dev);
    return hipGetLastError();
}

template <typename IN>
static hipError_t launch_rotate_rows_t(const IN* in, uint32_t nrows, uint32_t in_stride, uint32_t dim, const double* rt, float* out,
                                       int bad_to_nan, hipStream_t stream) {
    constexpr int RB = 16;
    if (nrows == 0) return hipSuccess;
    if (dim == 0 || dim > 1024) return hipErrorInvalidValue;
    const size_t lds = (size_t)RB * dim * 8;
    auto kern = rotate_rows_kernel<IN, RB>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((nrows + RB - 1) / RB), dim3(dim), lds, stream, in, nrows, in_stride, dim, rt, out, bad_to_nan);
    return hipGetLastError();
}

hipError_t launch_rotate_rows_f16(const void* rows_f16, uint32_t nrows, uint32_t dim, const double* rt, float* out, hipStream_t stream) {
    return launch_rotate_rows_t(static_cast<const _Float16*>(rows_f16), nrows, dim, dim, rt, out, 0, stream);
}

hipError_t launch_rotate_rows_f32(const float* rows, uint32_t nrows, uint32_t row_stride, uint32_t dim, const double* rt, float* out,
                                  hipStream_t stream) {
    return launch_rotate_rows_t(rows, nrows, row_stride ? row_stride : dim, dim, rt, out, 1, stream);
}

hipError_t launch_maxabs_f32(const float* v, size_t n, unsigned int* max_bits_dev, hipStream_t stream) {
    hipLaunchKernelGGL(maxabs_f32_kernel, dim3(2048), dim3(256), 0, stream, v, n, max_bits_dev);
    return hipGetLastError();
}

hipError_t launch_quantize_f32_i8(const float* v, size_t n, const unsigned int* max_bits_dev, void* out_i8, hipStream_t stream) {
    hipLaunchKernelGGL(quantize_f32_i8_kernel, dim3(2048), dim3(256), 0, stream, v, n, max_bits_dev, static_cast<signed char*>(out_i8));
    return hipGetLastError();
}

hipError_t launch_i8_stats_f32(const float* rows, const void* rows_i8, uint32_t nrows, uint32_t dim, const unsigned int* max_bits_dev,
                               unsigned int* stats_dev, hipStream_t stream) {
    hipLaunchKernelGGL(i8_stats_f32_kernel, dim3(2048), dim3(256), 0, stream, rows, static_cast<const signed char*>(rows_i8), nrows, dim,
                       max_bits_dev, stats_dev);
    return hipGetLastError();
}

hipError_t launch_pack_slab_4bit(const void* slab_f16, uint64_t count, uint32_t dim, unsigned int* max_bits_dev,
                                 void* out_4bit, hipStream_t stream, bool max_ready) {
    const int grid = 2048;
    if (!max_ready) {
        hipError_t e = launch_slab_maxabs(slab_f16, (size_t)count * dim, max_bits_dev, stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(pack_slab_4bit_kernel, dim3(grid), dim3(256), 0, stream,
                       static_cast<const unsigned short*>(slab_f16), count, dim, max_bits_dev,
                       static_cast<unsigned char*>(out_4bit));
    return hipGetLastError();
}

bool scan_4bit_fused_supported(int dim, int kcap) {
    return (dim == 128 || dim == 256 || dim == 384 || dim == 512 || dim == 768) && (kcap == 64 || kcap == 256);
}

template <int DIM, int KCAP>
static hipError_t launch_4bit_t(const ScanArgs& args, const void* slab_4bit, const void* query_4bit, int grid,
                                hipStream_t stream, int* occupancy) {
    const size_t lds = (size_t)kWavesPerBlock * 2 * KCAP * 8;
    auto kern = scan_i8_topk_kernel<DIM, KCAP, 4>;
    if (occupancy) {
        int blocks = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kern, 256, lds) != hipSuccess || blocks < 1) blocks = 1;
        *occupancy = blocks;
        return hipSuccess;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, args, static_cast<const signed char*>(slab_4bit),
                       static_cast<const signed char*>(query_4bit));
    return hipGetLastError();
}

template <int KCAP>
static hipError_t launch_4bit_dim(const ScanArgs& args, const void* slab, const void* query, int grid, hipStream_t stream,
                                  int* occupancy) {
    switch (args.dim) {
        case 128: return launch_4bit_t<128, KCAP>(args, slab, query, grid, stream, occupancy);
        case 256: return launch_4bit_t<256, KCAP>(args, slab, query, grid, stream, occupancy);
        case 384: return launch_4bit_t<384, KCAP>(args, slab, query, grid, stream, occupancy);
        case 512: return launch_4bit_t<512, KCAP>(args, slab, query, grid, stream, occupancy);
        case 768: return launch_4bit_t<768, KCAP>(args, slab, query, grid, stream, occupancy);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_4bit(const ScanArgs& args, const void* slab_4bit, const void* query_4bit, int kcap, int grid,
                            hipStream_t stream, int* occupancy) {
    if (kcap == 64) return launch_4bit_dim<64>(args, slab_4bit, query_4bit, grid, stream, occupancy);
    if (kcap == 256) return launch_4bit_dim<256>(args, slab_4bit, query_4bit, grid, stream, occupancy);
    return hipErrorInvalidValue;
}

hipError_t launch_score_rows_4bit(const ScanArgs& args, const void* slab_4bit, const void* query_4bit, u64* out_packed,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(score_rows_4bit_kernel, dim3((args.nrows + 255) / 256), dim3(256), 0, stream, args,
                       static_cast<const unsigned char*>(slab_4bit), static_cast<const unsigned char*>(query_4bit),
                       out_packed);
    return hipGetLastError();
}

bool scan_i8_fused_supported(int dim, int kcap) {
    return (dim == 128 || dim == 256 || dim == 384 || dim == 512 || dim == 768) && (kcap == 64 || kcap == 256);
}

template <int DIM, int KCAP>
static hipError_t launch_i8_t(const ScanArgs& args, const void* slab_i8, const void* query_i8, int grid,
                              hipStream_t stream, int* occupancy) {
    const size_t lds = (size_t)kWavesPerBlock * 2 * KCAP * 8;
    auto kern = scan_i8_topk_kernel<DIM, KCAP>;
    if (occupancy) {
        int blocks = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kern, 256, lds) != hipSuccess || blocks < 1) blocks = 1;
        *occupancy = blocks;
        return hipSuccess;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, args, static_cast<const signed char*>(slab_i8),
                       static_cast<const signed char*>(query_i8));
    return hipGetLastError();
}

template <int KCAP>
static hipError_t launch_i8_dim(const ScanArgs& args, const void* slab_i8, const void* query_i8, int grid,
                                hipStream_t stream, int* occupancy) {
    switch (args.dim) {
        case 128: return launch_i8_t<128, KCAP>(args, slab_i8, query_i8, grid, stream, occupancy);
        case 256: return launch_i8_t<256, KCAP>(args, slab_i8, query_i8, grid, stream, occupancy);
        case 384: return launch_i8_t<384, KCAP>(args, slab_i8, query_i8, grid, stream, occupancy);
        case 512: return launch_i8_t<512, KCAP>(args, slab_i8, query_i8, grid, stream, occupancy);
        case 768: return launch_i8_t<768, KCAP>(args, slab_i8, query_i8, grid, stream, occupancy);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_i8(const ScanArgs& args, const void* slab_i8, const void* query_i8, int kcap, int grid,
                          hipStream_t stream, int* occupancy) {
    if (kcap == 64) return launch_i8_dim<64>(args, slab_i8, query_i8, grid, stream, occupancy);
    if (kcap == 256) return launch_i8_dim<256>(args, slab_i8, query_i8, grid, stream, occupancy);
    return hipErrorInvalidValue;
}

hipError_t launch_score_rows_i8(const ScanArgs& args, const void* slab_i8, const void* query_i8, u64* out_packed,
                                hipStream_t stream) {
    hipLaunchKernelGGL(score_rows_i8_kernel, dim3((args.nrows + 255) / 256), dim3(256), 0, stream, args,
                       static_cast<const signed char*>(slab_i8), static_cast<const signed char*>(query_i8), out_packed);
    return hipGetLastError();
}

hipError_t launch_packed_rows(const u64* packed, uint32_t n, uint32_t* rows, hipStream_t stream) {
    hipLaunchKernelGGL(packed_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, packed, n, rows);
    return hipGetLastError();
}

hipError_t launch_pack_hits(const uint32_t* rows, const float* scores, uint32_t n, u64* packed, hipStream_t stream) {
    hipLaunchKernelGGL(pack_hits_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, rows, scores, n, packed);
    return hipGetLastError();
}

}  // namespace fsgpu
