// vector_index_batched.cpp — the batched search of VectorIndex on the matrix cores (the north-star path): a proven-bound filter over
// the slab (f16 or the int8 copy), candidates re-scored by the exact kernels, so that the hits equal the exact scan's bit for bit.
//   plan -> per round { sample (group maxima) -> main pass -> selections } -> fallback for queries whose lists overflowed
// Also here: the begin / end halves (tickets), the quantised two-pass batch (search.rs:514-661, 876-946) and the int8 filter's
// copy of the slab (rotated when the slab has outlier channels).  Kernels: mfma_scan.hip, mfma_wide.hip, int8_kernels.hip.
#include "vector_index.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <type_traits>

#include "../../include/fsgpu.h"
#include "lab_env.hpp"
#include "vector_index_internal.hpp"

namespace fsgpu {

using namespace detail;

// Batched search on the matrix cores; see mfma_scan.hip for the error bound that makes the result exact.
SearchError VectorIndex::search_top_k_batched_device(const float* queries_dev, uint32_t nq, uint32_t query_len,
                                                     uint32_t k, const uint64_t* allow_dev, uint32_t* out_rows_dev,
                                                     float* out_scores_dev, uint32_t* out_counts_dev,
                                                     hipStream_t stream, uint32_t* fallbacks, uint64_t* out_packed_dev) {
    // Which approximate scores filter the slab: the int8 slab on the integer matrix cores (half the bytes, half the MFMA
    // instructions of the f16 filter; a wider proven margin) unless this index has shown that its margin lets too many rows
    // through (outlier dimensions stretch the corpus-wide int8 scale), the caller forced one, or the shape is not covered.
    const bool strided = row_stride_ && row_stride_ != dim_ * 2;
    bool i8f = batched_filter != 1 && (batched_filter == 2 || !i8f_disabled_) && !f32_ && !strided && variant == 0 && knobs().filter != 1 &&
               scan_mfma_supported((int)dim_) && k >= 1 && k <= 64 && nrows_ >= 4 * 8192ull;
    // a few queries are not worth BUILDING the int8 copy for; once it exists (or the host asked for the int8 latency path) they
    // are answered from it too: one query 0.88 ms against 1.29 ms on the exact kernel at 10M x 384
    if (batched_filter == 0 && knobs().filter == 0 && nq < 16 && !filter_ready() && !int8_latency) i8f = false;
    if (i8f && !filter_ready()) {
        // the int8 copy of the slab (half its size again; rotated when the slab has outlier channels) is built on first use; no room
        // for it: the f16 filter needs none
        FSGPU_TRY(ensure_filter_copy(stream));
        if (!filter_ready()) i8f = false;
    }
    if (i8f) {
        uint32_t refiltered = 0;
        SearchError e = batched_impl(queries_dev, nq, query_len, k, allow_dev, out_rows_dev, out_scores_dev, out_counts_dev, stream,
                                     fallbacks, out_packed_dev, 0, 0, true, &refiltered);
        if (e.ok() && async_want_ >= 0 && async_state_[async_want_] == 1) {
            async_i8f_[async_want_] = true;   // (the bookkeeping below happens in _end, once the verdicts are in)
            return e;
        }
        if (e.ok()) i8f_account(nq, refiltered);
        return e;
    }
    return batched_impl(queries_dev, nq, query_len, k, allow_dev, out_rows_dev, out_scores_dev, out_counts_dev, stream,
                        fallbacks, out_packed_dev, 0, 0, false, nullptr);
}

// The int8 copy of the slab and its statistics (what the certified lone-query pass and the int8 filter read), built NOW instead of by
// the first batched search: a row-sharded handle switches its shards to the int8 latency path in one go.
SearchError VectorIndex::prepare_int8_latency() {
    const bool strided = row_stride_ && row_stride_ != dim_ * 2;
    if (f32_ || strided || nrows_ < 4 * 8192ull || !scan_mfma_supported((int)dim_) || i8f_disabled_) return ok();
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_TRY(ensure_filter_copy(stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    return ok();
}

// What a batch's verdicts teach the index about its int8 filter.
void VectorIndex::i8f_account(uint32_t nq, uint32_t refiltered) {
    i8f_queries += nq;
    i8f_refiltered += refiltered;
    // What overflows on a corpus with a dense score tail (outlier dimensions, big clusters) is the MAIN pass's lists: the rows within
    // the margin of the k-th best number a few hundred whatever the corpus size, but the main pass runs on the threshold of a 1/25
    // sample and lets N / RB times as many through (scripts/r04/i8_bound_study.py).  A handful of leftovers already costs a pass of
    // their own over the f16 slab — as much as the 512 queries they came with —, so:
    //   * more than 1 in 64 of a wide batch handed on: the second sample of this index doubles, ONCE (half as many survivors for
    //     +0.1 ms of sampling per 512-query pass at 10M rows).  2 x is as far as it goes: on the anisotropic / Zipf corpus at 10M rows
    //     (scripts/r04/outlier_census.py) 150 of 1,024 queries were handed on at the base sample, 69 at 2 x, 580 at 4 x — the larger
    //     sample's own selection then overflows its candidate pool;
    //   * an eighth of a batch still handed on twice in a row: the index gives the int8 filter up for the f16 filter (unless the
    //     caller pinned the filter).
    // Since round 5 a slab with outlier channels gets a ROTATED int8 copy (ensure_filter_copy), which removes the usual cause.
    if (nq >= wide_min_queries() && i8f_sample_boost_ < 2 && (uint64_t)refiltered * 64 > nq) {
        i8f_sample_boost_ *= 2;
        i8f_strikes_ = 0;
    } else if (nq >= 16 && (uint64_t)refiltered * 8 > nq) {
        if (++i8f_strikes_ >= 2 && batched_filter == 0 && knobs().filter == 0) i8f_disabled_ = true;
    } else {
        i8f_strikes_ = 0;
    }
}

SearchError VectorIndex::int8_filter_bound(const float* queries, uint32_t nq, uint32_t query_len, float* out_delta,
                                           float* out_query_scale, float* out_slab_scale, int8_t* out_queries_i8, int8_t* out_slab_i8) {
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (f32_ || (row_stride_ && row_stride_ != dim_ * 2) || nrows_ == 0 || nrows_ > 0xffffffffull)
        return make_error(FSGPU_ERR_INVALID_CONFIG, "the int8 filter serves f16 slabs only");
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_TRY(ensure_filter_copy(stream_, true));
    if (!filter_ready()) return make_error(FSGPU_ERR_DEVICE, "no room for the int8 copy of the slab");
    float slab_max = 0.f;
    FSGPU_HIP(hipMemcpyAsync(&slab_max, filter_max(), 4, hipMemcpyDeviceToHost, stream_));
    std::vector<float> unit(nq, 0.f);
    if (nq) {
        FSGPU_TRY(ws_queries_.reserve((size_t)nq * dim_ * 4));
        FSGPU_TRY(mf_qh_.reserve((size_t)nq * dim_ * 2));
        FSGPU_TRY(mf_delta_.reserve((size_t)nq * 8));
        float* delta_dev = static_cast<float*>(mf_delta_.ptr);
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, queries, (size_t)nq * dim_ * 4, hipMemcpyHostToDevice, stream_));
        FSGPU_TRY(prepare_filter_queries(static_cast<const float*>(ws_queries_.ptr), nq, nq, dim_, mf_qh_.ptr, delta_dev, delta_dev + nq, stream_));
        if (out_delta) FSGPU_HIP(hipMemcpyAsync(out_delta, delta_dev, (size_t)nq * 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipMemcpyAsync(unit.data(), delta_dev + nq, (size_t)nq * 4, hipMemcpyDeviceToHost, stream_));
        if (out_queries_i8) FSGPU_HIP(hipMemcpyAsync(out_queries_i8, mf_qh_.ptr, (size_t)nq * dim_, hipMemcpyDeviceToHost, stream_));
    }
    if (out_slab_i8) FSGPU_HIP(hipMemcpyAsync(out_slab_i8, filter_slab(), (size_t)nrows_ * dim_, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    const float slab_scale = slab_max > 0.f ? 127.0f / slab_max : 0.f;
    if (out_slab_scale) *out_slab_scale = slab_scale;
    if (out_query_scale)
        for (uint32_t i = 0; i < nq; ++i) {
            if (i8f_rot_) {   // the scale of the ROTATED query: integer-score units per exact-score unit / the slab's scale
                out_query_scale[i] = slab_scale > 0.f ? unit[i] / slab_scale : 0.f;
                continue;
            }
            float m = 0.f;   // quantize_i8_query's scale, as the kernel computes it
            for (uint32_t d = 0; d < dim_; ++d) m = std::fmax(m, std::fabs(queries[(size_t)i * dim_ + d]));
            out_query_scale[i] = m > 0.f ? 127.0f / m : 0.f;
        }
    return ok();
}

// ---- the int8 filter's copy of the slab --------------------------------------------------------------------------------------
//
// Unrotated (the default for slabs without outlier channels): the reference's own int8 slab (quantize_f16_le_bytes_to_i8_generic),
// shared with the int8 two-pass search, + its statistics.  Rotated (round 5): a copy of its own — rows R x quantised with THEIR
// max-abs — for slabs whose largest element is far above what an even spread of a row's norm over its dimensions gives: the
// corpus-wide scale then wastes the int8 range on a few channels, and the filter's margin (fixed in integer units) is several times
// wider in cosine units than it has to be (int8_kernels.hip; scripts/r05/rotation_bound_study.py: 0.063 -> 0.021 on the bench's
// outlier corpus, 1,023 -> 83 rows within the margin of the k-th best).  Decided once per index, on first use.
namespace {
// a fixed random orthogonal matrix (seeded; modified Gram-Schmidt twice, in double) as its TRANSPOSE [j][d], and |R^T R - I|_F
void make_rotation(uint32_t dim, std::vector<double>& rt, double* ortho_err) {
    std::vector<double> r((size_t)dim * dim);
    uint64_t st = 0x9E3779B97F4A7C15ull ^ ((uint64_t)dim << 32);
    auto next = [&]() {   // splitmix64 -> two uniforms -> a normal (Box-Muller)
        auto u64 = [&]() {
            st += 0x9E3779B97F4A7C15ull;
            uint64_t z = st;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        const double u1 = ((double)(u64() >> 11) + 1.0) / 9007199254740993.0, u2 = (double)(u64() >> 11) / 9007199254740992.0;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    };
    for (double& v : r) v = next();
    for (int pass = 0; pass < 2; ++pass)
        for (uint32_t i = 0; i < dim; ++i) {
            double* ri = r.data() + (size_t)i * dim;
            for (uint32_t j = 0; j < i; ++j) {
                const double* rj = r.data() + (size_t)j * dim;
                double d = 0.0;
                for (uint32_t x = 0; x < dim; ++x) d += ri[x] * rj[x];
                for (uint32_t x = 0; x < dim; ++x) ri[x] -= d * rj[x];
            }
            double n = 0.0;
            for (uint32_t x = 0; x < dim; ++x) n += ri[x] * ri[x];
            n = 1.0 / std::sqrt(n);
            for (uint32_t x = 0; x < dim; ++x) ri[x] *= n;
        }
    // rows orthonormal <=> R R^T = I <=> R^T R = I; measured as |R R^T - I|_F (the two Frobenius norms agree for a square matrix
    // up to the conditioning, which is 1 + O(err) here)
    double err2 = 0.0;
    for (uint32_t i = 0; i < dim; ++i)
        for (uint32_t j = 0; j <= i; ++j) {
            double d = 0.0;
            for (uint32_t x = 0; x < dim; ++x) d += r[(size_t)i * dim + x] * r[(size_t)j * dim + x];
            d -= i == j ? 1.0 : 0.0;
            err2 += (i == j ? 1.0 : 2.0) * d * d;
        }
    *ortho_err = std::sqrt(err2);
    rt.resize((size_t)dim * dim);
    for (uint32_t d = 0; d < dim; ++d)
        for (uint32_t j = 0; j < dim; ++j) rt[(size_t)j * dim + d] = r[(size_t)d * dim + j];
}
}  // namespace

SearchError VectorIndex::ensure_filter_copy(hipStream_t stream, bool must) {
    if (filter_ready()) return ok();
    const bool strided = row_stride_ && row_stride_ != dim_ * 2;
    if (f32_ || strided || nrows_ == 0) return ok();
    FSGPU_HIP(hipSetDevice(device_));
    if (!i8f_decided_) {
        // rotate? the slab's largest |element| against the largest row norm spread evenly over the dimensions: a Gaussian-like row
        // has max ~ 6 / sqrt(dim) of its norm (and so has every rotated row), the bench's outlier corpus 17.6 / sqrt(dim)
        // (the rotation kernels run one thread per dimension and a block of dimensions per row: 64 <= dim <= 1024 in EVERY mode —
        // "always" on a shape outside it keeps the unrotated copy instead of failing every batched search: ADVICE r05)
        const bool rot_shape = dim_ >= 64 && dim_ <= 1024;
        bool rot = filter_rotation == 2 && rot_shape;
        if (filter_rotation == 0 && rot_shape) {
            FSGPU_TRY(i8f_max_.reserve(8));
            unsigned int* w = static_cast<unsigned int*>(i8f_max_.ptr);
            FSGPU_HIP(launch_slab_maxabs(slab_dev_, (size_t)nrows_ * dim_, w, stream));
            FSGPU_HIP(launch_max_row_norm(slab_dev_, (uint32_t)nrows_, dim_, 0, w + 1, stream));
            float host[2] = {0.f, 0.f};
            FSGPU_HIP(hipMemcpyAsync(host, w, 8, hipMemcpyDeviceToHost, stream));
            FSGPU_HIP(hipStreamSynchronize(stream));
            rot = host[1] > 0.f && std::isfinite(host[0]) && std::isfinite(host[1]) &&
                  (double)host[0] * std::sqrt((double)dim_) > kRotateRatio * (double)host[1];
        }
        i8f_rot_ = rot;
        i8f_decided_ = true;
    }
    if (!i8f_rot_) {
        if (!i8_ready_) {
            if (!i8_slab_.reserve((size_t)nrows_ * dim_).ok()) {   // no room for the copy: the f16 paths need none
                (void)hipGetLastError();
                if (must) return make_error(FSGPU_ERR_DEVICE, "no room for the int8 copy of the slab");
                i8f_disabled_ = true;
                return ok();
            }
            FSGPU_TRY(i8_max_.reserve(4));
            FSGPU_HIP(launch_quantize_slab_i8(slab_dev_, (size_t)nrows_ * dim_, static_cast<unsigned int*>(i8_max_.ptr), i8_slab_.ptr, stream,
                                              quant_max_ready_));
            i8_ready_ = true;
        }
        if (!i8_stats_ready_) {
            FSGPU_TRY(i8_stats_.reserve(16));
            FSGPU_HIP(launch_i8_slab_stats(slab_dev_, i8_slab_.ptr, (uint32_t)nrows_, dim_, static_cast<const unsigned int*>(i8_max_.ptr),
                                           static_cast<unsigned int*>(i8_stats_.ptr), stream));
            i8_stats_ready_ = true;
        }
        return ok();
    }
    // the rotated copy: R (f64, transposed) -> two passes over the slab in chunks of rows — max-abs of the rotated values, then
    // quantise + statistics — through a chunk-sized f32 staging buffer
    if (!i8f_slab_.reserve((size_t)nrows_ * dim_).ok()) {
        (void)hipGetLastError();
        if (must) return make_error(FSGPU_ERR_DEVICE, "no room for the int8 copy of the slab");
        i8f_disabled_ = true;
        return ok();
    }
    std::vector<double> rt;
    double ortho_err = 0.0;
    make_rotation(dim_, rt, &ortho_err);
    rot_extra_coeff_ = (ortho_err + 2.01 * 5.9604644775390625e-8) * 1.001;   // |R^T R - I| + 2.01 x 2^-24 (two roundings to f32)
    FSGPU_TRY(rot_mat_.reserve(rt.size() * 8));
    FSGPU_HIP(hipMemcpyAsync(rot_mat_.ptr, rt.data(), rt.size() * 8, hipMemcpyHostToDevice, stream));
    FSGPU_HIP(hipStreamSynchronize(stream));   // rt is a local
    const uint32_t chunk = (uint32_t)std::min<uint64_t>(nrows_, 1u << 18);
    DeviceBuffer tmp;
    FSGPU_TRY(tmp.reserve((size_t)chunk * dim_ * 4));
    FSGPU_TRY(i8f_max_.reserve(8));
    FSGPU_TRY(i8f_stats_.reserve(16));
    unsigned int* maxw = static_cast<unsigned int*>(i8f_max_.ptr);
    unsigned int* stats = static_cast<unsigned int*>(i8f_stats_.ptr);
    const double* rmat = static_cast<const double*>(rot_mat_.ptr);
    float* t32 = static_cast<float*>(tmp.ptr);
    const unsigned char* slab8 = static_cast<const unsigned char*>(slab_dev_);
    SearchError err;
    auto pass = [&](bool second) -> SearchError {
        for (uint64_t r0 = 0; r0 < nrows_; r0 += chunk) {
            const uint32_t n = (uint32_t)std::min<uint64_t>(chunk, nrows_ - r0);
            FSGPU_HIP(launch_rotate_rows_f16(slab8 + (size_t)r0 * dim_ * 2, n, dim_, rmat, t32, stream));
            if (!second) {
                FSGPU_HIP(launch_maxabs_f32(t32, (size_t)n * dim_, maxw, stream));
            } else {
                signed char* dst = static_cast<signed char*>(i8f_slab_.ptr) + (size_t)r0 * dim_;
                FSGPU_HIP(launch_quantize_f32_i8(t32, (size_t)n * dim_, maxw, dst, stream));
                FSGPU_HIP(launch_i8_stats_f32(t32, dst, n, dim_, maxw, stats, stream));
            }
        }
        return ok();
    };
    FSGPU_HIP(hipMemsetAsync(maxw, 0, 8, stream));
    FSGPU_HIP(hipMemsetAsync(stats, 0, 16, stream));
    err = pass(false);
    if (err.ok()) err = pass(true);
    FSGPU_HIP(hipStreamSynchronize(stream));   // the staging buffer goes away with this scope
    tmp.release();
    FSGPU_TRY(err);
    i8f_ready_ = true;
    return ok();
}

// The filter's queries: quantised as quantize_i8_query does + the proven bound delta (prepare_queries_i8_filter_kernel) — of the
// ROTATED queries when the filter's copy is (the same map in f64, rounded once to f32; what that adds to the bound: rot_extra_coeff_).
SearchError VectorIndex::prepare_filter_queries(const float* q, uint32_t nq, uint32_t nq_pad, uint32_t q_stride, void* qi8, float* delta,
                                                float* unit, hipStream_t stream) {
    if (!i8f_rot_) {
        FSGPU_HIP(launch_prepare_queries_i8_filter(q, nq, nq_pad, dim_, q_stride, filter_max(), filter_stats(), qi8, delta, stream, unit));
        return ok();
    }
    FSGPU_TRY(rot_q_.reserve((size_t)std::max<uint32_t>(nq, 1) * dim_ * 4));
    float* rq = static_cast<float*>(rot_q_.ptr);
    FSGPU_HIP(launch_rotate_rows_f32(q, nq, q_stride, dim_, static_cast<const double*>(rot_mat_.ptr), rq, stream));
    FSGPU_HIP(launch_prepare_queries_i8_filter(rq, nq, nq_pad, dim_, dim_, filter_max(), filter_stats(), qi8, delta, stream, unit,
                                               rot_extra_coeff_));
    return ok();
}

// The shard's own max-abs into the quantisers' scale word (device), for a sharded index to reduce across shards
// (ncclAllReduce(max), SURVEY 8f-1: the reference quantises with ONE corpus-wide scale, simd.rs:1865-1886).
SearchError VectorIndex::compute_local_quant_max(unsigned int** max_bits_dev, hipStream_t stream) {
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_TRY(i8_max_.reserve(4));
    if (nrows_ == 0 || f32_) FSGPU_HIP(hipMemsetAsync(i8_max_.ptr, 0, 4, stream));
    else FSGPU_HIP(launch_slab_maxabs(slab_dev_, (size_t)nrows_ * dim_, static_cast<unsigned int*>(i8_max_.ptr), stream));
    *max_bits_dev = static_cast<unsigned int*>(i8_max_.ptr);
    return ok();
}

// The scale word now holds the CORPUS-wide max-abs: every quantised copy is (re)built from it, on first use.
void VectorIndex::adopt_global_quant_max() {
    quant_max_ready_ = true;
    i8_ready_ = n4_ready_ = n4u_ready_ = i8_stats_ready_ = false;
}

// int8 pass 1 on the matrix cores for a whole batch (exact integer scores), exact f16 rescore, top-k: the batched form of
// search_top_k_int8_two_pass (search.rs:514-661).  multiplier 0 counts as 1, as in the reference.
// bits = 4: the batched form of search_top_k_4bit_two_pass (search.rs:876-946) — the same pipeline over the 4-bit levels, kept one
// per byte so that the int8 matrix-core kernels serve them (that pass is bound by matrix instructions, not by bytes).
SearchError VectorIndex::search_top_k_int8_batched_device(const float* queries_dev, uint32_t nq, uint32_t query_len,
                                                          uint32_t k, uint32_t multiplier, uint32_t* out_rows_dev,
                                                          float* out_scores_dev, uint32_t* out_counts_dev,
                                                          hipStream_t stream, uint32_t* fallbacks, int bits) {
    return batched_impl(queries_dev, nq, query_len, k, nullptr, out_rows_dev, out_scores_dev, out_counts_dev, stream,
                        fallbacks, nullptr, multiplier ? multiplier : 1, 0, false, nullptr, bits == 4 ? 4 : 8);
}

SearchError VectorIndex::two_pass_candidates_device(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                                    uint32_t multiplier, int bits, uint64_t* approx_out_dev, uint64_t* exact_out_dev,
                                                    hipStream_t stream, uint32_t* fallbacks) {
    int32_t ticket = -1;
    FSGPU_TRY(two_pass_candidates_device_begin(queries_dev, nq, query_len, k, multiplier, bits, approx_out_dev, exact_out_dev, stream, &ticket));
    FSGPU_TRY(two_pass_candidates_device_end(ticket, fallbacks));
    if (nq) {
        FSGPU_HIP(hipSetDevice(device_));
        FSGPU_HIP(hipStreamSynchronize(stream));
    }
    return ok();
}

// ... in two halves, like search_top_k_batched_device_begin / _end (the same two tickets): begin enqueues pass 1, the candidate
// selection and the exact re-score; end waits for that search's event and answers what the batch could not (list overflow: a pile of
// tied integer scores at the threshold) per query.  ticket -1: nothing was enqueued that end would have to wait for.
SearchError VectorIndex::two_pass_candidates_device_begin(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                                          uint32_t multiplier, int bits, uint64_t* approx_out_dev, uint64_t* exact_out_dev,
                                                          hipStream_t stream, int32_t* ticket) {
    *ticket = -1;
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (nq == 0) return ok();
    const uint64_t mult = multiplier ? multiplier : 1;
    const uint64_t cc = std::max<uint64_t>((uint64_t)k * mult, k);
    if (cc > 256 || k == 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "sharded two-pass: 1 <= k, k * multiplier <= 256");
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_HIP(hipMemsetAsync(approx_out_dev, 0xff, (size_t)nq * cc * 8, stream));
    FSGPU_HIP(hipMemsetAsync(exact_out_dev, 0xff, (size_t)nq * cc * 8, stream));
    if (f32_) return make_error(FSGPU_ERR_INVALID_CONFIG, "two-pass searches need an F16 slab");
    if (nrows_ == 0) return ok();
    int t = -1;
    for (int i = 0; i < 2; ++i)
        if (async_state_[i] == 0) {
            t = i;
            break;
        }
    if (t < 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "two begun batched searches are outstanding: end one first");
    // the shard-local top-k the pass also produces (not used by the root): one area per ticket
    DeviceBuffer& io = t == 0 ? mf_io_ : mf_io2_;
    FSGPU_TRY(io.reserve((size_t)nq * (k * 8 + 4)));
    uint32_t* rows = static_cast<uint32_t*>(io.ptr);
    float* scores = reinterpret_cast<float*>(rows + (size_t)nq * k);
    uint32_t* counts = reinterpret_cast<uint32_t*>(scores + (size_t)nq * k);
    async_state_[t] = 2;
    async_i8f_[t] = false;
    async_nq_[t] = nq;
    async_fb_[t] = 0;
    async_want_ = t;
    tp_approx_out_ = reinterpret_cast<u64*>(approx_out_dev);
    tp_exact_out_ = reinterpret_cast<u64*>(exact_out_dev);
    tp_stride_ = (uint32_t)cc;
    const SearchError e = batched_impl(queries_dev, nq, query_len, k, nullptr, rows, scores, counts, stream, &async_fb_[t], nullptr,
                                       (uint32_t)mult, 0, false, nullptr, bits == 4 ? 4 : 8);
    tp_approx_out_ = tp_exact_out_ = nullptr;
    tp_stride_ = 0;
    async_want_ = -1;
    if (!e.ok()) {
        async_state_[t] = 0;
        return e;
    }
    *ticket = t;
    return ok();
}

SearchError VectorIndex::two_pass_candidates_device_end(int32_t ticket, uint32_t* fallbacks, uint32_t* late_answers) {
    if (fallbacks) *fallbacks = 0;
    if (late_answers) *late_answers = 0;
    if (ticket < 0) return ok();
    return search_top_k_batched_device_end(ticket, fallbacks, late_answers);
}

// ---- the batched (matrix-core) search: prepare -> per round { sample -> main -> finish } -> fallback -------------------------
//
// int8_mult == 0: f16 slab, f16-rounded queries, approximate scores + proven margin (mfma_scan.hip header).
// int8_mult >= 1: int8 slab, int8 queries, exact integer scores; the k * int8_mult best rows are the candidates.
// i8_filter (int8_mult == 0): int8 slab and queries as the FILTER of the exact search — integer scores + the proven margin of
//                 prepare_queries_i8_filter_kernel; queries it cannot certify are re-filtered on the f16 path (*refiltered).

// What one call fixes for all its rounds: the arguments, the sample sizes, the workspaces.
struct VectorIndex::BatchedPlan {
    static constexpr uint32_t GMAX = 160;    // queries per pass: 128 (160 opt-in), or 64 for small batches / tails
    static constexpr uint32_t CAPQ = 8192;   // entries one selection pass covers: block lists + pool fit it at the wide shape
    static constexpr uint32_t SPILL = 4096;  // per-query overflow area for candidates that did not fit their block's list
    static constexpr uint32_t KC = kSelectPool;  // approximate candidates re-scored exactly (at most)
    static constexpr uint32_t RA_MAX = 8192;
    // arguments
    const float* queries_dev = nullptr;
    uint32_t nq = 0, query_len = 0, k = 0;
    const uint64_t* allow_dev = nullptr;
    uint32_t* out_rows_dev = nullptr;
    float* out_scores_dev = nullptr;
    uint32_t* out_counts_dev = nullptr;
    hipStream_t stream = nullptr;
    uint32_t* fallbacks = nullptr;
    uint64_t* out_packed_dev = nullptr;
    uint32_t int8_mult = 0, query_stride = 0;
    uint32_t* refiltered = nullptr;
    int bits = 8;
    // derived
    bool i8f = false, i8 = false, strided = false, skip_b = false, wide_ok = false;
    uint32_t qs = 0;                  // floats between queries
    uint32_t RA = 4096;               // stage A sample rows (dense; <= 8192)
    uint32_t RB = 131072;             // stage B sample rows (upper bound; shrinks with the slab)
    uint32_t ksel_est = 0, ksel = 0;  // the rank the selections anchor on (estimate incl. the int8 filter's growth; exact)
    uint32_t N = 0, QCAP = 0, wide_max = 0, k_eff = 0;
    int wide_pref = 3;
    // per-query verdicts, written by the kernels straight into pinned host memory and read after ONE stream synchronisation
    uint32_t *overflow_all = nullptr, *counts_all = nullptr;
    float *delta = nullptr, *tau = nullptr, *unit = nullptr, *tau_floor = nullptr;
    uint32_t* pool_flag = nullptr;
    u64 *spill = nullptr, *pool = nullptr;
    uint32_t* spill_count = nullptr;
    bool big_pool_last = false;       // the last round's finish had the second-chance launch (debug print only)
    // two_pass_candidates_device: where this batch leaves its candidate pairs (a parked plan's fallback needs them in _end too)
    u64 *tp_approx = nullptr, *tp_exact = nullptr;
    uint32_t tp_stride = 0;
};

// One round: up to QCAP queries — the sample stages and every selection are single launches over all its query groups, only the
// main pass is one launch per group.
struct VectorIndex::BatchedRound {
    uint32_t g0 = 0;                  // first query of the round
    int wide_qt = 0, shape = 0, wpb = 0, full_grid = 0, wide_grid = 0;
    uint32_t G = 0, wide_mult = 1, ngroups = 0, QP = 0, ng = 0, tile_rows = 0;
    const float* qg = nullptr;
    uint32_t *overflow = nullptr, *cand_counts = nullptr, *cand_count = nullptr;
    u64* cand = nullptr;
    MfmaScanArgs a{};
    SelectArgs sb{};
    bool anchor = false, short_stages = false;
    int grid_for(uint32_t rows, uint32_t tile) const {
        int g = (int)(((rows + tile - 1) / tile + wpb - 1) / wpb);
        if (g > full_grid) g = full_grid;
        return g < 1 ? 1 : g;
    }
    // one candidate list of `slots` entries per (query, block); 16..32 slots, sized so that lists + pool fit one selection pass
    // when the grid allows (the wide shape's 256 blocks do)
    uint32_t slots_for(int grid) const {
        return std::min<uint32_t>((uint32_t)scan_mfma_max_slots(shape),
                                  std::max<uint32_t>(16, (BatchedPlan::CAPQ - BatchedPlan::KC) / (uint32_t)grid));
    }
};

SearchError VectorIndex::batched_impl(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                      const uint64_t* allow_dev, uint32_t* out_rows_dev, float* out_scores_dev,
                                      uint32_t* out_counts_dev, hipStream_t stream, uint32_t* fallbacks,
                                      uint64_t* out_packed_dev, uint32_t int8_mult, uint32_t query_stride, bool i8_filter,
                                      uint32_t* refiltered, int bits) {
    BatchedPlan p;
    p.queries_dev = queries_dev;
    p.nq = nq;
    p.query_len = query_len;
    p.k = k;
    p.allow_dev = allow_dev;
    p.out_rows_dev = out_rows_dev;
    p.out_scores_dev = out_scores_dev;
    p.out_counts_dev = out_counts_dev;
    p.stream = stream;
    p.fallbacks = fallbacks;
    p.out_packed_dev = out_packed_dev;
    p.int8_mult = int8_mult;
    p.query_stride = query_stride;   // floats between queries (0 = dim): an MRL prefix view searches the first dim_ dimensions of full-length queries
    p.refiltered = refiltered;
    p.bits = bits;
    p.i8f = i8_filter && int8_mult == 0;
    p.i8 = int8_mult != 0 || p.i8f;
    p.tp_approx = tp_approx_out_;
    p.tp_exact = tp_exact_out_;
    p.tp_stride = tp_stride_;
    if (refiltered) *refiltered = 0;
    if (fallbacks) *fallbacks = 0;
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (nq == 0) return ok();
    // Every batched search of this index — begun or blocking — works in ONE set of device workspaces (thresholds, candidate lists,
    // spill areas, prepared queries): a search on another stream than an outstanding ticket's is ordered behind that ticket's last
    // kernel (searches on the same stream queue behind it by themselves).
    for (int t = 0; t < 2; ++t)
        if (async_state_[t] == 1 && async_stream_[t] != stream && async_ev_[t]) {
            FSGPU_HIP(hipSetDevice(device_));
            FSGPU_HIP(hipStreamWaitEvent(stream, async_ev_[t], 0));
        }
    bool done = false;
    FSGPU_TRY(batched_prepare(p, &done));
    if (done) return ok();
    for (uint32_t g0 = 0; g0 < nq;) {
        BatchedRound r;
        FSGPU_TRY(batched_round_setup(p, r, g0));
        FSGPU_TRY(batched_sample(p, r));
        FSGPU_TRY(batched_main(p, r));
        FSGPU_TRY(batched_finish(p, r));
        g0 += r.ng;
    }
    // everything of this search is enqueued: the caller's window for host work that should run under it (one shot, outer call only)
    if (after_enqueue_fn && !hard_batch_) {
        void (*fn)(void*) = after_enqueue_fn;
        after_enqueue_fn = nullptr;
        fn(after_enqueue_ctx);
    }
    if (async_want_ >= 0 && !hard_batch_) {
        // fsgpu_search_topk_batched_device_begin: everything is enqueued — the verdicts are read (and the rare uncertified query
        // answered) by _end, behind an event instead of a stream synchronisation, so that the caller can enqueue its next search first
        const int t = async_want_;
        static_assert(std::is_trivially_copyable<BatchedPlan>::value, "the parked plan is copied as bytes");
        async_plan_[t].resize(sizeof(BatchedPlan));
        std::memcpy(async_plan_[t].data(), &p, sizeof(BatchedPlan));
        // (a DEVICE-scope release: the default event makes the GPU write back and invalidate its caches where it is recorded — ~30 us
        // between this search's last kernel and the next search's first, the very gap the two halves exist to close.  What the host
        // reads behind the event are the verdicts, which the kernels write to coherent pinned memory; the outputs in device memory
        // are read by work that is ordered behind them on the GPU, or through copies that bring their own release.)
        if (!async_ev_[t]) FSGPU_HIP(hipEventCreateWithFlags(&async_ev_[t], hipEventDisableTiming | hipEventReleaseToDevice));
        FSGPU_HIP(hipEventRecord(async_ev_[t], p.stream));
        async_stream_[t] = p.stream;
        async_state_[t] = 1;
        return ok();
    }
    return batched_fallback(p);
}

// Stage "prepare": the sample sizes, the shapes the matrix-core path does not cover (answered here, *done = true), the lazily
// built quantised copies and statistics, the workspaces.
SearchError VectorIndex::batched_prepare(BatchedPlan& p, bool* done) {
    *done = false;
    const uint32_t nq = p.nq, k = p.k;
    p.qs = p.query_stride ? p.query_stride : dim_;
    uint32_t RA = 4096, RB = 131072;
    if (knobs().ra > 0) RA = (uint32_t)knobs().ra;  // tuning experiments only
    if (knobs().rb > 0) RB = (uint32_t)knobs().rb;
    constexpr uint32_t RA_MAX = BatchedPlan::RA_MAX;
    if (RA < 256 || RA > RA_MAX || (RA & 63)) RA = 4096;
    // Small slabs (a row shard of a multi-GPU index): a dense sample of 8192 rows already gives a threshold that lets
    // only ~k N / 8192 rows of the main pass through, so the second sampling stage (a launch plus a selection, ~55 us)
    // is skipped when that many candidates fit the block lists comfortably.
    bool skip_b = false;
    // (Not when the main pass is the register-resident-query kernel, i.e. for batches of 256 and more: a row that passes its
    // threshold costs that kernel's 160-instruction tile loop a divergent append, and the looser threshold of a skipped stage
    // B lets 4 x as many through — 1.25M-row shard, 1,024 queries: main pass 0.366 -> 0.329 ms, 2.5M: 0.741 -> 0.642 ms.)
    const bool wide_main = knobs().wide != 0 && nq >= wide_min_queries(p.i8 && !p.i8f) && scan_wide_supported((int)dim_, p.i8 ? 1 : 2) && variant != 5 && variant != 6;
    if (knobs().ra <= 0 && !knobs().no_skip_b && !wide_main && nrows_ <= 4'000'000 && nrows_ >= 4 * (uint64_t)RA_MAX) {
        const uint64_t expect = (uint64_t)std::max<uint32_t>(k, 1) * (p.i8 ? std::max<uint32_t>(p.int8_mult, 1) : 1) * (nrows_ / RA_MAX);
        if (expect <= 4096) {
            RA = RA_MAX;
            skip_b = true;
        }
    }
    // The main pass lets ~ksel N / RB rows through and stage B ~ksel RB / RA: both must stay in the low thousands (block
    // lists, spill area, the selection's capacity), so the samples grow with the rank the selections anchor on.
    // (the int8 filter's margin lets a few times as many rows through each stage as its rank alone would: sized like a larger rank)
    const uint32_t i8f_growth = knobs().i8f_growth > 0 ? (uint32_t)knobs().i8f_growth : 4;
    const uint32_t ksel_est = std::max<uint32_t>(k, 1) * (p.int8_mult ? p.int8_mult : 1) * (p.i8f ? i8f_growth : 1);
    const uint32_t grow = knobs().rb > 0 ? 1 : std::min<uint32_t>(4, (ksel_est + 15) / 16);
    if (knobs().ra <= 0 && ksel_est > 32) RA = RA_MAX;
    // B = about 1/64 of the slab (times the growth), between 8 RA and the cap, a multiple of RA, at most a quarter of it
    RB = std::min<uint32_t>(RB * grow, std::max<uint32_t>(8 * RA, (uint32_t)(nrows_ / 64) * grow));
    if (knobs().rb <= 0) {
        // ... and large enough that the main pass lets ~1,000 rows per query through (ksel N / RB): beyond that the per-block
        // lists and the spill area of the hottest queries overflow (50M rows, k = 10: 10 of 1,024 queries fell back to the
        // exact kernels with the 131,072-row cap, none with 488k — a sample pass of 0.4 ms per 1,024 queries next to 37 ms)
        // (bounded so that the sample stage's own survivors, ksel RB / RA, stay within its lists too: a rank of 90 — the int8
        // fast tier's 3 x 30 candidates — already runs a 524k-row sample)
        const uint64_t need = std::min<uint64_t>((uint64_t)ksel_est * nrows_ / 1024, (uint64_t)6000 * RA_MAX / ksel_est);
        if (need > RB) {
            RB = (uint32_t)std::min<uint64_t>(need, nrows_ / 8);
            if (knobs().ra <= 0) RA = RA_MAX;   // keeps the sample stage's own survivors (ksel RB / RA) in the hundreds
        }
    }
    // (an index whose int8 margin overflowed the main pass's lists samples more: its wide rounds gate the second sample by rank,
    // so that stage's own survivors stay in the hundreds — see search_top_k_batched_device)
    if (p.i8f && wide_main && i8f_sample_boost_ > 1 && knobs().rb <= 0) RB = (uint32_t)std::min<uint64_t>((uint64_t)RB * i8f_sample_boost_, nrows_ / 6);
    if (knobs().rb_pct > 0) RB = (uint32_t)std::min<uint64_t>((uint64_t)RB * (uint32_t)knobs().rb_pct / 100, nrows_ / 4);
    RB = std::min<uint32_t>(RB, (uint32_t)(nrows_ / 4));
    RB = std::max<uint32_t>(RA, RB / RA * RA);
    p.RA = RA;
    p.RB = RB;
    p.skip_b = skip_b;
    p.ksel_est = ksel_est;
    // int8 mode: candidate_count of the reference (search.rs:603-607)
    uint64_t cc64 = std::min<uint64_t>((uint64_t)k * (p.int8_mult ? p.int8_mult : 1), nrows_);
    cc64 = std::max<uint64_t>(cc64, std::min<uint64_t>(k, nrows_));
    p.ksel = p.int8_mult ? (uint32_t)std::min<uint64_t>(cc64, 0xffffffffull) : k;  // rank that anchors the selections
    p.strided = row_stride_ && row_stride_ != dim_ * 2;   // an MRL prefix view
    const bool usable = scan_mfma_supported((int)dim_) && k >= 1 && k <= 64 && p.ksel <= kSelectMaxK && nrows_ >= 4 * (uint64_t)RA && variant != 4 &&
                        !f32_ && (!p.strided || (!p.i8 && p.qs >= dim_)) && (p.query_stride == 0 || !p.i8);
    if (!usable) {
        *done = true;
        return batched_unusable(p);
    }
    FSGPU_HIP(hipSetDevice(device_));
    p.N = (uint32_t)nrows_;
    hipStream_t stream = p.stream;
    if (p.i8 && p.bits == 4 && !n4u_ready_) {  // the 4-bit levels of VectorIndex::nibbles_slab(), one per byte: built lazily, once
        FSGPU_TRY(n4u_slab_.reserve((size_t)nrows_ * dim_));
        FSGPU_TRY(i8_max_.reserve(4));
        FSGPU_HIP(launch_quantize_slab_4bit_levels(slab_dev_, (size_t)nrows_ * dim_, static_cast<unsigned int*>(i8_max_.ptr),
                                                   n4u_slab_.ptr, stream, quant_max_ready_));
        n4u_ready_ = true;
    }
    if (p.i8f) {   // the filter's copy (rotated for slabs with outlier channels) + its statistics: built lazily, once
        FSGPU_TRY(ensure_filter_copy(stream, true));
    } else if (p.i8 && p.bits != 4 && !i8_ready_) {  // VectorIndex::int8_slab(): built lazily, once
        FSGPU_TRY(i8_slab_.reserve((size_t)nrows_ * dim_));
        FSGPU_TRY(i8_max_.reserve(4));
        FSGPU_HIP(launch_quantize_slab_i8(slab_dev_, (size_t)nrows_ * dim_, static_cast<unsigned int*>(i8_max_.ptr),
                                          i8_slab_.ptr, stream, quant_max_ready_));
        i8_ready_ = true;
    }
    if (!p.i8 && !mf_norm_ready_) {
        FSGPU_TRY(mf_max_norm_.reserve(4));
        FSGPU_HIP(launch_max_row_norm(slab_dev_, p.N, dim_, p.strided ? row_stride_ : 0, static_cast<unsigned int*>(mf_max_norm_.ptr), stream));
        mf_norm_ready_ = true;
    }
    // A large batch is answered a "round" of up to QCAP queries at a time: the sample stages and every selection of
    // a round are single launches over all its query groups (one block per query: 1024 blocks fill the chip where a
    // group's 128 leave half the CUs idle), only the main pass is one launch per group.
    constexpr uint32_t GMAX = BatchedPlan::GMAX, SPILL = BatchedPlan::SPILL, KC = BatchedPlan::KC;
    const uint32_t round_cap = knobs().round >= (int)GMAX ? (uint32_t)knobs().round : 1024;  // tuning experiments only
    // (at least the 256 slots of the smallest register-resident-query launch: 65..128 two-pass queries ride one, padded)
    const uint32_t QCAP = std::min<uint32_t>(round_cap, std::max<uint32_t>(std::max<uint32_t>(GMAX, 256), (nq + 127) / 128 * 128));
    p.QCAP = QCAP;
    FSGPU_TRY(mf_qh_.reserve((size_t)QCAP * dim_ * 2));
    FSGPU_TRY(mf_delta_.reserve(QCAP * 4));
    FSGPU_TRY(mf_tau_.reserve(QCAP * 16));
    FSGPU_TRY(mf_spill_.reserve((size_t)QCAP * SPILL * 8 + (size_t)QCAP * kMfmaSpillCountStride * 4));
    FSGPU_TRY(mf_dense_.reserve((size_t)QCAP * RA_MAX * 8));
    FSGPU_TRY(mf_sel_.reserve((size_t)QCAP * KC * 8));
    if (mf_shape_ < 0) {
        mf_shape_ = 2;                // 128-query kernel shape (mfma_scan.hip)
        if (knobs().mfma_shape) mf_shape_ = knobs().mfma_shape;  // tuning experiments only
        if (mf_shape_ < 1 || mf_shape_ > 3) mf_shape_ = 2;
        MfmaScanArgs probe{};
        probe.dim = dim_;
        probe.stage = 2;  // the main-pass instantiation
        FSGPU_HIP(launch_scan_mfma(probe, 0, 1, stream, &mf_per_cu_narrow_));
        FSGPU_HIP(launch_scan_mfma(probe, mf_shape_, 1, stream, &mf_per_cu_wide_));
        probe.elem_bytes = 1;
        // (rounds 3-5 ran int8 rows on shape 4 — 64-row tiles, 24 KB in flight per wave — whose main-pass instantiation spills inside
        // its tile loop; round 6 measured shape 2 against it at 10M rows, 100..255 queries: 8-12 % faster on 256-dimension rows, 0-4 % on
        // 384 — profiles/r06/lds_query_shape_ab.txt)
        mf_shape_i8_ = 2;
        if (knobs().mfma_shape_i8) mf_shape_i8_ = knobs().mfma_shape_i8;  // tuning experiments only
        if (mf_shape_i8_ < 1 || mf_shape_i8_ > 4) mf_shape_i8_ = 2;
        FSGPU_HIP(launch_scan_mfma(probe, 0, 1, stream, &mf_per_cu_narrow_i8_));
        FSGPU_HIP(launch_scan_mfma(probe, mf_shape_i8_, 1, stream, &mf_per_cu_wide_i8_));
        // 160-query shape: measured 1.49 ms per pass at 10M x 384 (0.64 of HBM peak) against 1.26 ms at 128 queries
        // (0.75) — 7 % more queries per second, but the pass is no longer HBM-bound; opt-in (FSGPU_USE_160=1)
        mf_use_160_ = knobs().use_160;
        if (mf_use_160_) {   // (experiments builds only: the shipped library does not carry the shape)
            FSGPU_HIP(launch_scan_mfma(probe, 5, 1, stream, &mf_per_cu_160_i8_));
            probe.elem_bytes = 2;
            FSGPU_HIP(launch_scan_mfma(probe, 5, 1, stream, &mf_per_cu_160_));
        }
    }
    // the register-resident-query main pass (mfma_wide.hip): 256 queries per launch by default
    // (384 per launch when that many queries are left: the matrix pipe is the bound there and fewer passes leave it more of
    // the power budget — measured 148 k against 130 k queries/s at 10M x 384)
    p.wide_pref = knobs().wide >= 0 ? knobs().wide : 3;
    p.wide_max = (uint32_t)std::max(2, std::min(knobs().wide_max > 0 ? knobs().wide_max : 5, scan_wide_max_query_tiles((int)dim_, p.i8 ? 1 : 2)));
    p.wide_ok = (p.wide_pref == 2 || p.wide_pref == 3) && scan_wide_supported((int)dim_, p.i8 ? 1 : 2) && variant != 5 && variant != 6;
    // per-query verdicts, written by the kernels straight into pinned host memory and read after ONE stream
    // synchronisation for the whole batch: [0, cap) = overflow flags, [cap, 2 cap) = candidate counts
    const uint32_t flag_cap = (nq + GMAX - 1) / GMAX * GMAX + 256;   // (a round's query slots may run up to 255 past its real queries)
    if (flag_cap > mf_flags_cap_) {
        // (three areas: blocking calls, and one per begun search — a begun search's verdicts must survive the next call's reset)
        if (async_state_[0] == 1 || async_state_[1] == 1)
            return make_error(FSGPU_ERR_INVALID_CONFIG, "a begun batched search is outstanding: end it before searching with a larger batch");
        if (mf_flags_host_) (void)hipHostFree(mf_flags_host_);
        mf_flags_host_ = nullptr;
        mf_flags_cap_ = 0;
        FSGPU_HIP(hipHostMalloc(reinterpret_cast<void**>(&mf_flags_host_), (size_t)flag_cap * 8 * 3, hipHostMallocMapped));
        mf_flags_cap_ = flag_cap;
    }
    uint32_t* flags_area = mf_flags_host_ + (size_t)((async_want_ >= 0 && !hard_batch_) ? 1 + async_want_ : 0) * mf_flags_cap_ * 2;
    p.overflow_all = flags_area;
    p.counts_all = flags_area + mf_flags_cap_;
    std::memset(flags_area, 0, (size_t)mf_flags_cap_ * 8);
    p.delta = static_cast<float*>(mf_delta_.ptr);
    p.tau = static_cast<float*>(mf_tau_.ptr);
    p.unit = p.tau + QCAP;   // int8 filter: integer-score units per exact-score unit, per query
    p.pool_flag = reinterpret_cast<uint32_t*>(p.unit + QCAP);   // finish: candidates did not fit the pool (per query of the round)
    p.tau_floor = reinterpret_cast<float*>(p.pool_flag + QCAP);   // the first sample's proven threshold, kept next to a heuristic one
    p.spill = static_cast<u64*>(mf_spill_.ptr);
    p.spill_count = reinterpret_cast<uint32_t*>(p.spill + (size_t)QCAP * SPILL);
    p.pool = static_cast<u64*>(mf_sel_.ptr);
    p.k_eff = std::min<uint32_t>(k, p.N);
    return ok();
}

// Shapes the matrix-core path does not cover: answered by the per-query kernels (or handed to the f16 branch).
SearchError VectorIndex::batched_unusable(BatchedPlan& p) {
    const uint32_t nq = p.nq, k = p.k;
    hipStream_t stream = p.stream;
    if (p.i8f)   // the f16 branch sorts them out
        return batched_impl(p.queries_dev, nq, p.query_len, k, p.allow_dev, p.out_rows_dev, p.out_scores_dev, p.out_counts_dev, stream,
                            p.fallbacks, p.out_packed_dev, 0, p.query_stride, false, nullptr);
    if (p.i8) {
        // per-query int8 two-pass through host staging (rare shapes: huge candidate counts, tiny or odd-dimension slabs)
        std::vector<float> q((size_t)nq * dim_), sc((size_t)nq * k);
        std::vector<uint32_t> rw((size_t)nq * k, 0xffffffffu), cnt(nq);
        FSGPU_HIP(hipMemcpyAsync(q.data(), p.queries_dev, q.size() * 4, hipMemcpyDeviceToHost, stream));
        FSGPU_HIP(hipStreamSynchronize(stream));
        for (uint32_t i = 0; i < nq; ++i)
            FSGPU_TRY(quantized_two_pass(q.data() + (size_t)i * dim_, dim_, k, p.int8_mult, p.bits, rw.data() + (size_t)i * k,
                                         sc.data() + (size_t)i * k, &cnt[i], p.tp_approx ? p.tp_approx + (size_t)i * p.tp_stride : nullptr,
                                         p.tp_exact ? p.tp_exact + (size_t)i * p.tp_stride : nullptr));
        if (p.out_rows_dev) FSGPU_HIP(hipMemcpyAsync(p.out_rows_dev, rw.data(), rw.size() * 4, hipMemcpyHostToDevice, stream));
        if (p.out_scores_dev) FSGPU_HIP(hipMemcpyAsync(p.out_scores_dev, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, stream));
        if (p.out_counts_dev) FSGPU_HIP(hipMemcpyAsync(p.out_counts_dev, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice, stream));
        FSGPU_HIP(hipStreamSynchronize(stream));
        if (p.fallbacks) *p.fallbacks = nq;
        return ok();
    }
    if (p.query_stride)
        return make_error(FSGPU_ERR_INVALID_CONFIG, "strided queries need the matrix-core path (caller falls back per query)");
    if (p.fallbacks) *p.fallbacks = nq;
    if (p.out_packed_dev) {
        FSGPU_TRY(search_top_k_packed_device(p.queries_dev, nq, p.query_len, k, p.allow_dev, p.out_packed_dev, stream));
        if (!p.out_rows_dev) return ok();
    }
    return search_top_k_device(p.queries_dev, nq, p.query_len, k, p.allow_dev, p.out_rows_dev, p.out_scores_dev, p.out_counts_dev, stream);
}

// The geometry of the round that starts at query g0, the prepared (rounded / quantised) queries, the scan arguments every stage shares.
SearchError VectorIndex::batched_round_setup(const BatchedPlan& p, BatchedRound& r, uint32_t g0) {
    const bool i8 = p.i8;
    hipStream_t stream = p.stream;
    r.g0 = g0;
    const uint32_t left = p.nq - g0;
    // Main pass at 384 / 256 queries per launch (mfma_wide.hip: queries in registers, row tiles through an LDS-DMA
    // ring) when that many are left; the sample stages then run as sub-groups of 128 on the LDS-query kernel.
    r.wide_qt = 0;
    if (p.wide_ok && left >= wide_min_queries(p.i8 && !p.i8f)) {   // 128-query groups per launch
        // as many as the registers hold (f16 rows of 384 dimensions: 3, their int8 form: 5), the round's groups spread evenly
        // over its passes (8 groups: 3 + 3 + 2 on f16 rows, 4 + 4 on int8 rows)
        // (the last group may be partly padding: 129..255 queries ride ONE 256-slot pass — the LDS-query kernel answered them as
        // 128 + the rest in two passes over the slab, 1.69-1.91 ms against 0.92 per tier at 10M rows: the chunks the many-queries
        // engine forms for 129..255 callers)
        const uint32_t groups_left = std::min<uint32_t>((left + 127) / 128, p.QCAP / 128);
        const uint32_t passes = (groups_left + p.wide_max - 1) / p.wide_max;
        r.wide_qt = p.wide_pref == 2 ? 2 : std::max(2, (int)((groups_left + passes - 1) / passes));
    }
    // 160, 128 or 64 queries per pass
    // (the exact search's int8 filter takes the 128-slot shape for ANY batch: at 10M rows 16..64 queries cost 0.80-0.81 ms on the
    // 64-query shape against 0.69 ms for 65 on the 128-slot one at 256 dimensions, 1.01-1.04 against 0.95 at 384; the int8 two-pass,
    // whose threshold is a rank in the sample rather than an anchored exact score, measured the other way round — 0.83 against 0.87 —
    // and keeps the 64-query shape: profiles/r06/lds_query_shape_ab.txt)
    const bool narrow_ok = !(p.i8f && !knobs().narrow_i8f);
    r.shape = r.wide_qt ? (i8 ? mf_shape_i8_ : mf_shape_)
                        : ((left > 64 || !narrow_ok) && variant != 5) ? ((left > 128 && mf_use_160_) ? 5 : (i8 ? mf_shape_i8_ : mf_shape_)) : 0;
    r.G = (uint32_t)scan_mfma_query_tiles(r.shape) * 16;
    r.wide_mult = r.wide_qt ? (uint32_t)r.wide_qt : 1;   // sample groups per main-pass launch
    // this round: `ngroups` groups of G queries (the last one may be partly padding), QP query slots, ng real queries
    r.ngroups = left >= r.G ? std::min<uint32_t>(left / r.G, p.QCAP / r.G) : 1;
    if (r.wide_qt) {   // whole launches: the padded tail rounds up when the slots exist, down otherwise
        const uint32_t want = std::min<uint32_t>((left + r.G - 1) / r.G, p.QCAP / r.G);
        r.ngroups = want / r.wide_mult * r.wide_mult;
        if (r.ngroups < want && r.ngroups + r.wide_mult <= p.QCAP / r.G && left > r.ngroups * r.G) r.ngroups += r.wide_mult;
    }
    r.QP = r.ngroups * r.G;
    r.ng = std::min(r.QP, left);
    r.wpb = scan_mfma_waves_per_block(r.shape);
    const int per_cu = r.shape == 5 ? (i8 ? mf_per_cu_160_i8_ : mf_per_cu_160_)
                                    : (i8 ? (r.shape ? mf_per_cu_wide_i8_ : mf_per_cu_narrow_i8_)
                                          : (r.shape ? mf_per_cu_wide_ : mf_per_cu_narrow_));
    r.full_grid = num_cus_ * per_cu;
    r.tile_rows = (uint32_t)scan_mfma_rows_per_tile(r.shape);
    r.qg = p.queries_dev + (size_t)g0 * p.qs;
    r.overflow = p.overflow_all + g0;
    r.cand_counts = p.counts_all + g0;
    if (p.i8f)
        FSGPU_TRY(prepare_filter_queries(r.qg, r.ng, r.QP, p.qs, mf_qh_.ptr, p.delta, p.unit, stream));
    else if (i8) FSGPU_HIP(launch_prepare_queries_i8(r.qg, r.ng, r.QP, dim_, mf_qh_.ptr, p.delta, stream, p.bits));
    else
        FSGPU_HIP(launch_prepare_queries(r.qg, r.ng, r.QP, dim_, p.qs, static_cast<const unsigned int*>(mf_max_norm_.ptr),
                                         mf_qh_.ptr, p.delta, stream));
    r.wide_grid = num_cus_ * mf_per_cu_wide_main_;
    FSGPU_TRY(mf_cand_.reserve((size_t)r.QP * std::max(r.full_grid, r.wide_grid) * kMfmaMaxSlots * 8));
    r.cand = static_cast<u64*>(mf_cand_.ptr);
    FSGPU_TRY(mf_cand_count_.reserve((size_t)r.QP * r.wide_grid * 4));   // the wide kernels' list lengths (no padding)
    r.cand_count = static_cast<uint32_t*>(mf_cand_count_.ptr);
    MfmaScanArgs& a = r.a;
    a = MfmaScanArgs{};
    a.slab = p.i8f ? filter_slab() : i8 ? (p.bits == 4 ? n4u_slab_.ptr : i8_slab_.ptr) : slab_dev_;
    a.elem_bytes = i8 ? 1 : 2;
    a.live = reinterpret_cast<const u64*>(live_dev_);
    a.allow = reinterpret_cast<const u64*>(p.allow_dev);
    a.queries = mf_qh_.ptr;
    a.tau = p.tau;
    a.cand = r.cand;
    a.spill = p.spill;
    a.spill_count = p.spill_count;
    a.spill_cap = BatchedPlan::SPILL;
    a.overflow = r.overflow;
    a.dim = dim_;
    a.row_stride = p.strided ? row_stride_ : 0;
    a.row_base = (uint32_t)row_base_;
    a.nrows = p.N;
    // (a row shard's stages are short: the lists' padding and the selection's reads are a visible part of them — 8 / 16 slots
    // there, overflow goes to the spill area; 1.25M rows: 0.809 -> 0.789 ms per 1,024 queries, nothing at 10M)
    r.short_stages = nrows_ < 4'000'000;
    // int8 filter: the sample stages' thresholds are anchored on EXACT scores (their candidates are re-scored from the f16
    // slab right in the selection): one delta below the k-th best instead of two — the margin's multiplier on the rows each
    // stage lets through is exponential in it
    r.anchor = p.i8f && !knobs().no_anchor;
    return ok();
}

// Stage "sample": A = dense approximate scores of a small sample -> a first threshold; B = the rows of a larger sample at or above
// it, one short list per (query, block) -> the threshold the main pass runs with.  Samples are 64-row groups spread evenly over
// the slab: B = every stride_b-th group, A = a subset of B.
SearchError VectorIndex::batched_sample(const BatchedPlan& p, BatchedRound& r) {
    constexpr uint32_t SPILL = BatchedPlan::SPILL;
    hipStream_t stream = p.stream;
    const bool i8 = p.i8, i8f = p.i8f, skip_b = p.skip_b;
    const uint32_t N = p.N, RA = p.RA, RB = p.RB, QP = r.QP, ksel = p.ksel;
    MfmaScanArgs& a = r.a;
    const uint32_t groups_a = RA / 64, groups_b = RB / 64;
    const uint32_t stride_b = (N / 64) / groups_b;  // >= 4
    // The int8 filter's wide rounds (ranks up to kGroupsTaken): ONE sample pass that appends nothing — every block reports, per query,
    // its four best GROUPS of 8 rows (best approximate score | where), and the selection re-scores the rows of the best 24 groups from
    // the f16 slab: tau = max(a_k - 2 delta, S_k x unit - delta) exactly as the exact-anchor step below, with no first sample to gate
    // the second, no lists, no divergent append path in the sample's loop (stage A 0.05 ms + stage B 0.20 + its selection 0.07 per
    // 1,024 queries at 10M rows became 0.1 + 0.03).
    // (under a tombstone / allow bitmap the sample pass takes its maxima over live, allowed rows only: with the maxima over ALL rows a
    // group's best row was as likely filtered out as the bitmap is sparse, and the threshold anchored on what was left of 24 groups
    // let 1.7 x the rows through the main pass at 50 % allowed — 199 k against 308 k queries/s for the thresholded stages)
    // (the batched int8 / 4-bit two-pass takes the same pass: its pass-1 scores are the reference's own, so the k x multiplier-th best
    // group maximum IS a valid threshold — no re-score, ranks up to 64)
    // (round 6: ranks up to 32 with an anchor, up to 128 by rank — the two-tier flow's fetch of k x 3 = 30 per tier and the fast tier's
    // 3 x 30 = 90 int8 candidates took the two thresholded stages + two selections below: 0.8 ms of a 3.6 ms step at 10M rows)
    const bool rank_groups = i8 && !i8f && ksel <= kGroupsRankMax;
    const bool group_sample = (r.anchor ? ksel <= kGroupsTakenMax : rank_groups) && r.wide_qt != 0 && !skip_b && !knobs().no_wide_b &&
                              !knobs().no_group_sample && (dim_ & 7) == 0 && dim_ <= 1024 && scan_wide_group_maxima_supported((int)dim_, r.wide_qt);
    if (group_sample) {
        const int grid_g = std::min(r.wide_grid, (int)std::max<uint32_t>(1, RB / 64 / 4));   // at least 4 sample groups per block
        // (enough groups for the picks: the rank form needs ksel of them — and not all from a wave or two of the selection)
        // (the ranks round 6 added — above 24 with an anchor, above 64 by rank — only with the FULL 1,024-entry sample of a large slab: on a
        // 43k-row slab the 30th best of 160 group maxima is so loose a threshold that 40 % of a batch overflowed its lists and was
        // re-filtered on the f16 slab — results right, time wrong; scripts/fuzz_sharded.py found it through a second bug, see
        // search_top_k_batched_device_end's late_answers)
        const uint32_t entries = (uint32_t)grid_g * 4;
        const bool extended = r.anchor ? ksel > kGroupsTaken : ksel > 64;
        if (grid_g * 4 <= 1024 && entries >= (r.anchor ? 96u : 4u * std::min(ksel, 64u)) && (!extended || entries >= 1024)) {
            MfmaScanArgs c = a;
            c.dense = nullptr;
            c.stage = 3;
            c.group_stride = stride_b;
            c.group_count = groups_b;
            c.slots = 4;
            c.groups = r.ngroups / r.wide_mult;
            c.queries = mf_qh_.ptr;
            c.tau = p.tau;
            c.cand = r.cand;
            c.cand_count = nullptr;
            c.spill = p.spill;
            c.spill_count = p.spill_count;
            c.overflow = r.overflow;
            FSGPU_HIP(launch_scan_wide(c, r.wide_qt, grid_g, stream, nullptr));
            GroupSelectArgs g{};
            g.groups = r.cand;
            g.nentries = (uint32_t)grid_g * 4;
            g.k = ksel;
            g.delta = p.delta;
            g.anchor_unit = p.unit;
            g.tau_out = p.tau;
            g.overflow = r.overflow;
            g.spill_reset = p.spill_count;   // the main pass appends from zero
            g.slab = slab_dev_;
            g.live = a.live;
            g.allow = a.allow;
            g.queries = r.qg;
            g.dim = dim_;
            g.nrows = N;
            g.row_base = (uint32_t)row_base_;
            g.query_stride = p.qs;
            g.hreduce = hreduce;
            g.valid_queries = r.ng;
            g.rank_only = r.anchor ? 0u : 1u;
            FSGPU_HIP(launch_select_groups(g, (int)QP, stream));
            a.dense = nullptr;
            a.stage = 1;
            a.group_stride = stride_b;
            a.group_count = groups_b;
            a.groups = r.ngroups;
            SelectArgs& sb = r.sb;   // what the main pass and the finish expect from this stage
            sb = SelectArgs{};
            sb.lists = r.cand;
            sb.k = ksel;
            sb.take_topk = (i8 && !i8f) ? 1 : 0;
            sb.delta = p.delta;
            sb.overflow = r.overflow;
            sb.spill = p.spill;
            sb.spill_count = p.spill_count;
            sb.spill_cap = SPILL;
            return ok();
        }
    }
    // stage A: dense approximate scores of the A sample -> tau = (k-th best) - 2 delta
    a.dense = static_cast<u64*>(mf_dense_.ptr);
    a.stage = 0;
    a.group_stride = stride_b * (groups_b / groups_a);
    a.group_count = groups_a;
    a.slots = 0;
    a.groups = r.ngroups;
    FSGPU_HIP(launch_scan_mfma(a, r.shape, r.grid_for(RA, 16), stream, nullptr));
    SelectArgs sa{};
    sa.lists = a.dense;
    sa.q_stride = RA;
    sa.l_stride = RA;
    sa.nlists = 1;
    sa.list_len = RA;
    sa.k = ksel;
    sa.delta = p.delta;
    sa.tau_out = p.tau;
    auto set_rescore = [&](SelectArgs& x) {
        x.slab = slab_dev_;
        x.queries = r.qg;
        x.dim = dim_;
        x.row_stride = 0;
        x.query_stride = p.qs;
        x.nrows = N;
        x.row_base = (uint32_t)row_base_;
        x.hreduce = hreduce;
        x.k_out = p.k_eff;
    };
    // (a wide round's second sample only anchors the main pass's threshold — that pass visits every row — so it may take ANY
    // subset of the sample: the rows above the first sample's ~9th best score WITHOUT a margin, half as many as the proven
    // threshold lets through, and the first selection needs no exact re-score)
    // (the int8 two-pass takes the same shortcut: its threshold is the ksel-th best integer score of whatever subset came through)
    const bool heur_b = (r.anchor || (i8 && !i8f)) && r.wide_qt != 0 && !skip_b && !knobs().no_wide_b && !knobs().no_heur_b && ksel >= 8;
    if (heur_b) {
        // rank r of the first sample: the second sample holds RB / RA x as many rows above that score as the first (r, up to
        // an order statistic's spread ~ Gamma(r)), and k of them are needed — r = 5 + k / 8 puts "fewer than k came through"
        // (which only costs that query a looser threshold) near 1e-6 per query for k <= 64 and RB / RA >= 47; every rank less is
        // ~47 fewer appends per query in the append-bound sample pass (r = 9 -> 6 at k = 10: 0.7 % of a step, scripts/r03/sweep_rb.sh)
        sa.heur_rank = knobs().heur_rank > 0 ? (uint32_t)std::min<int>(knobs().heur_rank, (int)ksel) : std::min<uint32_t>(ksel, 5 + ksel / 8);
        sa.tau_floor_out = p.tau_floor;
    } else if (r.anchor) {
        set_rescore(sa);
        sa.anchor_unit = p.unit;
    }
    sa.valid_queries = r.ng;   // (the launch covers the round's padded query slots)
    sa.spill_reset = p.spill_count;   // the round's spill counters start at zero for the stage that follows (B, or the main pass)
    FSGPU_HIP(launch_select(sa, (int)QP, stream));
    // stage B: the B sample's rows at or above tau, one short list per (query, block) -> tighter tau; the rows
    // still at or above it form the pool carried into the last selection
    a.dense = nullptr;
    a.stage = 1;
    a.group_stride = stride_b;
    a.group_count = groups_b;
    const int grid_b = r.grid_for(RB, r.tile_rows);
    a.slots = r.slots_for(grid_b);
    // (a wide round samples on the register-resident-query kernel too: one launch per main-pass group, the same lists)
    const bool wide_b = r.wide_qt && !skip_b && !knobs().no_wide_b;
    const int wide_grid_b = std::min(r.wide_grid, (int)std::max<uint32_t>(1, RB / 64 / 4));   // at least 4 sample groups per block
    if (wide_b) a.slots = knobs().slots_b > 0 ? (uint32_t)knobs().slots_b : r.short_stages ? 8 : kWideSlots;
    if (!skip_b) {
        if (wide_b) {
            // ONE launch for all the round's main-pass groups (gridDim.y: group g's arrays follow group g - 1's, mfma_wide.hip)
            MfmaScanArgs c = a;
            c.groups = r.ngroups / r.wide_mult;
            c.queries = mf_qh_.ptr;
            c.tau = p.tau;
            c.cand = r.cand;
            c.cand_count = r.cand_count;
            c.spill = p.spill;
            c.spill_count = p.spill_count;
            c.overflow = r.overflow;
            FSGPU_HIP(launch_scan_wide(c, r.wide_qt, wide_grid_b, stream, nullptr));
        } else {
            FSGPU_HIP(launch_scan_mfma(a, r.shape, grid_b, stream, nullptr));
        }
    }
    const int lists_b = wide_b ? wide_grid_b : grid_b;
    SelectArgs& sb = r.sb;
    sb = SelectArgs{};
    sb.lists = r.cand;
    sb.q_stride = (uint64_t)lists_b * a.slots;
    sb.l_stride = a.slots;
    sb.nlists = (uint32_t)lists_b;
    sb.list_len = a.slots;
    sb.list_counts = wide_b ? r.cand_count : nullptr;
    sb.k = ksel;
    sb.take_topk = (i8 && !i8f) ? 1 : 0;
    sb.delta = p.delta;
    sb.overflow = r.overflow;
    sb.spill = p.spill;
    sb.spill_count = p.spill_count;
    sb.spill_cap = SPILL;
    if (!skip_b) {
        sb.tau_out = p.tau;
        sb.pool_out = p.pool;
        if (r.anchor) {
            set_rescore(sb);
            sb.anchor_unit = p.unit;
        }
        if (heur_b) sb.tau_floor_in = p.tau_floor;
        sb.valid_queries = r.ng;   // (the launch covers the round's padded query slots)
        sb.spill_reset = p.spill_count;   // ... and at zero again for the main pass
        FSGPU_HIP(launch_select(sb, (int)QP, stream));
        sb.spill_reset = nullptr;
        sb.valid_queries = 0;
        sb.anchor_unit = nullptr;
        sb.tau_floor_in = nullptr;
    }
    return ok();
}

// Stage "main": every row (the wide pass: all the round's query groups in ONE launch) or every group the B sample did not cover (the
// LDS-query kernel: one launch per query group) against the round's thresholds; candidates into per-(query, block) lists + the spill
// area.
// RULE: the spill counters must be zero on entry.  Nothing here clears them (a hipMemsetAsync per stage cost a stream bubble): every
// path into stage B or into this stage runs a selection launch with SelectArgs::spill_reset set first (batched_sample: the
// group-maxima selection, stage A's selection, stage B's selection).  FSGPU_DEBUG_BATCHED checks it.
SearchError VectorIndex::batched_main(const BatchedPlan& p, BatchedRound& r) {
    constexpr uint32_t SPILL = BatchedPlan::SPILL, CAPQ = BatchedPlan::CAPQ, KC = BatchedPlan::KC;
    hipStream_t stream = p.stream;
    MfmaScanArgs& a = r.a;
    SelectArgs& sb = r.sb;
    // (an event pair idles the stream ~6 us on either side of a launch: with a period, the main launches of every n-th call are timed)
    const bool profiling = this->profiling && (profile_period <= 1 || profile_tick_++ % (uint32_t)profile_period == 0);
    if (p.skip_b) {
        a.group_stride = 1;  // nothing was sampled by a stage B: the main pass visits every group
        a.group_count = 0;
    }
    // stage C: every group the B sample did not cover
    a.stage = 2;
    const int main_grid = r.wide_qt ? r.wide_grid : r.full_grid;
    if (r.wide_qt) {  // the wide main pass visits every row (no skip test in its loop): stage B only tightened tau
        a.group_stride = 1;
        a.group_count = 0;
    }
    // (the wide pass appends to global lists: 16 slots per (query, block) keep lists + pool inside one selection pass;
    // ranks above 32 — the int8 fast tier anchors on 90 — let ~1,700 rows per query through and get 32)
    a.slots = r.wide_qt ? (knobs().slots_main > 0 ? (uint32_t)knobs().slots_main
                           : p.ksel_est > 32 ? (r.short_stages && p.i8f ? 16 : kWideSlots)
                                             : std::min<uint32_t>(16, std::max<uint32_t>(8, (CAPQ - KC) / (uint32_t)main_grid)))
                        : r.slots_for(r.full_grid);
    if (knobs().debug_batched) {   // the rule above
        std::vector<uint32_t> counts((size_t)r.QP * kMfmaSpillCountStride);
        FSGPU_HIP(hipStreamSynchronize(stream));
        FSGPU_HIP(hipMemcpy(counts.data(), p.spill_count, counts.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < counts.size(); i += kMfmaSpillCountStride)
            if (counts[i] != 0) return make_error(FSGPU_ERR_DEVICE, "batched_main: spill counter of query slot " + std::to_string(i / kMfmaSpillCountStride) + " is not zero on entry");
    }
    a.groups = 1;
    const uint32_t GM = r.G * r.wide_mult;  // queries per main-pass launch
    // The register-resident-query kernel takes ALL the round's groups in one launch (gridDim.y = passes over the slab): a group's
    // blocks start on a CU as the previous group's block leaves it, so a step pays one launch ramp and one chip-wide tail instead of
    // one per 512 queries — what a 1.25M-row shard, whose pass is 0.2 ms, feels most.
#ifdef FSGPU_LAB_SPLIT_LAUNCHES   // lab: one launch per 512-query group, as before round 4 (same-box A/B of the merged launch)
    const uint32_t wide_groups = 0;
#else
    const uint32_t wide_groups = r.wide_qt ? r.ngroups / r.wide_mult : 0;
#endif
    if (wide_groups) {
        MfmaScanArgs c = a;
        c.groups = wide_groups;
        c.queries = mf_qh_.ptr;
        c.tau = p.tau;
        c.cand = r.cand;
        c.cand_count = r.cand_count;
        c.spill = p.spill;
        c.spill_count = p.spill_count;
        c.overflow = r.overflow;
        c.reverse = knobs().no_reverse ? 0 : (mf_pass_parity_ & 1);   // group g walks in direction (parity + g) & 1
        mf_pass_parity_ += wide_groups;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profiling) {
            FSGPU_HIP(hipEventCreateWithFlags(&e0, hipEventReleaseToDevice));   // (timing only: no cache write-back around the launch)
            FSGPU_HIP(hipEventCreateWithFlags(&e1, hipEventReleaseToDevice));
            FSGPU_HIP(hipEventRecord(e0, stream));
        }
        FSGPU_HIP(launch_scan_wide(c, r.wide_qt, main_grid, stream, nullptr));
        if (profiling) {
            FSGPU_HIP(hipEventRecord(e1, stream));
            events_.emplace_back(e0, e1);
            profiled_rows_ += (uint64_t)p.N * wide_groups;   // every group streams the whole slab
            profiled_elem_bytes_ = p.i8 ? 1 : 2;
        }
    }
    for (uint32_t j = 0; !wide_groups && j < r.ngroups / r.wide_mult; ++j) {
        MfmaScanArgs c = a;
        c.queries = static_cast<const unsigned char*>(mf_qh_.ptr) + (size_t)j * GM * dim_ * (p.i8 ? 1 : 2);
        c.tau = p.tau + (size_t)j * GM;
        c.cand = r.cand + (size_t)j * GM * main_grid * a.slots;
        c.cand_count = r.wide_qt ? r.cand_count + (size_t)j * GM * main_grid : nullptr;
        c.spill = p.spill + (size_t)j * GM * SPILL;
        c.spill_count = p.spill_count + (size_t)j * GM * kMfmaSpillCountStride;
        c.overflow = r.overflow + (size_t)j * GM;
        c.reverse = knobs().no_reverse ? 0 : (mf_pass_parity_++ & 1);  // consecutive passes alternate direction
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profiling) {
            FSGPU_HIP(hipEventCreate(&e0));
            FSGPU_HIP(hipEventCreate(&e1));
            FSGPU_HIP(hipEventRecord(e0, stream));
        }
        if (r.wide_qt) {
            c.groups = 1;
            FSGPU_HIP(launch_scan_wide(c, r.wide_qt, main_grid, stream, nullptr));
        } else {
            FSGPU_HIP(launch_scan_mfma(c, r.shape, r.full_grid, stream, nullptr));
        }
        if (profiling) {
            FSGPU_HIP(hipEventRecord(e1, stream));
            events_.emplace_back(e0, e1);
            profiled_rows_ += (p.skip_b || r.wide_qt) ? p.N : p.N - p.RB;
            profiled_elem_bytes_ = p.i8 ? 1 : 2;
        }
    }
    sb.q_stride = (uint64_t)main_grid * a.slots;
    sb.l_stride = a.slots;
    sb.nlists = (uint32_t)main_grid;
    sb.list_len = a.slots;
    sb.list_counts = r.wide_qt ? r.cand_count : nullptr;
    sb.extra = (p.skip_b || r.wide_qt) ? nullptr : p.pool;
    sb.extra_len = (p.skip_b || r.wide_qt) ? 0 : KC;
    sb.tau_out = nullptr;
    sb.pool_out = nullptr;
    return ok();
}

// Stage "finish": every row whose approximate score is within 2 delta of the k-th best (more than KC of them: the query goes to
// the exact path) is re-scored in the reference's order; the best k exact entries are the answer.
SearchError VectorIndex::batched_finish(BatchedPlan& p, BatchedRound& r) {
    hipStream_t stream = p.stream;
    SelectArgs& sb = r.sb;
    const uint32_t k = p.k, g0 = r.g0;
    sb.cand_counts = r.cand_counts;
    // the int8 filter's margin (and whatever it hands on to the f16 filter) can put thousands of rows within reach of the k-th
    // score: the finish re-scores up to 8,192 of them per query instead of 1,024
    const bool second_chance = (p.i8f || hard_batch_) && !knobs().no_big_pool;
    p.big_pool_last = second_chance;
    sb.pool_flag = second_chance ? p.pool_flag : nullptr;
    sb.slab = slab_dev_;
    sb.queries = r.qg;
    sb.dim = dim_;
    sb.row_stride = p.strided ? row_stride_ : 0;
    sb.query_stride = p.qs;
    sb.nrows = p.N;
    sb.row_base = (uint32_t)row_base_;
    sb.hreduce = hreduce;
    sb.k_out = p.k_eff;
    sb.out_stride = k;
    sb.out_rows = p.out_rows_dev ? p.out_rows_dev + (size_t)g0 * k : nullptr;
    sb.out_scores = p.out_scores_dev ? p.out_scores_dev + (size_t)g0 * k : nullptr;
    sb.out_counts = p.out_counts_dev ? p.out_counts_dev + g0 : nullptr;
    sb.out_packed = p.out_packed_dev ? reinterpret_cast<u64*>(p.out_packed_dev) + (size_t)g0 * k : nullptr;
    if (p.int8_mult && p.tp_approx) {   // a sharded index's shard: the candidate pairs themselves (two_pass_candidates_device)
        sb.cand_approx_out = p.tp_approx + (size_t)g0 * p.tp_stride;
        sb.cand_exact_out = p.tp_exact + (size_t)g0 * p.tp_stride;
        sb.cand_out_stride = p.tp_stride;
    }
#ifdef FSGPU_EXPERIMENTS
    static unsigned long long* sel_stamps = nullptr;   // FSGPU_SELECT_STAMPS=1: shader clocks of the phases of blocks 0, 256, 512, 768 of the finish
    if (fsgpu::lab_env("FSGPU_SELECT_STAMPS") && !sel_stamps) (void)hipHostMalloc(reinterpret_cast<void**>(&sel_stamps), 64 * 8, hipHostMallocMapped);
    if (sel_stamps) {
        if (sel_stamps[8]) {
            for (int b = 0; b < 4; ++b) {
                std::fprintf(stderr, "[select stamps] block %4d start %+8lld:", b * 256, (long long)(sel_stamps[b * 16] - sel_stamps[0]));
                for (int i = 1; i <= 8; ++i) std::fprintf(stderr, " %lld", (long long)(sel_stamps[b * 16 + i] - sel_stamps[b * 16]));
                std::fprintf(stderr, " nc=%lld\n", (long long)sel_stamps[b * 16 + 15]);
            }
        }
        std::memset(sel_stamps, 0, 64 * 8);
        sb.stamps = sel_stamps;
    }
#endif
    FSGPU_HIP(launch_select(sb, (int)r.ng, stream));
    sb.stamps = nullptr;
    if (second_chance) {   // queries whose candidates did not fit the pool: the sorted finish over the same lists (others return at once)
        sb.big_pool = 1;
        FSGPU_HIP(launch_select(sb, (int)r.ng, stream));
    }
    return ok();
}

// Stage "fallback": ONE stream synchronisation for the whole batch, then the host reads the per-query verdicts — margin / capacity
// overflow, or fewer than k candidates — and the uncertified queries are answered by the exact kernels (the int8 filter hands a
// larger set to the f16 filter first; the int8 two-pass to its per-query form).
SearchError VectorIndex::batched_fallback(BatchedPlan& p, bool already_waited) {
    constexpr uint32_t KC = BatchedPlan::KC;
    hipStream_t stream = p.stream;
    const uint32_t nq = p.nq, k = p.k, k_eff = p.k_eff;
    // (polling the stream with hipStreamQuery before blocking was measured: no change at 10M rows or on a 1.25M-row shard,
    // profiles/r04/step_overheads.txt — the runtime's wait is already an active one for waits this short)
    if (!already_waited) FSGPU_HIP(hipStreamSynchronize(stream));
    std::vector<uint32_t> fb;
    for (uint32_t i = 0; i < nq; ++i)
        if (p.overflow_all[i] || p.counts_all[i] < k_eff) fb.push_back(i);
    if (knobs().debug_batched) {
        uint32_t big = 0, slot = 0, few = 0, mx = 0;
        for (uint32_t i = 0; i < nq; ++i) {
            if (p.counts_all[i] > (p.big_pool_last ? 8192u : KC)) ++big;
            else if (p.overflow_all[i]) ++slot;
            if (p.counts_all[i] < k_eff) ++few;
            mx = std::max(mx, p.counts_all[i]);
        }
        std::fprintf(stderr, "[fsgpu batched] nq=%u k=%u fallbacks=%zu  pool_overflow=%u  slot_or_skip=%u  few=%u  max_cand=%u\n", nq, k,
                     fb.size(), big, slot, few, mx);
        for (size_t j = 0; j < fb.size() && j < 4; ++j)
            std::fprintf(stderr, "    query %u: overflow=%u candidates=%u\n", fb[j], p.overflow_all[fb[j]], p.counts_all[fb[j]]);
    }
    const uint32_t total_fallbacks = (uint32_t)fb.size();
    if (total_fallbacks && p.i8 && !p.i8f) {
        // list/spill overflow (a pile of tied scores at the threshold): the per-query int8 two-pass answers those
        std::vector<float> qh(dim_), sc(k);
        std::vector<uint32_t> rw(k);
        for (uint32_t i : fb) {
            uint32_t cnt = 0;
            FSGPU_HIP(hipMemcpyAsync(qh.data(), p.queries_dev + (size_t)i * dim_, (size_t)dim_ * 4, hipMemcpyDeviceToHost, stream));
            FSGPU_HIP(hipStreamSynchronize(stream));
            std::fill(rw.begin(), rw.end(), 0xffffffffu);
            FSGPU_TRY(quantized_two_pass(qh.data(), dim_, k, p.int8_mult, p.bits, rw.data(), sc.data(), &cnt,
                                         p.tp_approx ? p.tp_approx + (size_t)i * p.tp_stride : nullptr,
                                         p.tp_exact ? p.tp_exact + (size_t)i * p.tp_stride : nullptr));
            if (p.out_rows_dev) FSGPU_HIP(hipMemcpyAsync(p.out_rows_dev + (size_t)i * k, rw.data(), (size_t)k * 4, hipMemcpyHostToDevice, stream));
            if (p.out_scores_dev) FSGPU_HIP(hipMemcpyAsync(p.out_scores_dev + (size_t)i * k, sc.data(), (size_t)k * 4, hipMemcpyHostToDevice, stream));
            if (p.out_counts_dev) FSGPU_HIP(hipMemcpyAsync(p.out_counts_dev + i, &cnt, 4, hipMemcpyHostToDevice, stream));
            FSGPU_HIP(hipStreamSynchronize(stream));
        }
    } else if (total_fallbacks) {
        // compact the uncertified queries, answer them with the exact kernels (8 per pass), scatter the hits back
        const size_t nf = fb.size();
        auto align_up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
        const size_t o_idx = 0, o_q = align_up(o_idx + nf * 4, 256), o_rows = align_up(o_q + nf * dim_ * 4, 256),
                     o_scores = align_up(o_rows + nf * k * 4, 256), o_counts = align_up(o_scores + nf * k * 4, 256),
                     total = align_up(o_counts + nf * 4, 256);
        // (the int8 filter hands its leftovers to a nested f16-filter call, which may itself use mf_fallback_)
        DeviceBuffer& fbuf = p.i8f ? mf_fallback2_ : mf_fallback_;
        FSGPU_TRY(fbuf.reserve(total));
        unsigned char* base = static_cast<unsigned char*>(fbuf.ptr);
        uint32_t* idx_dev = reinterpret_cast<uint32_t*>(base + o_idx);
        float* q_dev = reinterpret_cast<float*>(base + o_q);
        uint32_t* rows_dev = reinterpret_cast<uint32_t*>(base + o_rows);
        float* scores_dev = reinterpret_cast<float*>(base + o_scores);
        uint32_t* counts_dev = reinterpret_cast<uint32_t*>(base + o_counts);
        FSGPU_HIP(hipMemcpyAsync(idx_dev, fb.data(), nf * 4, hipMemcpyHostToDevice, stream));
        FSGPU_HIP(hipStreamSynchronize(stream));  // fb is a stack-owned pageable buffer
        FSGPU_HIP(launch_gather_queries(p.queries_dev, idx_dev, (uint32_t)nf, dim_, p.qs, q_dev, stream));
        if (p.i8f && nf > 8) {
            // rows within the int8 margin of the k-th best did not fit the lists (or the query cannot be certified on the int8
            // slab at all): the f16 filter, whose margin is ~20 x narrower, answers these as a batch of its own
            uint32_t inner_fb = 0;
            hard_batch_ = true;
            const SearchError inner = batched_impl(q_dev, (uint32_t)nf, p.query_len, k, p.allow_dev, rows_dev, scores_dev, counts_dev, stream,
                                                   &inner_fb, nullptr, 0, 0, false, nullptr);
            hard_batch_ = false;
            FSGPU_TRY(inner);
            if (p.refiltered) *p.refiltered = (uint32_t)nf;
            if (p.fallbacks) *p.fallbacks = inner_fb;
            FSGPU_HIP(launch_scatter_hits(idx_dev, (uint32_t)nf, k, rows_dev, scores_dev, counts_dev, p.out_rows_dev,
                                          p.out_scores_dev, p.out_counts_dev, reinterpret_cast<u64*>(p.out_packed_dev), stream));
            return ok();
        }
        if (p.i8f && p.refiltered) *p.refiltered = (uint32_t)nf;
        FSGPU_TRY(fused_search(q_dev, (uint32_t)nf, k, k_eff, p.allow_dev, rows_dev, scores_dev, counts_dev, nullptr, stream));
        FSGPU_HIP(launch_scatter_hits(idx_dev, (uint32_t)nf, k, rows_dev, scores_dev, counts_dev, p.out_rows_dev,
                                      p.out_scores_dev, p.out_counts_dev, reinterpret_cast<u64*>(p.out_packed_dev), stream));
    }
    if (p.fallbacks) *p.fallbacks = total_fallbacks;
    return ok();
}

// The batched search in two halves (fsgpu_search_topk_batched_device_begin / _end): begin enqueues everything and returns a ticket;
// end waits for THAT search's last kernel (an event — not the stream, which may already hold the caller's next search), reads the
// verdicts and answers the rare uncertified query.  Two tickets at most; queries and outputs stay the caller's until end.
SearchError VectorIndex::search_top_k_batched_device_begin(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                                           const uint64_t* allow_dev, uint32_t* out_rows_dev, float* out_scores_dev,
                                                           uint32_t* out_counts_dev, hipStream_t stream, uint64_t* out_packed_dev,
                                                           int32_t* ticket) {
    int t = -1;
    for (int i = 0; i < 2; ++i)
        if (async_state_[i] == 0) {
            t = i;
            break;
        }
    if (t < 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "two begun batched searches are outstanding: end one first");
    async_state_[t] = 2;   // (complete unless batched_impl parks its plan: shapes answered by the per-query kernels finish inside)
    async_i8f_[t] = false;
    async_nq_[t] = nq;
    async_fb_[t] = 0;
    async_want_ = t;
    const SearchError e = search_top_k_batched_device(queries_dev, nq, query_len, k, allow_dev, out_rows_dev, out_scores_dev, out_counts_dev,
                                                      stream, &async_fb_[t], out_packed_dev);
    async_want_ = -1;
    if (!e.ok()) {
        async_state_[t] = 0;
        return e;
    }
    *ticket = t;
    return ok();
}

SearchError VectorIndex::search_top_k_batched_device_end(int32_t ticket, uint32_t* fallbacks, uint32_t* late_answers) {
    if (late_answers) *late_answers = 0;
    if (ticket < 0 || ticket > 1 || async_state_[ticket] == 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "no such begun batched search");
    const int t = ticket;
    if (async_state_[t] == 1) {
        FSGPU_HIP(hipSetDevice(device_));
        FSGPU_HIP(hipEventSynchronize(async_ev_[t]));
        BatchedPlan p;
        std::memcpy(&p, async_plan_[t].data(), sizeof(BatchedPlan));
        uint32_t refiltered = 0;
        p.refiltered = async_i8f_[t] ? &refiltered : nullptr;   // (begin's were the addresses of its own locals)
        p.fallbacks = &async_fb_[t];
        async_state_[t] = 0;   // (before the fallback: it may search again, blocking, on this index)
        FSGPU_TRY(batched_fallback(p, true));
        if (async_i8f_[t]) i8f_account(async_nq_[t], refiltered);
        // queries whose hits were written by work enqueued HERE, behind everything begin enqueued: answered by the exact kernels
        // (counted in *fallbacks) or handed by the int8 filter to the f16 filter (re-filtered: certified there, so NOT a fallback) — a
        // caller that chained work to begin's last kernel (a shard's exchange) has to chain it again behind these
        if (late_answers) *late_answers = async_fb_[t] + refiltered;
    }
    async_state_[t] = 0;
    if (fallbacks) *fallbacks = async_fb_[t];
    return ok();
}

SearchError VectorIndex::search_top_k_batched(const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                              const uint64_t* allow, uint32_t* out_rows, float* out_scores,
                                              uint32_t* out_counts, uint32_t* fallbacks, const uint64_t* allow_resident_dev,
                                              bool queries_on_device) {
    if (fallbacks) *fallbacks = 0;
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (nq == 0) return ok();
    if (k == 0 || nrows_ == 0) {
        for (uint32_t q = 0; q < nq; ++q) out_counts[q] = 0;
        return ok();
    }
    FSGPU_HIP(hipSetDevice(device_));
    const size_t qbytes = (size_t)nq * dim_ * 4;
    // results: rows | scores | counts adjacent in ONE device block, so that one copy brings them up (fetch_batched_results)
    auto align_up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
    const size_t o_rows = 0, o_scores = align_up((size_t)nq * k * 4, 256), o_counts = align_up(o_scores + (size_t)nq * k * 4, 256),
                 total = align_up(o_counts + (size_t)nq * 4, 256);
    FSGPU_TRY(ws_out_.reserve(total));
    unsigned char* base = static_cast<unsigned char*>(ws_out_.ptr);
    const float* q_dev = queries;
    if (!queries_on_device) {
        FSGPU_TRY(ws_queries_.reserve(qbytes));
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, queries, qbytes, hipMemcpyHostToDevice, stream_));
        q_dev = static_cast<const float*>(ws_queries_.ptr);
    }
    const uint64_t* allow_dev = allow ? allow_resident_dev : nullptr;
    if (allow && !allow_dev) {
        const size_t words = (size_t)((nrows_ + 63) / 64);
        FSGPU_TRY(ws_allow_.reserve(words * 8));
        FSGPU_HIP(hipMemcpyAsync(ws_allow_.ptr, allow, words * 8, hipMemcpyHostToDevice, stream_));
        allow_dev = static_cast<const uint64_t*>(ws_allow_.ptr);
    }
    FSGPU_TRY(search_top_k_batched_device(q_dev, nq, query_len, k, allow_dev, reinterpret_cast<uint32_t*>(base + o_rows),
                                          reinterpret_cast<float*>(base + o_scores), reinterpret_cast<uint32_t*>(base + o_counts), stream_,
                                          fallbacks));
    return fetch_batched_results(base, o_rows, o_scores, o_counts, total, nq, k, out_rows, out_scores, out_counts);
}

// The three result arrays of a batched search, adjacent in one device block, back to the caller: ONE copy into the index's pinned block
// and three host memcpys — three copies into the caller's (pageable) arrays are three staged transfers of ~25 us each, 8 % of a
// 64-query search at 10M rows.  Falls back to those when the pinned block cannot be had.  Synchronises the stream.
SearchError VectorIndex::fetch_batched_results(const unsigned char* base, size_t o_rows, size_t o_scores, size_t o_counts, size_t total,
                                               uint32_t nq, uint32_t k, uint32_t* out_rows, float* out_scores, uint32_t* out_counts) {
    const size_t span = total - o_rows;
    unsigned char* pin = static_cast<unsigned char*>(pinned_batch_io(span));
    if (pin) {
        FSGPU_HIP(hipMemcpyAsync(pin, base + o_rows, span, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipStreamSynchronize(stream_));
        std::memcpy(out_rows, pin, (size_t)nq * k * 4);
        std::memcpy(out_scores, pin + (o_scores - o_rows), (size_t)nq * k * 4);
        std::memcpy(out_counts, pin + (o_counts - o_rows), (size_t)nq * 4);
        return ok();
    }
    FSGPU_HIP(hipMemcpyAsync(out_rows, base + o_rows, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_scores, base + o_scores, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_counts, base + o_counts, (size_t)nq * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    return ok();
}

SearchError VectorIndex::search_top_k_int8_batched(const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                                   uint32_t multiplier, uint32_t* out_rows, float* out_scores,
                                                   uint32_t* out_counts, uint32_t* fallbacks, int bits, bool queries_on_device) {
    if (fallbacks) *fallbacks = 0;
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (nq == 0) return ok();
    // what the fast path does not cover goes through the per-query search, like the reference (search.rs:579-585);
    // an index with a doc-id table also does (resolve_hits dedups by doc id there)
    std::vector<float> host_copy;
    if (k == 0 || nrows_ == 0 || !wal_.empty() || has_doc_ids()) {
        if (queries_on_device) {   // (the per-query entry points take host vectors)
            host_copy.resize((size_t)nq * dim_);
            FSGPU_HIP(hipSetDevice(device_));
            FSGPU_HIP(hipMemcpy(host_copy.data(), queries, host_copy.size() * 4, hipMemcpyDeviceToHost));
            queries = host_copy.data();
        }
        for (uint32_t i = 0; i < nq; ++i)
            FSGPU_TRY(bits == 4 ? search_top_k_4bit_two_pass(queries + (size_t)i * dim_, query_len, k, multiplier,
                                                             out_rows + (size_t)i * k, out_scores + (size_t)i * k, &out_counts[i])
                                : search_top_k_int8_two_pass(queries + (size_t)i * dim_, query_len, k, multiplier,
                                                             out_rows + (size_t)i * k, out_scores + (size_t)i * k, &out_counts[i]));
        if (fallbacks) *fallbacks = nq;
        return ok();
    }
    FSGPU_HIP(hipSetDevice(device_));
    // dedicated staging: the per-query fallback inside reuses the ws_* workspaces
    auto align_up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
    const size_t o_q = 0, o_rows = align_up((size_t)nq * dim_ * 4, 256), o_scores = align_up(o_rows + (size_t)nq * k * 4, 256),
                 o_counts = align_up(o_scores + (size_t)nq * k * 4, 256), total = align_up(o_counts + (size_t)nq * 4, 256);
    FSGPU_TRY(mf_io_.reserve(total));
    unsigned char* base = static_cast<unsigned char*>(mf_io_.ptr);
    float* q_dev = reinterpret_cast<float*>(base + o_q);
    uint32_t* rows_dev = reinterpret_cast<uint32_t*>(base + o_rows);
    float* scores_dev = reinterpret_cast<float*>(base + o_scores);
    uint32_t* counts_dev = reinterpret_cast<uint32_t*>(base + o_counts);
    const float* q_in = queries;
    if (!queries_on_device) {
        FSGPU_HIP(hipMemcpyAsync(q_dev, queries, (size_t)nq * dim_ * 4, hipMemcpyHostToDevice, stream_));
        q_in = q_dev;
    }
    FSGPU_TRY(search_top_k_int8_batched_device(q_in, nq, query_len, k, multiplier, rows_dev, scores_dev, counts_dev,
                                               stream_, fallbacks, bits));
    return fetch_batched_results(base, o_rows, o_scores, o_counts, total, nq, k, out_rows, out_scores, out_counts);
}

}  // namespace fsgpu
