// vector_index_lone.cpp — VectorIndex, one query at a time: the certified int8 pass (default once the int8 copy exists), the exact
// kernels behind it, both as begin / end halves so that a sharded handle overlaps its shards; and the quantised two-pass searches
// (search.rs:514-661, 876-946) with their latency lanes.  Results equal the exact scan's bit for bit on every path.
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

// See vector_index.hpp.  Same results as the exact kernels, bit for bit: the candidates are re-scored in the reference's order
// (gather_dot_kernel) and the certificate is the int8 filter's proven bound (prepare_queries_i8_filter_kernel: the quantised query IS
// quantize_i8_query's, the slab IS quantize_f16_le_bytes_to_i8's) applied to the list's own scores: every true top-k row has
// idot >= idot_k - 2 delta, and the kept list is exactly the 256 largest idot.
SearchError VectorIndex::certified_i8_lone_query(const float* query, uint32_t k, uint32_t* out_rows, float* out_scores,
                                                 uint32_t* out_count, bool* certified) {
    *certified = false;
    bool enqueued = false;
    FSGPU_TRY(certified_i8_enqueue(query, k, &enqueued));
    if (!enqueued) return ok();
    return certified_i8_check(out_rows, out_scores, out_count, certified);
}

// Four launches behind one another, no copy (the query and every result live in the pinned staging block, which the kernels address
// directly); nothing is waited for:
//   prepare   the query quantised as the filter does + its proven bound delta
//   scan      the int8 copy, every block keeps its LK best (integer score, row) entries
//   cut       the best score any block may have DROPPED: the maximum over the full lists' last entries
//   finish    select_kernel: tau = (k-th best approximate score) - 2 delta, the entries at or above it re-scored in the reference's
//             order from the f16 slab, the k best exact entries out
// *enqueued = false: a shape the pass does not cover, nothing was launched.
SearchError VectorIndex::certified_i8_enqueue(const float* query, uint32_t k, bool* enqueued) {
    *enqueued = false;
    constexpr uint32_t LK = 32;
    const uint32_t k_eff = (uint32_t)std::min<uint64_t>(k, nrows_);
    if (k_eff == 0 || k_eff > LK || nrows_ < 4096 || (dim_ & 7) || !scan_i8_fused_supported((int)dim_, 64) || pinned_io() == nullptr) return ok();
    const size_t qbytes = (size_t)dim_ * 4;
    const size_t o_out = (qbytes + 255) & ~(size_t)255, o_flags = (o_out + (size_t)k * 8 + 4 + 255) & ~(size_t)255;
    if (o_flags + 64 > kPinnedIoBytes) return ok();
    unsigned char* io = static_cast<unsigned char*>(io_host_);
    FSGPU_TRY(ws_i8_query_.reserve(dim_));
    std::memcpy(io, query, qbytes);
    const float* q_pin = reinterpret_cast<const float*>(io);
    float* delta_pin = reinterpret_cast<float*>(io + o_flags);
    float* tau_pin = delta_pin + 1;
    float* cut_pin = delta_pin + 2;
    uint32_t* ncand_pin = reinterpret_cast<uint32_t*>(delta_pin + 3);
    uint32_t* overflow_pin = reinterpret_cast<uint32_t*>(delta_pin + 4);
    *overflow_pin = 0;
    *ncand_pin = 0;
    FSGPU_TRY(prepare_filter_queries(q_pin, 1, 1, dim_, ws_i8_query_.ptr, delta_pin, nullptr, stream_));
    ScanArgs a = base_args(q_pin, nullptr);
    int per_cu = 1;
    FSGPU_HIP(launch_scan_i8(a, filter_slab(), ws_i8_query_.ptr, 64, 1, stream_, &per_cu));
    (void)per_cu;   // one block per CU: 256 lists x 32 entries are ONE pass of the finish (8,192 entries)
    int grid = num_cus_;
    const int max_useful = (int)(((nrows_ + 15) / 16 + 3) / 4);
    grid = std::max(1, std::min(grid, max_useful));
    FSGPU_TRY(ws_partial_.reserve((size_t)grid * LK * 8));
    a.partial = static_cast<u64*>(ws_partial_.ptr);
    a.k = LK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (profiling) {
        FSGPU_HIP(hipEventCreate(&e0));
        FSGPU_HIP(hipEventCreate(&e1));
        FSGPU_HIP(hipEventRecord(e0, stream_));
    }
    FSGPU_HIP(launch_scan_i8(a, filter_slab(), ws_i8_query_.ptr, 64, grid, stream_, nullptr));
    if (profiling) {
        FSGPU_HIP(hipEventRecord(e1, stream_));
        events_.emplace_back(e0, e1);
        profiled_rows_ += nrows_;
        profiled_elem_bytes_ = 1;
    }
    FSGPU_HIP(launch_list_cut(a.partial, (uint32_t)grid, LK, cut_pin, stream_));
    SelectArgs f{};
    f.lists = a.partial;
    f.q_stride = (uint64_t)grid * LK;
    f.l_stride = LK;
    f.nlists = (uint32_t)grid;
    f.list_len = LK;
    f.k = k_eff;
    f.delta = delta_pin;
    f.tau_out = tau_pin;
    f.cand_counts = ncand_pin;
    f.overflow = overflow_pin;
    f.slab = slab_dev_;
    f.queries = q_pin;
    f.dim = dim_;
    f.nrows = (uint32_t)nrows_;
    f.row_base = (uint32_t)row_base_;
    f.row_stride = (row_stride_ && row_stride_ != dim_ * 2) ? row_stride_ : 0;
    f.hreduce = hreduce;
    f.k_out = k_eff;
    f.out_stride = k;
    f.out_rows = reinterpret_cast<uint32_t*>(io + o_out);
    f.out_scores = reinterpret_cast<float*>(io + o_out + (size_t)k * 4);
    f.out_counts = reinterpret_cast<uint32_t*>(io + o_out + (size_t)k * 8);
    FSGPU_HIP(launch_select(f, 1, stream_));
    cert_k_ = k;
    *enqueued = true;
    return ok();
}

// The other half: ONE synchronisation, then the certificate.  The answer is the exact search's when every row whose approximate score
// reaches tau was in some list: cut < tau (a list that is not full dropped nothing), no more candidates than the finish holds,
// delta >= 0.  Otherwise nothing is written and the caller's staged path answers.
SearchError VectorIndex::certified_i8_check(uint32_t* out_rows, float* out_scores, uint32_t* out_count, bool* certified) {
    *certified = false;
    const uint32_t k = cert_k_;
    const uint32_t k_eff = (uint32_t)std::min<uint64_t>(k, nrows_);
    const size_t qbytes = (size_t)dim_ * 4;
    const size_t o_out = (qbytes + 255) & ~(size_t)255, o_flags = (o_out + (size_t)k * 8 + 4 + 255) & ~(size_t)255;
    unsigned char* io = static_cast<unsigned char*>(io_host_);
    const float* delta_pin = reinterpret_cast<const float*>(io + o_flags);
    const uint32_t* ncand_pin = reinterpret_cast<const uint32_t*>(delta_pin + 3);
    const uint32_t* overflow_pin = reinterpret_cast<const uint32_t*>(delta_pin + 4);
    const uint32_t* rows_pin = reinterpret_cast<const uint32_t*>(io + o_out);
    const float* scores_pin = reinterpret_cast<const float*>(io + o_out + (size_t)k * 4);
    const uint32_t* count_pin = reinterpret_cast<const uint32_t*>(io + o_out + (size_t)k * 8);
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    const float delta = delta_pin[0], tau = delta_pin[1], cut = delta_pin[2];
    if (!(delta >= 0.f)) return ok();   // a query the bound cannot cover (zero, non-finite, a slab with non-finite values)
    if (*overflow_pin != 0 || *ncand_pin > kSelectPool) return ok();   // more rows within the margin than the finish re-scores
    if (!(cut < tau)) return ok();      // a block may have dropped a row within the margin (NaN compares false: not certified)
    if (*count_pin < k_eff) return ok();
    std::memcpy(out_rows, rows_pin, (size_t)k * 4);
    std::memcpy(out_scores, scores_pin, (size_t)k * 4);
    *out_count = *count_pin;
    ++i8f_queries;
    *certified = true;
    return ok();
}

// ---- a lone query in two halves (vector_index.hpp) ------------------------------------------------------------------------
//
// search_top_k for ONE host query without a filter: begin enqueues on the index's own stream and returns, end waits and writes the
// hits.  What begin picks — the certified int8 pass, the staged filter path, the exact kernels — is what search_top_k always picked
// for a lone caller; a row-sharded handle begins the query on every shard before it ends any.
SearchError VectorIndex::lone_exact_begin(const float* query, uint32_t k) {
    lone_ = LoneState{};
    lone_.query = query;
    lone_.k = k;
    if (k == 0 || nrows_ == 0) {
        lone_.kind = kLoneEmpty;
        return ok();
    }
    FSGPU_HIP(hipSetDevice(device_));
    const size_t qbytes = (size_t)dim_ * 4;
    FSGPU_TRY(ws_queries_.reserve(qbytes));
    const size_t io_need = qbytes + (size_t)k * 8 + 4 + 256;
    if (io_need > kPinnedIoBytes || pinned_io() == nullptr) {
        lone_.kind = kLoneUnpinned;
        return ok();
    }
    unsigned char* io = static_cast<unsigned char*>(io_host_);
    float* q_pin = reinterpret_cast<float*>(io);
    uint32_t* rows_pin = reinterpret_cast<uint32_t*>(io + ((qbytes + 63) & ~(size_t)63));
    float* scores_pin = reinterpret_cast<float*>(rows_pin + k);
    uint32_t* counts_pin = reinterpret_cast<uint32_t*>(scores_pin + k);
    // opted in (fsgpu_index_set_int8_latency): the same hits through the int8 filter + exact re-score — half the bytes of the
    // exact kernel's pass; anything that path does not cover falls through to the exact kernels inside it
    const bool i8_shape = batched_filter != 1 && !i8f_disabled_ && !f32_ && !(row_stride_ && row_stride_ != dim_ * 2) && nrows_ >= 4 * 8192ull;
    const bool via_filter = int8_latency && !exact_only_ && i8_shape && k <= 64;
    // By default (round 5): an index that already HOLDS the int8 copy and its statistics — some batched search built them — answers a
    // lone query with the certified pass over that copy too: the rows and score bits of the exact kernels from half the bytes
    // (10M x 384: p50 0.67 against 1.27 ms; 1M: 0.12 against 0.17).  Nothing is built for it, an uncertified query goes to the exact
    // kernels, and fsgpu_search_topk_exact keeps those kernels reachable as they are.
    const bool by_default = !via_filter && !exact_only_ && i8_shape && k <= 32 && filter_ready() && variant == 0;
    // a lone query of a fused-kernel shape travels in the scan kernel's argument block: no H2D copy in front of the scan
    const uint32_t k_lat = (uint64_t)k < nrows_ ? k : (uint32_t)nrows_;
    const bool in_kernarg = !via_filter && !f32_ && dim_ % 8 == 0 && k_lat <= 256 && variant == 0 &&
                            scan_kernarg_query_supported((int)dim_, k_lat <= 64 ? 64 : 256);
    if ((via_filter && k <= 32 && filter_ready() && variant == 0) || by_default) {
        // A failed certificate costs a whole pass over the int8 copy (a query with more rows inside the margin than the finish
        // holds, or so many in one block's share that its list dropped one), so the single pass backs off: after a failure the
        // next 1, 2, 4 ... 64 lone queries go straight to the staged path (the exact kernels when the pass is the default); a
        // success resets it.
        // (the pass reads the query from the pinned staging block itself: no H2D copy in front of it)
        if (cert_skip_ > 0) {
            --cert_skip_;
        } else {
            bool enqueued = false;
            FSGPU_TRY(certified_i8_enqueue(query, k, &enqueued));
            if (enqueued) {
                lone_.kind = kLoneCertified;
                lone_.staged_behind = via_filter;
                return ok();
            }
        }
    }
    if (via_filter && async_state_[0] != 0 && async_state_[1] != 0) {   // both tickets of the staged path are out: end() answers, blocking
        lone_.kind = kLoneStagedBlocking;
        return ok();
    }
    if (!in_kernarg) {
        std::memcpy(q_pin, query, qbytes);
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, q_pin, qbytes, hipMemcpyHostToDevice, stream_));
    }
    if (via_filter) {
        FSGPU_TRY(search_top_k_batched_device_begin(static_cast<const float*>(ws_queries_.ptr), 1, dim_, k, nullptr, rows_pin, scores_pin,
                                                    counts_pin, stream_, nullptr, &lone_.ticket));
        lone_.kind = kLoneStaged;
    } else {
        host_query_hint_ = in_kernarg ? query : nullptr;
        const SearchError se = search_top_k_device(static_cast<const float*>(ws_queries_.ptr), 1, dim_, k, nullptr, rows_pin, scores_pin,
                                                   counts_pin, stream_);
        host_query_hint_ = nullptr;
        FSGPU_TRY(se);
        lone_.kind = kLoneExact;
    }
    return ok();
}

SearchError VectorIndex::lone_exact_end(uint32_t* out_rows, float* out_scores, uint32_t* out_count) {
    const LoneState st = lone_;
    lone_ = LoneState{};
    const uint32_t k = st.k;
    if (st.kind == kLoneEmpty) {
        *out_count = 0;
        return ok();
    }
    if (st.kind == kLoneNone) return make_error(FSGPU_ERR_INVALID_CONFIG, "no lone query was begun on this index");
    FSGPU_HIP(hipSetDevice(device_));
    const size_t qbytes = (size_t)dim_ * 4;
    if (st.kind == kLoneUnpinned) {   // no pinned staging block: pageable copies either side of the exact kernels
        FSGPU_TRY(ws_rows_.reserve((size_t)k * 4));
        FSGPU_TRY(ws_scores_.reserve((size_t)k * 4));
        FSGPU_TRY(ws_counts_.reserve(4));
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, st.query, qbytes, hipMemcpyHostToDevice, stream_));
        FSGPU_TRY(search_top_k_device(static_cast<const float*>(ws_queries_.ptr), 1, dim_, k, nullptr, static_cast<uint32_t*>(ws_rows_.ptr),
                                      static_cast<float*>(ws_scores_.ptr), static_cast<uint32_t*>(ws_counts_.ptr), stream_));
        FSGPU_HIP(hipMemcpyAsync(out_rows, ws_rows_.ptr, (size_t)k * 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipMemcpyAsync(out_scores, ws_scores_.ptr, (size_t)k * 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipMemcpyAsync(out_count, ws_counts_.ptr, 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipStreamSynchronize(stream_));
        return ok();
    }
    unsigned char* io = static_cast<unsigned char*>(io_host_);
    float* q_pin = reinterpret_cast<float*>(io);
    uint32_t* rows_pin = reinterpret_cast<uint32_t*>(io + ((qbytes + 63) & ~(size_t)63));
    float* scores_pin = reinterpret_cast<float*>(rows_pin + k);
    uint32_t* counts_pin = reinterpret_cast<uint32_t*>(scores_pin + k);
    bool staged_blocking = st.kind == kLoneStagedBlocking;
    if (st.kind == kLoneCertified) {
        bool certified = false;
        FSGPU_TRY(certified_i8_check(out_rows, out_scores, out_count, &certified));
        if (certified) {
            cert_backoff_ = 0;
            return ok();
        }
        cert_backoff_ = cert_backoff_ ? std::min<uint32_t>(cert_backoff_ * 2, 64) : 1;
        cert_skip_ = cert_backoff_;
        if (st.staged_behind) {
            staged_blocking = true;
        } else {   // the pass was the default, not an opt-in: the exact kernels answer
            const uint32_t k_lat = (uint64_t)k < nrows_ ? k : (uint32_t)nrows_;
            const bool in_kernarg = !f32_ && dim_ % 8 == 0 && k_lat <= 256 && variant == 0 && scan_kernarg_query_supported((int)dim_, k_lat <= 64 ? 64 : 256);
            if (!in_kernarg) {
                std::memcpy(q_pin, st.query, qbytes);
                FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, q_pin, qbytes, hipMemcpyHostToDevice, stream_));
            }
            host_query_hint_ = in_kernarg ? st.query : nullptr;
            const SearchError se = search_top_k_device(static_cast<const float*>(ws_queries_.ptr), 1, dim_, k, nullptr, rows_pin, scores_pin,
                                                       counts_pin, stream_);
            host_query_hint_ = nullptr;
            FSGPU_TRY(se);
        }
    }
    if (staged_blocking) {   // the staged filter path, in one piece
        std::memcpy(q_pin, st.query, qbytes);
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, q_pin, qbytes, hipMemcpyHostToDevice, stream_));
        uint32_t fb = 0;
        FSGPU_TRY(search_top_k_batched_device(static_cast<const float*>(ws_queries_.ptr), 1, dim_, k, nullptr, rows_pin, scores_pin, counts_pin,
                                              stream_, &fb));
    } else if (st.kind == kLoneStaged) {
        uint32_t fb = 0;
        FSGPU_TRY(search_top_k_batched_device_end(st.ticket, &fb));
    }
    FSGPU_HIP(hipStreamSynchronize(stream_));
    std::memcpy(out_rows, rows_pin, (size_t)k * 4);
    std::memcpy(out_scores, scores_pin, (size_t)k * 4);
    *out_count = *counts_pin;
    return ok();
}

// search_top_k_int8_two_pass_impl (crates/frankensearch-index/src/search.rs:589-661)
SearchError VectorIndex::search_top_k_int8_two_pass(const float* query, uint32_t query_len, uint32_t k,
                                                    uint32_t multiplier, uint32_t* out_rows, float* out_scores,
                                                    uint32_t* out_count) {
    return quantized_two_pass(query, query_len, k, multiplier, 8, out_rows, out_scores, out_count);
}

// search_top_k_4bit_two_pass (crates/frankensearch-index/src/search.rs:876-946)
SearchError VectorIndex::search_top_k_4bit_two_pass(const float* query, uint32_t query_len, uint32_t k,
                                                    uint32_t multiplier, uint32_t* out_rows, float* out_scores,
                                                    uint32_t* out_count) {
    return quantized_two_pass(query, query_len, k, multiplier, 4, out_rows, out_scores, out_count);
}

// The two-pass searches' lane for ONE caller (see quantized_two_pass).  query / qi: the f32 query and its quantised form (host);
// rows / scores: [k] on the host.  *answered = false: nothing was written.
SearchError VectorIndex::two_pass_lone_certified(const float* query, const unsigned char* qi, uint32_t qbytes, uint32_t k, uint32_t k_eff,
                                                 uint32_t cc, int bits, const void* qslab, uint32_t* rows, float* scores, uint32_t* count,
                                                 bool* answered) {
    *answered = false;
    bool enqueued = false;
    FSGPU_TRY(two_pass_lone_enqueue(query, qi, qbytes, k, k_eff, cc, bits, qslab, false, &enqueued));
    if (!enqueued) return ok();
    return two_pass_lone_check(rows, scores, count, nullptr, nullptr, answered);
}

// Enqueue only: pass 1 keeping 32 entries per block, the cut, the cc best pass-1 entries (best first, into pinned memory), their exact
// scores, the k best of those.  want_pairs: the candidates' exact entries go to pinned memory as well, aligned with the pass-1 entries
// (what a row-sharded handle's root merges).
SearchError VectorIndex::two_pass_lone_enqueue(const float* query, const unsigned char* qi, uint32_t qbytes, uint32_t k, uint32_t k_eff,
                                               uint32_t cc, int bits, const void* qslab, bool want_pairs, bool* enqueued) {
    *enqueued = false;
    constexpr uint32_t LK = 32;
    const bool fused = bits == 8 ? scan_i8_fused_supported((int)dim_, 64) : scan_4bit_fused_supported((int)dim_, 64);
    if (!fused || pinned_io() == nullptr) return ok();
    const size_t fbytes = (size_t)dim_ * 4;
    const size_t o_qi = (fbytes + 255) & ~(size_t)255, o_out = (o_qi + qbytes + 255) & ~(size_t)255,
                 o_flags = (o_out + (size_t)k * 8 + 4 + 255) & ~(size_t)255;
    const size_t o_approx = (o_flags + 64 + 255) & ~(size_t)255, o_exact = (o_approx + (size_t)cc * 8 + 255) & ~(size_t)255;
    if (o_exact + (size_t)cc * 8 > kPinnedIoBytes) return ok();
    unsigned char* io = static_cast<unsigned char*>(io_host_);
    std::memcpy(io, query, fbytes);
    std::memcpy(io + o_qi, qi, qbytes);
    const float* q_pin = reinterpret_cast<const float*>(io);
    float* delta_pin = reinterpret_cast<float*>(io + o_flags);   // the pass-1 scores are the reference's own: no margin
    float* cut_pin = delta_pin + 2;
    *delta_pin = 0.f;
    int grid = num_cus_;   // 256 lists x 32 entries: what the sorted selection holds in one piece
    const int max_useful = (int)(((nrows_ + 15) / 16 + 3) / 4);
    grid = std::max(1, std::min(grid, max_useful));
    if ((size_t)grid * LK > 8192) return ok();
    // (every wave of the scan reads the whole quantised query: from device memory, not over the bus; the finish's one block reads
    // the f32 query where it lies)
    FSGPU_TRY(ws_i8_query_.reserve(qbytes));
    FSGPU_TRY(ws_partial_.reserve((size_t)grid * LK * 8));
    FSGPU_TRY(ws_cand_packed_.reserve((size_t)cc * 8));
    FSGPU_TRY(ws_cand_rows_.reserve((size_t)cc * 4));
    FSGPU_TRY(ws_cand_scores_.reserve((size_t)cc * 4));
    FSGPU_HIP(hipMemcpyAsync(ws_i8_query_.ptr, io + o_qi, qbytes, hipMemcpyHostToDevice, stream_));
    ScanArgs a = base_args(q_pin, nullptr);
    a.partial = static_cast<u64*>(ws_partial_.ptr);
    a.k = LK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (profiling) {
        FSGPU_HIP(hipEventCreate(&e0));
        FSGPU_HIP(hipEventCreate(&e1));
        FSGPU_HIP(hipEventRecord(e0, stream_));
    }
    if (bits == 8) FSGPU_HIP(launch_scan_i8(a, qslab, ws_i8_query_.ptr, 64, grid, stream_, nullptr));
    else FSGPU_HIP(launch_scan_4bit(a, qslab, ws_i8_query_.ptr, 64, grid, stream_, nullptr));
    if (profiling) {
        FSGPU_HIP(hipEventRecord(e1, stream_));
        events_.emplace_back(e0, e1);
        profiled_rows_ += nrows_;
    }
    FSGPU_HIP(launch_list_cut(a.partial, (uint32_t)grid, LK, cut_pin, stream_));
    // the cc best pass-1 entries of the 8,192 kept (ONE pass of the merge; the selection's sorted finish took 0.10 ms here), their exact
    // scores, the k best of those — the general sequence's kernels over lists a third as long
    u64* approx_pin = reinterpret_cast<u64*>(io + o_approx);
    u64* exact_pin = reinterpret_cast<u64*>(io + o_exact);
    uint32_t* cand_rows = static_cast<uint32_t*>(ws_cand_rows_.ptr);
    float* cand_scores = static_cast<float*>(ws_cand_scores_.ptr);
    u64* cand_packed = static_cast<u64*>(ws_cand_packed_.ptr);
    MergeArgs m;
    m.lists = a.partial;
    m.q_stride = (uint64_t)grid * LK;
    m.l_stride = LK;
    m.nlists = (uint32_t)grid;
    m.list_len = LK;
    m.k = cc;
    m.out_stride = cc;
    m.out_rows = cand_rows;
    m.out_scores = nullptr;
    m.out_counts = nullptr;
    m.out_packed = approx_pin;   // best first: the certificate reads the last one
    FSGPU_HIP(launch_merge_topk(m, 1, stream_));
    FSGPU_HIP(hipMemsetAsync(cand_scores, 0, (size_t)cc * 4, stream_));
    FSGPU_HIP(launch_gather_dot(a, cand_rows, cc, cand_scores, stream_));
    FSGPU_HIP(launch_pack_hits(cand_rows, cand_scores, cc, cand_packed, stream_));
    if (want_pairs) FSGPU_HIP(launch_pack_hits(cand_rows, cand_scores, cc, exact_pin, stream_));
    MergeArgs m2;
    m2.lists = cand_packed;
    m2.q_stride = cc;
    m2.l_stride = cc;
    m2.nlists = 1;
    m2.list_len = cc;
    m2.k = k_eff;
    m2.out_stride = k;
    m2.out_rows = reinterpret_cast<uint32_t*>(io + o_out);
    m2.out_scores = reinterpret_cast<float*>(io + o_out + (size_t)k * 4);
    m2.out_counts = reinterpret_cast<uint32_t*>(io + o_out + (size_t)k * 8);
    m2.out_packed = nullptr;
    m2.lists_sorted = 0;  // candidates arrive in pass-1 order
    FSGPU_HIP(launch_merge_topk(m2, 1, stream_));
    tp_lane_ = TwoPassLane{k, cc, o_out, o_flags, o_approx, o_exact};
    *enqueued = true;
    return ok();
}

// The other half: one synchronisation, then the certificate — complete when no list was full (nothing dropped) or the cc-th best entry
// outranks everything dropped, STRICTLY: a dropped row with the same integer score may have the smaller row id.
// approx_out / exact_out (may be null): the cc candidate pairs.
SearchError VectorIndex::two_pass_lone_check(uint32_t* rows, float* scores, uint32_t* count, u64* approx_out, u64* exact_out, bool* answered) {
    *answered = false;
    const TwoPassLane L = tp_lane_;
    unsigned char* io = static_cast<unsigned char*>(io_host_);
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    const float cut = *(reinterpret_cast<const float*>(io + L.o_flags) + 2);
    const u64* approx_pin = reinterpret_cast<const u64*>(io + L.o_approx);
    bool complete = cut == -INFINITY;
    if (!complete && approx_pin[L.cc - 1] != ~0ull) {
        float tau;
        const uint32_t tb = (uint32_t)(approx_pin[L.cc - 1] >> 32);
        std::memcpy(&tau, &tb, 4);
        complete = cut < tau;
    }
    if (!complete) return ok();
    if (rows) std::memcpy(rows, io + L.o_out, (size_t)L.k * 4);
    if (scores) std::memcpy(scores, io + L.o_out + (size_t)L.k * 4, (size_t)L.k * 4);
    if (count) *count = *reinterpret_cast<const uint32_t*>(io + L.o_out + (size_t)L.k * 8);
    if (approx_out) std::memcpy(approx_out, approx_pin, (size_t)L.cc * 8);
    if (exact_out) std::memcpy(exact_out, io + L.o_exact, (size_t)L.cc * 8);
    *answered = true;
    return ok();
}

// quantize_i8_query (search.rs:1616-1626) / pack_4bit_query (:1640-1653): the query's own max-abs scale, round half away from zero,
// clamp; NaN -> 0
static void quantize_query_host(const float* query, uint32_t dim, int bits, std::vector<unsigned char>& qi) {
    const uint32_t qbytes = bits == 8 ? dim : (dim + 1) / 2;
    qi.assign(qbytes, 0);
    float max_abs = 0.f;
    for (uint32_t i = 0; i < dim; ++i) {
        const float v = std::fabs(query[i]);
        if (v > max_abs) max_abs = v;
    }
    const float lim = bits == 8 ? 127.0f : 7.0f;
    const bool usable = bits == 8 ? max_abs > 0.f : max_abs > 1e-9f;
    const float scale = usable ? lim / max_abs : 0.f;
    if (bits == 4 || usable) {
        for (uint32_t i = 0; i < dim; ++i) {
            float v = std::round(query[i] * scale);
            if (v != v) v = 0.f;
            v = std::min(std::max(v, -lim), lim);
            const int qv = (int)v;
            if (bits == 8) qi[i] = (unsigned char)(signed char)qv;
            else qi[i / 2] |= (unsigned char)((qv & 0xF) << ((i & 1) ? 4 : 0));
        }
    }
}

// The quantised copy a two-pass search scans, built lazily, once (VectorIndex::int8_slab() / nibbles_slab(), search.rs:988-1000).
SearchError VectorIndex::ensure_two_pass_slab(int bits, const void** qslab) {
    const size_t n = (size_t)nrows_;
    const uint32_t qbytes = bits == 8 ? dim_ : (dim_ + 1) / 2;
    if (bits == 8 && !i8_ready_) {
        FSGPU_TRY(i8_slab_.reserve(n * dim_));
        FSGPU_TRY(i8_max_.reserve(4));
        FSGPU_HIP(launch_quantize_slab_i8(slab_dev_, n * dim_, static_cast<unsigned int*>(i8_max_.ptr), i8_slab_.ptr,
                                          stream_, quant_max_ready_));
        i8_ready_ = true;
    }
    if (bits == 4 && !n4_ready_) {
        FSGPU_TRY(n4_slab_.reserve(n * qbytes));
        FSGPU_TRY(i8_max_.reserve(4));
        FSGPU_HIP(launch_pack_slab_4bit(slab_dev_, nrows_, dim_, static_cast<unsigned int*>(i8_max_.ptr), n4_slab_.ptr,
                                        stream_, quant_max_ready_));
        n4_ready_ = true;
    }
    *qslab = bits == 8 ? i8_slab_.ptr : n4_slab_.ptr;
    return ok();
}

// One query of a row-sharded two-pass search, this shard's half, in two halves: begin enqueues (the lone caller's lane when the
// shape allows, else the batched sequence with one query), end yields the shard's cc_out = max(k * multiplier, k) candidate pairs
// (pass-1 entry, exact entry; kEmpty beyond the candidates) — what two_pass_candidates_device yields for one query.
SearchError VectorIndex::lone_two_pass_begin(const float* query, uint32_t k, uint32_t multiplier, int bits) {
    lone_ = LoneState{};
    lone_.query = query;
    lone_.k = k;
    lone_.mult = multiplier ? multiplier : 1;
    lone_.bits = bits == 4 ? 4 : 8;
    const uint64_t cc_out64 = std::max<uint64_t>((uint64_t)k * lone_.mult, k);
    if (cc_out64 > 256 || k == 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "sharded two-pass: 1 <= k, k * multiplier <= 256");
    lone_.cc_out = (uint32_t)cc_out64;
    if (f32_) return make_error(FSGPU_ERR_INVALID_CONFIG, "two-pass searches need an F16 slab");
    if (nrows_ == 0) {
        lone_.kind = kLoneEmpty;
        return ok();
    }
    FSGPU_HIP(hipSetDevice(device_));
    const void* qslab = nullptr;
    FSGPU_TRY(ensure_two_pass_slab(lone_.bits, &qslab));
    uint64_t cc64 = std::min<uint64_t>((uint64_t)k * lone_.mult, nrows_);
    cc64 = std::max<uint64_t>(cc64, std::min<uint64_t>(k, nrows_));
    const uint32_t cc = (uint32_t)cc64, k_eff = (uint32_t)std::min<uint64_t>(k, nrows_);
    lone_.cc = cc;
    if (cc <= kSelectMaxK && k_eff <= 64 && k <= 64 && (dim_ & 7) == 0 && nrows_ >= 4096 && !(row_stride_ && row_stride_ != dim_ * 2) && variant == 0) {
        if (tp_skip_ > 0) {
            --tp_skip_;
        } else {
            std::vector<unsigned char> qi;
            quantize_query_host(query, dim_, lone_.bits, qi);
            bool enqueued = false;
            FSGPU_TRY(two_pass_lone_enqueue(query, qi.data(), (uint32_t)qi.size(), k, k_eff, cc, lone_.bits, qslab, true, &enqueued));
            if (enqueued) {
                lone_.kind = kLoneTwoPassLane;
                return ok();
            }
        }
    }
    FSGPU_TRY(ws_pairs_.reserve((size_t)lone_.cc_out * 16));
    if (async_state_[0] != 0 && async_state_[1] != 0) {
        lone_.kind = kLoneTwoPassBlocking;
        return ok();
    }
    FSGPU_TRY(ws_queries_.reserve((size_t)dim_ * 4));
    FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, query, (size_t)dim_ * 4, hipMemcpyHostToDevice, stream_));
    u64* pairs = static_cast<u64*>(ws_pairs_.ptr);
    FSGPU_TRY(two_pass_candidates_device_begin(static_cast<const float*>(ws_queries_.ptr), 1, dim_, k, lone_.mult, lone_.bits,
                                               reinterpret_cast<uint64_t*>(pairs), reinterpret_cast<uint64_t*>(pairs + lone_.cc_out), stream_,
                                               &lone_.ticket));
    lone_.kind = kLoneTwoPassBatched;
    return ok();
}

SearchError VectorIndex::lone_two_pass_end(uint64_t* out_approx, uint64_t* out_exact) {
    const LoneState st = lone_;
    lone_ = LoneState{};
    if (st.kind == kLoneNone) return make_error(FSGPU_ERR_INVALID_CONFIG, "no lone query was begun on this index");
    for (uint32_t i = 0; i < st.cc_out; ++i) out_approx[i] = out_exact[i] = ~0ull;
    if (st.kind == kLoneEmpty) return ok();
    FSGPU_HIP(hipSetDevice(device_));
    bool blocking = st.kind == kLoneTwoPassBlocking;
    if (st.kind == kLoneTwoPassLane) {
        bool answered = false;
        FSGPU_TRY(two_pass_lone_check(nullptr, nullptr, nullptr, reinterpret_cast<u64*>(out_approx), reinterpret_cast<u64*>(out_exact), &answered));
        if (answered) {
            tp_backoff_ = 0;
            return ok();
        }
        tp_backoff_ = tp_backoff_ ? std::min<uint32_t>(tp_backoff_ * 2, 64) : 1;
        tp_skip_ = tp_backoff_;
        FSGPU_TRY(ws_pairs_.reserve((size_t)st.cc_out * 16));
        blocking = true;
    }
    u64* pairs = static_cast<u64*>(ws_pairs_.ptr);
    if (blocking) {   // the general sequence, in one piece (quantized_two_pass hands the pairs on when asked to)
        FSGPU_HIP(hipMemsetAsync(pairs, 0xff, (size_t)st.cc_out * 16, stream_));
        std::vector<uint32_t> rows(st.k);
        std::vector<float> scores(st.k);
        uint32_t cnt = 0;
        FSGPU_TRY(quantized_two_pass(st.query, dim_, st.k, st.mult, st.bits, rows.data(), scores.data(), &cnt, pairs, pairs + st.cc_out));
    } else {
        uint32_t fb = 0;
        FSGPU_TRY(two_pass_candidates_device_end(st.ticket, &fb));
    }
    FSGPU_HIP(hipMemcpyAsync(out_approx, pairs, (size_t)st.cc_out * 8, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_exact, pairs + st.cc_out, (size_t)st.cc_out * 8, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    return ok();
}

// Shared body of the int8 (bits = 8) and 4-bit (bits = 4) two-pass searches: quantised pass 1 over the lazily built
// slab, exact f16 rescore of the candidates, best-first selection of k.
SearchError VectorIndex::quantized_two_pass(const float* query, uint32_t query_len, uint32_t k, uint32_t multiplier,
                                            int bits, uint32_t* out_rows, float* out_scores, uint32_t* out_count,
                                            u64* approx_out_dev, u64* exact_out_dev) {
    *out_count = 0;
    // anything the fast path does not cover goes through the exact search (search.rs:579-585)
    if (!approx_out_dev && (k == 0 || nrows_ == 0 || !wal_.empty() || f32_)) {  // ... || quantization != F16
        if (has_doc_ids()) return search_hits(query, query_len, k, out_rows, out_scores, out_count);
        FSGPU_TRY(ensure_query_dimension(query_len));
        if (k == 0 || nrows_ == 0) return ok();
        return search_top_k(query, 1, query_len, k, nullptr, out_rows, out_scores, out_count);
    }
    FSGPU_TRY(ensure_query_dimension(query_len));
    FSGPU_HIP(hipSetDevice(device_));
    const size_t n = (size_t)nrows_;
    const uint32_t qbytes = bits == 8 ? dim_ : (dim_ + 1) / 2;  // quantised bytes per vector
    const void* qslab = nullptr;
    FSGPU_TRY(ensure_two_pass_slab(bits, &qslab));
    const uint64_t mult = multiplier ? multiplier : 1;
    uint64_t cc64 = std::min<uint64_t>((uint64_t)k * mult, nrows_);
    cc64 = std::max<uint64_t>(cc64, std::min<uint64_t>(k, nrows_));
    const uint32_t cc = (uint32_t)cc64;
    const uint32_t k_eff = (uint32_t)std::min<uint64_t>(k, nrows_);
    std::vector<unsigned char> qi;
    quantize_query_host(query, dim_, bits, qi);
    std::vector<uint32_t> rows(k);
    std::vector<float> scores(k);
    uint32_t count = 0;
    bool answered = false;
    // The lone caller's lane: pass 1 keeping 32 entries per block, the selection's finish (candidates = the cc best pass-1 entries,
    // exact re-score, k best) — three launches, nothing copied — certified on the host: the cc-th best pass-1 entry lies strictly
    // above everything a block can have dropped.  Otherwise (and for a row-sharded index's shards, which hand the candidate pairs on)
    // the general sequence below answers; a failed certificate backs off like the exact search's (certified_i8_lone_query).
    // (worth it from 65 candidates on, where the general sequence's block lists no longer fit the one-pass merge — the two-tier
    // host's fast tier fetches 30 x 3: 10M x 256 p50 0.59 -> 0.50 ms; below that both sequences measured the same)
    if (!approx_out_dev && !exact_out_dev && cc > 64 && cc <= kSelectMaxK && k_eff <= 64 && k <= 64 && (dim_ & 7) == 0 && nrows_ >= 4096 &&
        !(row_stride_ && row_stride_ != dim_ * 2) && variant == 0) {
        if (tp_skip_ > 0) {
            --tp_skip_;
        } else {
            FSGPU_TRY(two_pass_lone_certified(query, qi.data(), qbytes, k, k_eff, cc, bits, qslab, rows.data(), scores.data(), &count, &answered));
            if (answered) {
                tp_backoff_ = 0;
            } else {
                tp_backoff_ = tp_backoff_ ? std::min<uint32_t>(tp_backoff_ * 2, 64) : 1;
                tp_skip_ = tp_backoff_;
            }
        }
    }
    if (!answered) {
    FSGPU_TRY(ws_i8_query_.reserve(qbytes));
    FSGPU_TRY(ws_queries_.reserve((size_t)dim_ * 4));
    FSGPU_TRY(ws_cand_packed_.reserve((size_t)cc * 8));
    FSGPU_TRY(ws_cand_rows_.reserve((size_t)cc * 4));
    FSGPU_TRY(ws_cand_scores_.reserve((size_t)cc * 4));
    FSGPU_TRY(ws_rows_.reserve((size_t)k * 4));
    FSGPU_TRY(ws_scores_.reserve((size_t)k * 4));
    FSGPU_TRY(ws_counts_.reserve(4));
    // both query forms go through the pinned staging block when it exists (DMA instead of pageable staging)
    const size_t qin_bytes = (((size_t)dim_ * 4 + qbytes) + 255) & ~(size_t)255;
    if (qin_bytes <= kPinnedIoBytes / 2 && pinned_io() != nullptr) {
        unsigned char* io = static_cast<unsigned char*>(io_host_);
        std::memcpy(io, query, (size_t)dim_ * 4);
        std::memcpy(io + (size_t)dim_ * 4, qi.data(), qbytes);
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, io, (size_t)dim_ * 4, hipMemcpyHostToDevice, stream_));
        FSGPU_HIP(hipMemcpyAsync(ws_i8_query_.ptr, io + (size_t)dim_ * 4, qbytes, hipMemcpyHostToDevice, stream_));
    } else {
        FSGPU_HIP(hipMemcpyAsync(ws_i8_query_.ptr, qi.data(), qbytes, hipMemcpyHostToDevice, stream_));
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, query, (size_t)dim_ * 4, hipMemcpyHostToDevice, stream_));
    }
    ScanArgs a = base_args(static_cast<const float*>(ws_queries_.ptr), nullptr);
    u64* cand_packed = static_cast<u64*>(ws_cand_packed_.ptr);
    uint32_t* cand_rows = static_cast<uint32_t*>(ws_cand_rows_.ptr);
    float* cand_scores = static_cast<float*>(ws_cand_scores_.ptr);
    // ---- pass 1: top-cc rows by the int8 dot ----
    const int kcap = cc <= 64 ? 64 : 256;
    auto launch_pass1 = [&](int grid, int* occ) {
        return bits == 8 ? launch_scan_i8(a, qslab, ws_i8_query_.ptr, kcap, grid, stream_, occ)
                         : launch_scan_4bit(a, qslab, ws_i8_query_.ptr, kcap, grid, stream_, occ);
    };
    const bool fused = bits == 8 ? scan_i8_fused_supported((int)dim_, kcap) : scan_4bit_fused_supported((int)dim_, kcap);
    if (cc <= 256 && fused) {
        int per_cu = 1;
        FSGPU_HIP(launch_pass1(1, &per_cu));
        // one block per CU for int8: the quantised rows are short, so four double-buffered waves already keep the HBM pipe full,
        // and every extra block is another candidate list for the merge and another top-k to maintain (10M x 256, 90
        // candidates: p50 0.65 -> 0.59 ms; 10M x 384, 30 candidates: 0.73 -> 0.69 ms).  FSGPU_I8_PER_CU overrides.
        // 4-bit rows are half as long again: two blocks per CU (10M x 384: 0.44 -> 0.41 ms against one, 0.44 against four).
        per_cu = std::min(per_cu, knobs().i8_per_cu > 0 ? knobs().i8_per_cu : (bits == 8 ? 1 : 2));
        int grid = num_cus_ * per_cu;
        const int max_useful = (int)(((nrows_ + 15) / 16 + 3) / 4);
        if (grid > max_useful) grid = max_useful;
        if (grid < 1) grid = 1;
        FSGPU_TRY(ws_partial_.reserve((size_t)grid * cc * 8));
        a.partial = static_cast<u64*>(ws_partial_.ptr);
        a.k = cc;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profiling) {
            FSGPU_HIP(hipEventCreate(&e0));
            FSGPU_HIP(hipEventCreate(&e1));
            FSGPU_HIP(hipEventRecord(e0, stream_));
        }
        FSGPU_HIP(launch_pass1(grid, nullptr));
        if (profiling) {
            FSGPU_HIP(hipEventRecord(e1, stream_));
            events_.emplace_back(e0, e1);
            profiled_rows_ += nrows_;
        }
        MergeArgs m;
        m.lists = a.partial;
        m.q_stride = (uint64_t)grid * cc;
        m.l_stride = cc;
        m.nlists = (uint32_t)grid;
        m.list_len = cc;
        m.k = cc;
        m.out_stride = cc;
        m.out_rows = cand_rows;
        m.out_scores = nullptr;
        m.out_counts = nullptr;
        m.out_packed = approx_out_dev;   // (a sharded index's root wants the pass-1 entries themselves)
        FSGPU_HIP(launch_merge_topk(m, 1, stream_));
    } else {
        FSGPU_TRY(ws_keys_a_.reserve(n * 8));
        FSGPU_TRY(ws_keys_b_.reserve(n * 8));
        size_t tmp_bytes = 0;
        FSGPU_HIP(sort_keys_desc_temp_bytes(n, &tmp_bytes));
        FSGPU_TRY(ws_sort_tmp_.reserve(tmp_bytes));
        u64* keys_a = static_cast<u64*>(ws_keys_a_.ptr);
        u64* keys_b = static_cast<u64*>(ws_keys_b_.ptr);
        if (bits == 8) FSGPU_HIP(launch_score_rows_i8(a, qslab, ws_i8_query_.ptr, keys_a, stream_));
        else FSGPU_HIP(launch_score_rows_4bit(a, qslab, ws_i8_query_.ptr, keys_a, stream_));
        FSGPU_HIP(launch_packed_to_sortkey(keys_a, n, stream_));
        FSGPU_HIP(sort_keys_desc(ws_sort_tmp_.ptr, ws_sort_tmp_.bytes, keys_a, keys_b, n, stream_, sortkey_varying_bits(a.live || a.allow)));
        FSGPU_HIP(launch_sorted_keys_to_rows(keys_b, cc, cand_rows, static_cast<uint32_t*>(ws_counts_.ptr), stream_, approx_out_dev));
    }
    // ---- pass 2: exact f16 rescore of the candidates, then the usual best-first selection of k ----
    FSGPU_HIP(hipMemsetAsync(cand_scores, 0, (size_t)cc * 4, stream_));
    FSGPU_HIP(launch_gather_dot(a, cand_rows, cc, cand_scores, stream_));
    FSGPU_HIP(launch_pack_hits(cand_rows, cand_scores, cc, cand_packed, stream_));
    if (exact_out_dev) FSGPU_HIP(hipMemcpyAsync(exact_out_dev, cand_packed, (size_t)cc * 8, hipMemcpyDeviceToDevice, stream_));
    MergeArgs m2;
    m2.lists = cand_packed;
    m2.q_stride = cc;
    m2.l_stride = cc;
    m2.nlists = 1;
    m2.list_len = cc;
    m2.k = k_eff;
    m2.out_stride = k;
    m2.out_rows = static_cast<uint32_t*>(ws_rows_.ptr);
    m2.out_scores = static_cast<float*>(ws_scores_.ptr);
    m2.out_counts = static_cast<uint32_t*>(ws_counts_.ptr);
    const bool pin_out = (size_t)k * 8 + 4 + qin_bytes <= kPinnedIoBytes && pinned_io() != nullptr;
    if (pin_out) {
        unsigned char* io = static_cast<unsigned char*>(io_host_) + qin_bytes;
        m2.out_rows = reinterpret_cast<uint32_t*>(io);
        m2.out_scores = reinterpret_cast<float*>(io + (size_t)k * 4);
        m2.out_counts = reinterpret_cast<uint32_t*>(io + (size_t)k * 8);
    }
    m2.out_packed = nullptr;
    m2.lists_sorted = 0;  // candidates arrive in pass-1 (int8) order
    FSGPU_HIP(launch_merge_topk(m2, 1, stream_));
    if (pin_out) {  // the last merge wrote into pinned host memory
        FSGPU_HIP(hipStreamSynchronize(stream_));
        std::memcpy(rows.data(), m2.out_rows, (size_t)k * 4);
        std::memcpy(scores.data(), m2.out_scores, (size_t)k * 4);
        count = *m2.out_counts;
    } else {
        FSGPU_HIP(hipMemcpyAsync(rows.data(), ws_rows_.ptr, (size_t)k * 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipMemcpyAsync(scores.data(), ws_scores_.ptr, (size_t)k * 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipMemcpyAsync(&count, ws_counts_.ptr, 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipStreamSynchronize(stream_));
    }
    }   // (!answered)
    // resolve_hits (search.rs:1503-1558): first (best) hit per doc id when the index knows doc ids
    uint32_t outn = 0;
    for (uint32_t i = 0; i < count; ++i) {
        bool dup = false;
        if (has_doc_ids()) {
            const size_t r = (size_t)(rows[i] - row_base_);
            const char* di = doc_blob_.data() + doc_offsets_[r];
            const size_t dl = (size_t)(doc_offsets_[r + 1] - doc_offsets_[r]);
            for (uint32_t j = 0; j < outn && !dup; ++j) {
                const size_t rj = (size_t)(out_rows[j] - row_base_);
                const size_t lj = (size_t)(doc_offsets_[rj + 1] - doc_offsets_[rj]);
                dup = lj == dl && std::memcmp(doc_blob_.data() + doc_offsets_[rj], di, dl) == 0;
            }
        }
        if (dup) continue;
        out_rows[outn] = rows[i];
        out_scores[outn] = scores[i];
        ++outn;
    }
    *out_count = outn;
    return ok();
}

}  // namespace fsgpu
