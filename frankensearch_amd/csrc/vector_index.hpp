// vector_index.hpp — host-side mirror of the reference's VectorIndex / Model2VecEmbedder for the
// device-resident path (names and error behaviour follow crates/frankensearch-index/src/lib.rs:819,
// src/search.rs:192-494 and crates/frankensearch-embed/src/model2vec_embedder.rs:55-58).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "kernels.hpp"

namespace fsgpu {

// SearchError (crates/frankensearch-core/src/error.rs:57-176) carried as code + detail.
struct SearchError {
    int32_t code = 0;
    std::string detail;
    bool ok() const { return code == 0; }
};

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    SearchError reserve(size_t want);
    void release();
};

// VectorIndexWriter for FSVI v1 (lib.rs:3637-3672, 3752-3943): validate, stable-sort by (FNV-1a(doc_id), doc_id), write.
SearchError write_fsvi_v1(const char* path, const char* embedder_id, const char* embedder_revision, uint32_t dim, uint64_t n,
                          const char* const* doc_ids, const uint32_t* doc_id_lens, const float* vectors,
                          uint8_t compaction_gen, int device, uint8_t quantization = 1);

class VectorIndex {
  public:
    VectorIndex() = default;
    ~VectorIndex();
    VectorIndex(const VectorIndex&) = delete;
    VectorIndex& operator=(const VectorIndex&) = delete;

    // VectorIndex::open for a raw slab (host copy) / an adopted device slab / an FSVI v1 file.
    // f32_rows: the slab holds raw little-endian f32 rows (Quantization::F32) instead of f16
    SearchError init_host(int device, uint32_t dim, uint64_t nrows, const void* slab, const uint64_t* live,
                          uint64_t row_base, bool f32_rows = false);
    bool f32_rows() const { return f32_; }
    SearchError init_device(int device, uint32_t dim, uint64_t nrows, const void* slab_dev, const uint64_t* live_dev,
                            uint64_t row_base);
    SearchError open_fsvi(const char* path, int device);
    // catalog of a row-sharded index: the file's tables stay here (doc ids, tombstones, WAL, hit resolution), the slab is handed
    // back in `image` for the shards; topk_override replaces this object's own scan inside search_hits
    struct FsviImage {
        std::vector<uint8_t> bytes;
        size_t slab_offset = 0;
        uint32_t dim = 0;
        uint64_t nrows = 0;
        bool f32_rows = false;
    };
    SearchError open_fsvi_catalog(const char* path, FsviImage* image);
    std::function<SearchError(const float* query, uint32_t k, uint32_t* rows, float* scores, uint32_t* count)> topk_override;
    const std::vector<uint64_t>& live_host() const { return live_host_; }

    uint64_t record_count() const { return nrows_; }
    uint32_t dimension() const { return dim_; }

    // search_top_k over nq queries (host pointers; synchronous).
    // allow_resident_dev: the device copy of `allow` when the caller keeps one (fsgpu_allow_bitmap: a filter reused across
    // searches is uploaded once, not per call); null = `allow` is uploaded for this call
    SearchError search_top_k(const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                             const uint64_t* allow, uint32_t* out_rows, float* out_scores, uint32_t* out_counts,
                             const uint64_t* allow_resident_dev = nullptr);
    // same with device pointers, enqueued on `stream`.
    SearchError search_top_k_device(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                    const uint64_t* allow_dev, uint32_t* out_rows_dev, float* out_scores_dev,
                                    uint32_t* out_counts_dev, hipStream_t stream);
    // Batched search on the matrix cores (mfma_scan.hip): groups of 64 queries per HBM pass, approximate f16-query
    // scores filtered by a proven margin and re-scored in the reference order, so the outputs equal
    // search_top_k_device bit for bit.  Queries it cannot serve (k > 64, unsupported dim, margin overflow) run on the
    // exact kernels.  Synchronises the stream (the fallback decision is taken on the host).
    SearchError search_top_k_batched_device(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                            const uint64_t* allow_dev, uint32_t* out_rows_dev, float* out_scores_dev,
                                            uint32_t* out_counts_dev, hipStream_t stream, uint32_t* fallbacks,
                                            uint64_t* out_packed_dev = nullptr);
    // ... in two halves: begin enqueues the whole search and returns a ticket (0 / 1; two may be outstanding), end waits for that
    // search alone, reads the verdicts and runs the fallbacks.  Queries and outputs stay the caller's until end.
    SearchError search_top_k_batched_device_begin(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                                  const uint64_t* allow_dev, uint32_t* out_rows_dev, float* out_scores_dev,
                                                  uint32_t* out_counts_dev, hipStream_t stream, uint64_t* out_packed_dev, int32_t* ticket);
    // late_answers (may be null): queries whose hits were written by work enqueued in END — exact fallbacks and queries the int8 filter
    // handed to the f16 filter; a caller that ordered work behind begin's last kernel must order it again behind these
    SearchError search_top_k_batched_device_end(int32_t ticket, uint32_t* fallbacks, uint32_t* late_answers = nullptr);
    SearchError search_top_k_batched(const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                     const uint64_t* allow, uint32_t* out_rows, float* out_scores, uint32_t* out_counts,
                                     uint32_t* fallbacks, const uint64_t* allow_resident_dev = nullptr,
                                     bool queries_on_device = false);   // queries: a device pointer (an encoder's device output)
    // Batched search_top_k_int8_two_pass (search.rs:514-661): int8 pass 1 on the matrix cores (exact integer scores, so the
    // k*multiplier candidates are exactly the reference's), exact f16 rescore, top-k.  No doc-id dedup (raw row ids).
    SearchError search_top_k_int8_batched_device(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                                 uint32_t multiplier, uint32_t* out_rows_dev, float* out_scores_dev,
                                                 uint32_t* out_counts_dev, hipStream_t stream, uint32_t* fallbacks, int bits = 8);
    // bits = 4: batched search_top_k_4bit_two_pass (search.rs:876-946)
    SearchError search_top_k_int8_batched(const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                          uint32_t multiplier, uint32_t* out_rows, float* out_scores, uint32_t* out_counts,
                                          uint32_t* fallbacks, int bits = 8, bool queries_on_device = false);
    // Shard-local HALF of a two-pass search (bits 8: search_top_k_int8_two_pass, 4: search_top_k_4bit_two_pass) for a row-sharded
    // index: this shard's cc = max(k * multiplier, k) pass-1 candidates per query as two aligned packed lists [nq, cc] — position
    // i is one row: its pass-1 entry (integer score as f32 bits | global row) and its exact entry; kEmpty beyond the candidates.
    // The root repeats the selection over all shards' candidates (launch_two_pass_merge).  Synchronises the stream.
    SearchError two_pass_candidates_device(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                           uint32_t multiplier, int bits, uint64_t* approx_out_dev, uint64_t* exact_out_dev,
                                           hipStream_t stream, uint32_t* fallbacks);
    // ... in two halves (the tickets of search_top_k_batched_device_begin / _end): nothing is waited for in begin
    SearchError two_pass_candidates_device_begin(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                                 uint32_t multiplier, int bits, uint64_t* approx_out_dev, uint64_t* exact_out_dev,
                                                 hipStream_t stream, int32_t* ticket);
    SearchError two_pass_candidates_device_end(int32_t ticket, uint32_t* fallbacks, uint32_t* late_answers = nullptr);
    // Shard-local search whose result stays packed (score bits << 32 | global row; ~0 padding) for the
    // cross-GPU exchange: out_packed_dev is [nq, k].  Fused tiers only (k <= 256, dim % 8 == 0).
    SearchError search_top_k_packed_device(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                           const uint64_t* allow_dev, uint64_t* out_packed_dev, hipStream_t stream);
    SearchError gather_dot(const float* query, uint32_t query_len, const uint32_t* rows, uint32_t n, float* out);
    // ... for MANY queries in one launch: out[i] = dot(queries[qidx[i]], row rows[i]) (host arrays; quality_scores_for_hits of a chunk)
    SearchError gather_dot_batched(const float* queries, uint32_t nq, uint32_t query_len, const uint32_t* rows, const uint32_t* qidx, uint32_t n,
                                   float* out);

    // A lone query in two halves (one at a time per index, on the index's own stream): begin enqueues and returns, end waits and
    // writes the hits — search_top_k(query, 1, ...) without a filter is exactly begin + end.  A row-sharded handle begins the query
    // on every shard before it ends any, so the shards' passes run side by side from ONE host thread.
    SearchError lone_exact_begin(const float* query, uint32_t k);
    SearchError lone_exact_end(uint32_t* out_rows, float* out_scores, uint32_t* out_count);   // [k], [k], [1]
    // ... of a two-pass search (bits 8: search_top_k_int8_two_pass, 4: search_top_k_4bit_two_pass), this shard's half: end yields
    // max(k * multiplier, k) candidate pairs (pass-1 entry | exact entry, aligned; kEmpty beyond the candidates)
    SearchError lone_two_pass_begin(const float* query, uint32_t k, uint32_t multiplier, int bits);
    SearchError lone_two_pass_end(uint64_t* out_approx, uint64_t* out_exact);

    // search_top_k + scan_wal + resolve_hits (search.rs:426-494, 1449-1475, 1493-1558): GPU top-k of the main
    // rows, host merge of the resident WAL entries, WAL shadowing and doc-id dedup.  Needs a doc-id table.
    SearchError search_hits(const float* query, uint32_t query_len, uint32_t k, uint32_t* out_rows, float* out_scores,
                            uint32_t* out_count);
    // VectorIndex::search_top_k_int8_two_pass (search.rs:514-661): int8 pass-1 over the lazily built int8 slab,
    // exact f16 rescore of the k*multiplier candidates; falls back to the exact search when a WAL is resident.
    SearchError search_top_k_int8_two_pass(const float* query, uint32_t query_len, uint32_t k, uint32_t multiplier,
                                           uint32_t* out_rows, float* out_scores, uint32_t* out_count);
    // VectorIndex::search_top_k_4bit_two_pass (search.rs:876-946): packed signed-nibble pass-1 (a quarter of the f16
    // bytes), exact f16 rescore; same fallbacks.
    SearchError search_top_k_4bit_two_pass(const float* query, uint32_t query_len, uint32_t k, uint32_t multiplier,
                                           uint32_t* out_rows, float* out_scores, uint32_t* out_count);
    // VectorIndex::mrl_search_with_stats (crates/frankensearch-index/src/mrl.rs:241-395): truncated scan over the first
    // search_dims dimensions (a strided view of the same slab), resident WAL entries, rescore over rescore_dims, top-k.
    struct MrlStats {
        uint32_t scan_dims = 0, rescore_dims = 0, candidates_rescored = 0;
        uint64_t records_scanned = 0;
        bool fell_back_to_full = false;
    };
    SearchError mrl_search(const float* query, uint32_t query_len, uint32_t k, uint32_t search_dims, uint32_t rescore_dims,
                           uint32_t rescore_top_k, uint32_t* out_rows, float* out_scores, uint32_t* out_count,
                           MrlStats* stats);
    // mrl_search for nq host queries at once: batched truncated scan on the matrix cores + one re-score launch (no stats)
    SearchError mrl_search_batched(const float* queries, uint32_t nq, uint32_t query_len, uint32_t k, uint32_t search_dims,
                                   uint32_t rescore_dims, uint32_t rescore_top_k, uint32_t* out_rows, float* out_scores,
                                   uint32_t* out_counts, uint32_t* fallbacks);
    // VectorIndex::append (lib.rs:2532-2720): resident WAL entry, immediately searchable.
    SearchError wal_append(const char* doc_id, uint32_t len, const float* vector, uint32_t vector_len);
    uint64_t wal_record_count() const { return wal_.size(); }
    SearchError doc_id_at(uint32_t row, const char** ptr, uint32_t* len) const;
    SearchError soft_delete(const char* doc_id, uint32_t len, int32_t* deleted);
    SearchError allow_bitmap_for_hashes(const uint64_t* hashes, uint32_t n, uint64_t* bitmap_out, uint64_t* matched) const;
    SearchError set_live_bitmap(const uint64_t* live);
    bool has_doc_ids() const { return !doc_offsets_.empty(); }
    // record-table accessors for the two-tier alignment (two_tier.rs:758-840) and quality_scores_for_hits (:1566-1631)
    bool row_tombstoned(uint64_t r) const { return !live_host_.empty() && !((live_host_[r >> 6] >> (r & 63)) & 1ull); }
    uint64_t doc_hash_at(uint64_t r) const { return doc_hashes_[r]; }
    // VectorIndex::find_index_by_doc_id (lib.rs:3395-3421): first LIVE main row with this doc id, -1 if none
    int64_t find_index_by_doc_id(const char* doc_id, uint32_t len) const;
    // latest resident WAL entry of a doc id (-1 if none) and its dot_product_f32_f32 with a query
    int64_t wal_latest(const char* doc_id, uint32_t len) const;
    float wal_dot(size_t wal_index, const float* query) const;
    const std::string& wal_doc_id(size_t wal_index) const { return wal_[wal_index].doc_id; }

    std::mutex& mutex() { return mu_; }
    int device() const { return device_; }
    int32_t hreduce = 0;
    int32_t variant = 0;
    uint64_t filter_gathered = 0, filter_scanned = 0;  // filtered host searches by path
    bool profiling = false;
    int profile_period = 1;   // fsgpu_index_set_profiling(n > 1): the merged main launch of every n-th batched step is timed
    uint32_t profile_tick_ = 0;
    // one-shot hook of the next batched search (fsgpu_index_set_after_enqueue_hook): called before the search blocks on its stream
    void (*after_enqueue_fn)(void*) = nullptr;
    void* after_enqueue_ctx = nullptr;
    // filter of the exact batched search (fsgpu_index_set_batched_filter): 0 = automatic, 1 = f16 slab, 2 = int8 slab
    int32_t batched_filter = 0;
    // fsgpu_index_set_int8_latency: unfiltered fsgpu_search_topk calls of a few queries go through the int8 filter too
    bool int8_latency = false;
    // fsgpu_index_set_filter_rotation: 0 = automatic (rotate the filter's copy when the slab has outlier channels), 1 = never, 2 = always.
    // Takes effect when the filter's copy is built (first batched search / fsgpu_index_int8_filter_bound).
    int32_t filter_rotation = 0;
    bool filter_rotated() const { return i8f_rot_; }
    bool exact_only_ = false;   // fsgpu_search_topk_exact: the call in flight takes the exact kernels whatever copies the index holds
    // Which bits of (score sortkey << 32 | ~row) can differ over this slab's rows: all of the score half, the row bits below the
    // highest one in which the first and the last row id differ — unless empty entries (key 0: tombstoned / filtered rows) are among them.
    uint64_t sortkey_varying_bits(bool may_hold_empty) const {
        if (may_hold_empty || nrows_ == 0) return ~0ull;
        const uint64_t lo = row_base_, hi = row_base_ + nrows_ - 1;
        uint64_t x = (lo ^ hi) & 0xffffffffull, mask = 0;
        while (x) {
            mask = (mask << 1) | 1ull;
            x >>= 1;
        }
        return 0xffffffff00000000ull | mask;
    }
    SearchError prepare_int8_latency();   // builds the int8 copy + its statistics now (else: the first batched search does)
    uint64_t i8f_queries = 0, i8f_refiltered = 0;  // queries the int8 filter took / handed on to the f16 filter
    bool int8_filter_active() const { return batched_filter == 2 || (batched_filter == 0 && !i8f_disabled_); }
    // The certificate of the int8 filter, for inspection: per query the bound delta on |int8 score - exact score x slab scale
    // x query scale| (< 0: not certifiable) and the query scale 127 / max|q|; the slab scale 127 / max|x|; the quantised queries
    // (nq x dim int8) and, when out_slab_i8 is given, the int8 slab (nrows x dim).  Builds the int8 slab if need be.
    SearchError int8_filter_bound(const float* queries, uint32_t nq, uint32_t query_len, float* out_delta, float* out_query_scale,
                                  float* out_slab_scale, int8_t* out_queries_i8, int8_t* out_slab_i8);
    // a row shard of a larger index: its max-abs for the cross-shard reduction, then the corpus-wide value adopted as THE scale
    SearchError compute_local_quant_max(unsigned int** max_bits_dev, hipStream_t stream);
    void adopt_global_quant_max();
    hipStream_t stream() const { return stream_; }
    VectorIndex* mrl_view(uint32_t dims);  // strided prefix view of this slab (created on first use)
    // Concurrent callers (the reference's scan is `&self`, lock-free, any number of callers: search.rs:192): replicas of this
    // index over the SAME slab and live bitmap, each with its own stream, workspaces and mutex, so that row-level searches from
    // different host threads run side by side instead of queueing on one stream.  The caller holds the owner's state lock.
    static constexpr size_t kLanes = 4;                // this index + 3 replicas
    SearchError ensure_replicas();                      // idempotent
    VectorIndex* replica(size_t i) { return i < replicas_.size() ? replicas_[i].get() : nullptr; }
    size_t replica_count() const { return replicas_.size(); }
    void sync_replicas();                               // after a mutation: live bitmap pointer, hreduce, variant
    SearchError scan_time(double* total_ms, uint64_t* launches, uint64_t* rows, bool reset);

  private:
    SearchError ensure_query_dimension(uint32_t query_len) const;
    SearchError open_fsvi_impl(const char* path, int device, FsviImage* image);
    void* pinned_io();
    SearchError batched_impl(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k, const uint64_t* allow_dev,
                             uint32_t* out_rows_dev, float* out_scores_dev, uint32_t* out_counts_dev, hipStream_t stream,
                             uint32_t* fallbacks, uint64_t* out_packed_dev, uint32_t int8_mult, uint32_t query_stride,
                             bool i8_filter, uint32_t* refiltered, int bits = 8);
    // ... and its stages (vector_index.cpp): what a call fixes for all its rounds, one round's geometry and arguments
    struct BatchedPlan;
    struct BatchedRound;
    SearchError batched_prepare(BatchedPlan& p, bool* done);
    SearchError batched_unusable(BatchedPlan& p);
    SearchError batched_round_setup(const BatchedPlan& p, BatchedRound& r, uint32_t g0);
    SearchError batched_sample(const BatchedPlan& p, BatchedRound& r);
    SearchError batched_main(const BatchedPlan& p, BatchedRound& r);
    SearchError batched_finish(BatchedPlan& p, BatchedRound& r);
    SearchError batched_fallback(BatchedPlan& p, bool already_waited = false);
    void i8f_account(uint32_t nq, uint32_t refiltered);
    SearchError quantized_two_pass(const float* query, uint32_t query_len, uint32_t k, uint32_t multiplier, int bits,
                                   uint32_t* out_rows, float* out_scores, uint32_t* out_count, u64* approx_out_dev = nullptr,
                                   u64* exact_out_dev = nullptr);
    // The lone query of the int8 latency path: ONE fused pass over the int8 slab that keeps the 256 best integer scores
    // (scan_i8_topk_kernel, 0.75 of HBM peak), their exact re-score, and a certificate — the 256th integer score lies more than
    // 2 delta below the k-th, so every row that could reach the exact top k was among them.  *certified = false: nothing was
    // written, the staged filter path answers.
    SearchError certified_i8_lone_query(const float* query, uint32_t k, uint32_t* out_rows, float* out_scores, uint32_t* out_count,
                                        bool* certified);
    SearchError two_pass_lone_certified(const float* query, const unsigned char* qi, uint32_t qbytes, uint32_t k, uint32_t k_eff, uint32_t cc,
                                        int bits, const void* qslab, uint32_t* rows, float* scores, uint32_t* count, bool* answered);
    // ... and the halves of both lanes (enqueue only / one synchronisation + the certificate)
    SearchError certified_i8_enqueue(const float* query, uint32_t k, bool* enqueued);
    SearchError certified_i8_check(uint32_t* out_rows, float* out_scores, uint32_t* out_count, bool* certified);
    SearchError two_pass_lone_enqueue(const float* query, const unsigned char* qi, uint32_t qbytes, uint32_t k, uint32_t k_eff, uint32_t cc,
                                      int bits, const void* qslab, bool want_pairs, bool* enqueued);
    SearchError two_pass_lone_check(uint32_t* rows, float* scores, uint32_t* count, u64* approx_out, u64* exact_out, bool* answered);
    SearchError ensure_two_pass_slab(int bits, const void** qslab);
    // the int8 filter's copy of the slab, its scale word and statistics: the reference's own int8 slab (shared with the two-pass
    // search), or a ROTATED copy of its own (vector_index.cpp, "the int8 filter's copy of the slab")
    static constexpr double kRotateRatio = 9.0;   // max |element| x sqrt(dim) / max row norm above which the copy is rotated
    bool filter_ready() const { return i8f_rot_ ? i8f_ready_ : (i8_ready_ && i8_stats_ready_); }
    const void* filter_slab() const { return i8f_rot_ ? i8f_slab_.ptr : i8_slab_.ptr; }
    const unsigned int* filter_max() const { return static_cast<const unsigned int*>(i8f_rot_ ? i8f_max_.ptr : i8_max_.ptr); }
    const unsigned int* filter_stats() const { return static_cast<const unsigned int*>(i8f_rot_ ? i8f_stats_.ptr : i8_stats_.ptr); }
    SearchError ensure_filter_copy(hipStream_t stream, bool must = false);   // decides the rotation on first use; builds what is missing
    SearchError prepare_filter_queries(const float* q, uint32_t nq, uint32_t nq_pad, uint32_t q_stride, void* qi8, float* delta, float* unit,
                                       hipStream_t stream);
    DeviceBuffer i8f_slab_, i8f_max_, i8f_stats_, rot_mat_, rot_q_;
    bool i8f_decided_ = false, i8f_rot_ = false, i8f_ready_ = false;
    double rot_extra_coeff_ = 0.0;
    enum LoneKind : int {
        kLoneNone = 0, kLoneEmpty, kLoneUnpinned, kLoneCertified, kLoneStaged, kLoneStagedBlocking, kLoneExact,
        kLoneTwoPassLane, kLoneTwoPassBatched, kLoneTwoPassBlocking
    };
    struct LoneState {   // the lone query in flight (lone_*_begin .. lone_*_end)
        int kind = kLoneNone;
        const float* query = nullptr;   // the caller's, valid until end
        uint32_t k = 0, mult = 0, cc = 0, cc_out = 0;
        int bits = 8;
        int32_t ticket = -1;
        bool staged_behind = false;     // a failed certificate is followed by the staged filter path (opt-in) / the exact kernels (default)
    };
    LoneState lone_;
    uint32_t cert_k_ = 0;   // k of the certified pass in flight
    struct TwoPassLane {    // the two-pass lane in flight: its k, candidate count and offsets into the pinned staging block
        uint32_t k = 0, cc = 0;
        size_t o_out = 0, o_flags = 0, o_approx = 0, o_exact = 0;
    };
    TwoPassLane tp_lane_;
    SearchError common_init(int device);
    SearchError fused_search(const float* queries_dev, uint32_t nq, uint32_t k_out, uint32_t k_eff,
                             const uint64_t* allow_dev, uint32_t* out_rows_dev, float* out_scores_dev,
                             uint32_t* out_counts_dev, u64* out_packed_dev, hipStream_t stream);
    hipError_t gather_dot_any(const ScanArgs& a, const uint32_t* rows, uint32_t n, float* out, hipStream_t stream) const;
    SearchError gather_search(const float* queries_dev, uint32_t nq, uint32_t k, const uint32_t* rows_dev, uint32_t n,
                              uint32_t* out_rows_dev, float* out_scores_dev, uint32_t* out_counts_dev, hipStream_t stream);
    SearchError general_search(const float* queries_dev, uint32_t nq, uint32_t k_out, uint32_t k_eff,
                               const uint64_t* allow_dev, uint32_t* out_rows_dev, float* out_scores_dev,
                               uint32_t* out_counts_dev, hipStream_t stream);
    ScanArgs base_args(const float* queries_dev, const uint64_t* allow_dev) const;

    std::mutex mu_;
    int device_ = -1;
    int num_cus_ = 256;
    uint32_t dim_ = 0;
    uint64_t nrows_ = 0;
    uint64_t row_base_ = 0;
    uint32_t row_stride_ = 0;  // bytes between rows; dim_*2 except for the MRL prefix views
    std::map<uint32_t, std::unique_ptr<VectorIndex>> views_;  // strided prefix views of this slab, by dimension
    std::vector<std::unique_ptr<VectorIndex>> replicas_;      // lanes for concurrent row-level searches
    const void* slab_dev_ = nullptr;
    const uint64_t* live_dev_ = nullptr;
    bool owns_slab_ = false;
    bool f32_ = false;  // Quantization::F32 slab: served by the general path (f32_kernels.hip)
    bool catalog_only_ = false;  // the tables of a sharded index: no device state of its own
    DeviceBuffer slab_own_, live_own_;
    hipStream_t stream_ = nullptr;
    // workspaces (grown on demand, reused)
    DeviceBuffer ws_partial_, ws_queries_, ws_allow_, ws_rows_, ws_scores_, ws_counts_, ws_keys_a_, ws_keys_b_,
        ws_sort_tmp_, ws_gather_rows_, ws_gather_out_, i8_slab_, n4_slab_, i8_max_, ws_i8_query_, ws_cand_packed_,
        ws_cand_rows_, ws_cand_scores_, mf_max_norm_, mf_qh_, mf_delta_, mf_tau_, mf_cand_, mf_dense_, mf_sel_,
        mf_fallback_, mf_fallback2_, mf_spill_, mf_io_, mf_io2_, i8_stats_, n4u_slab_, mf_cand_count_, ws_pairs_;
    DeviceBuffer ws_out_;   // rows | scores | counts of a blocking batched search (one block: one copy up)
    bool i8_ready_ = false, n4_ready_ = false, i8_stats_ready_ = false, n4u_ready_ = false;
    bool quant_max_ready_ = false;   // i8_max_ holds a corpus-wide max-abs handed in by a sharded index: the quantisers keep it
    u64* tp_approx_out_ = nullptr;   // two_pass_candidates_device: where the batch in flight leaves its candidate pairs
    u64* tp_exact_out_ = nullptr;
    uint32_t tp_stride_ = 0;         // entries between queries in both
    bool hard_batch_ = false;     // the batch in flight is the int8 filter's leftovers (nested f16-filter call)
    bool i8f_disabled_ = false;   // the int8 filter left too many queries uncertified on this slab (or its copy does not fit)
    uint32_t i8f_strikes_ = 0;
    uint32_t cert_skip_ = 0, cert_backoff_ = 0;   // the lone query's single-pass certificate: calls still to skip / the current back-off
    uint32_t tp_skip_ = 0, tp_backoff_ = 0;       // ... and the two-pass searches' lone-caller lane
    // search_top_k_batched_device_begin / _end: the ticket being begun (-1: none), per ticket 0 free / 1 plan parked / 2 finished inside begin
    int async_want_ = -1;
    uint8_t async_state_[2] = {0, 0};
    bool async_i8f_[2] = {false, false};
    uint32_t async_nq_[2] = {0, 0}, async_fb_[2] = {0, 0};
    hipEvent_t async_ev_[2] = {nullptr, nullptr};
    hipStream_t async_stream_[2] = {nullptr, nullptr};   // the stream each outstanding ticket was enqueued on
    std::vector<unsigned char> async_plan_[2];   // the parked BatchedPlan (plain data: pointers and sizes), defined in the .cpp
    uint32_t i8f_sample_boost_ = 1;   // 1 or 2: the second sample of the int8 filter's wide rounds grows before the filter is given up
    bool mf_norm_ready_ = false;
    int mf_shape_i8_ = 4, mf_per_cu_160_ = 1, mf_per_cu_160_i8_ = 1;
    bool mf_use_160_ = false;
    int mf_per_cu_wide_main_ = 1;  // mfma_wide.hip: one 512-thread block per CU (its LDS ring + candidate lists take ~130 KB)
    int mf_shape_ = -1, mf_per_cu_narrow_ = 1, mf_per_cu_wide_ = 1, mf_per_cu_narrow_i8_ = 1, mf_per_cu_wide_i8_ = 1;  // batched-scan launch shapes (probed once)
    static constexpr size_t kPinnedIoBytes = 256 * 1024;
    void* io_host_ = nullptr;  // pinned staging for the single-query latency paths
    void* batch_io_host_ = nullptr;   // pinned block the results of batched searches come up through (pinned_batch_io)
    size_t batch_io_bytes_ = 0;
    bool batch_io_failed_ = false;
    void* pinned_batch_io(size_t bytes);
    SearchError fetch_batched_results(const unsigned char* base, size_t o_rows, size_t o_scores, size_t o_counts, size_t total, uint32_t nq,
                                      uint32_t k, uint32_t* out_rows, float* out_scores, uint32_t* out_counts);
    const float* host_query_hint_ = nullptr;   // search_top_k's lone query, still in host memory: fused_search launches the scan with it in the argument block
    bool io_failed_ = false;
    uint32_t* mf_flags_host_ = nullptr;                              // pinned per-query verdicts of the batched scan
    uint32_t mf_flags_cap_ = 0;
    // profiling events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events_;
    uint32_t mf_pass_parity_ = 0;  // main passes of the batched path alternate their direction over the slab
    uint64_t profiled_rows_ = 0;  // slab rows streamed by the timed launches
    uint32_t profiled_elem_bytes_ = 2;
    // FSVI host-side tables
    std::vector<uint64_t> live_host_;
    std::vector<uint64_t> doc_hashes_;
    std::vector<uint64_t> doc_offsets_;  // nrows+1 offsets into doc_blob_
    std::string doc_blob_;
    // resident WAL entries (crates/frankensearch-index/src/wal.rs WalEntry)
    struct WalEntry {
        std::string doc_id;
        std::vector<float> embedding;
    };
    std::vector<WalEntry> wal_;
};

class Model2VecEmbedder {
  public:
    ~Model2VecEmbedder();
    SearchError init(int device, const float* table, uint32_t vocab, uint32_t dim);
    // out_dev (on this embedder's device, may be null): the vectors are left in device memory; out (may then be null): host copy
    SearchError embed_batch(const uint32_t* ids, const uint32_t* offsets, uint32_t n, float* out, float* out_dev = nullptr);
    uint32_t dimension() const { return dim_; }
    int device() const { return device_; }

  private:
    std::mutex mu_;
    int device_ = -1;
    uint32_t vocab_ = 0, dim_ = 0;
    DeviceBuffer table_, ids_, offsets_, out_;
    hipStream_t stream_ = nullptr;
};

SearchError merge_packed_lists_device(int device, const uint64_t* lists_dev, uint32_t nq, uint32_t nlists,
                                      uint32_t list_len, uint64_t q_stride, uint64_t l_stride, uint32_t k,
                                      uint32_t* out_rows_dev, float* out_scores_dev, uint32_t* out_counts_dev,
                                      hipStream_t stream);

}  // namespace fsgpu
