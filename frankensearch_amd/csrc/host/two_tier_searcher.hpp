// two_tier_searcher.hpp — C++ restatement of the reference's synchronous two-phase searcher over the fsgpu C ABI
// (crates/frankensearch-fusion/src/sync_searcher.rs:616-943).  Host code only; see include/fshost.h.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/fshost.h"

namespace fshost {

struct Hit {  // VectorHit / ScoredResult with an owned doc id (crates/frankensearch-core/src/types.rs:88-134)
    std::string doc_id;
    float score;
    uint32_t index;
};

struct Outcome {
    std::vector<fshost_hit> initial, final_results;
    fshost_metrics metrics{};
    // SearchPhase::RefinementFailed (sync_searcher.rs:820-839): the quality pool could not be produced; final_results are the
    // initial results and the search still succeeds
    bool refinement_failed = false;
    std::string skip_reason;
};

// One tier of the pair as the searcher uses it: an fsgpu_index, or an fsgpu_sharded handle over the GPUs of the node (SURVEY 8e:
// the fast and quality slabs shard identically; the host makes the same calls either way).
struct Tier {
    fsgpu_index* index = nullptr;
    fsgpu_sharded* sharded = nullptr;
    uint32_t dimension() const { return index ? fsgpu_index_dimension(index) : fsgpu_sharded_dimension(sharded); }
    // VectorIndex::search_top_k(query, fetch, None) -> row-level hits, best first (int8_multiplier != 0: search_top_k_int8_two_pass)
    fsgpu_status search_rows(const float* query, uint32_t len, uint32_t fetch, uint32_t int8_multiplier, uint32_t* rows, float* scores,
                             uint32_t* count) const;
    // search_top_k + scan_wal + resolve_hits (doc-id table, WAL, dedup)
    fsgpu_status search_hits(const float* query, uint32_t len, uint32_t fetch, uint32_t* rows, float* scores, uint32_t* count) const;
    fsgpu_status doc_id(uint32_t row, const char** ptr, uint32_t* len) const;
};

class ManyEngine;

class SyncTwoTierSearcher {
  public:
    SyncTwoTierSearcher(fsgpu_index* fast, fsgpu_index* quality, fsgpu_m2v* fast_embedder, fsgpu_bert* quality_embedder,
                        const fshost_two_tier_config& cfg);
    // both tiers behind row-sharded handles (fshost_two_tier_create_sharded)
    SyncTwoTierSearcher(fsgpu_sharded* fast, fsgpu_sharded* quality, fsgpu_m2v* fast_embedder, fsgpu_bert* quality_embedder,
                        const fshost_two_tier_config& cfg);
    ~SyncTwoTierSearcher();
    SyncTwoTierSearcher(const SyncTwoTierSearcher&) = delete;
    SyncTwoTierSearcher& operator=(const SyncTwoTierSearcher&) = delete;
    fsgpu_status init_status() const { return init_status_; }
    const std::string& init_detail() const { return init_detail_; }
    // Returns an fsgpu status; `detail` is filled on failure.
    fsgpu_status search(const uint32_t* fast_ids, uint32_t n_fast, const int32_t* quality_ids, uint32_t n_quality, uint32_t k,
                        const fsgpu_scored_doc* lexical, uint32_t n_lexical, Outcome* out, std::string* detail) const;
    // ... the flow itself, one query on the calling thread (search() routes here unless dynamic batching is on)
    fsgpu_status search_unbatched(const uint32_t* fast_ids, uint32_t n_fast, const int32_t* quality_ids, uint32_t n_quality, uint32_t k,
                                  const fsgpu_scored_doc* lexical, uint32_t n_lexical, Outcome* out, std::string* detail) const;

    // The same flow for MANY queries at once (fshost_two_tier_search_many; two_tier_many.cpp): batched embeds and batched tier
    // searches, pipelined over chunks, the per-query fusion (the very functions search() runs) on a pool of host threads.
    struct ManyArgs {
        const uint32_t* fast_ids = nullptr;
        const uint32_t* fast_offsets = nullptr;      // [nq + 1]
        const int32_t* quality_ids = nullptr;
        const uint32_t* quality_offsets = nullptr;   // [nq + 1]
        uint32_t nq = 0, k = 0;
        const fsgpu_scored_doc* lexical = nullptr;   // every query's list back to back, or null
        const uint32_t* lexical_offsets = nullptr;   // [nq + 1]
        uint32_t chunk = 0;                          // queries per pipeline step (0: 1,024)
        uint32_t fusion_threads = 0;                 // 0: automatic
        fshost_hit* initial_out = nullptr;           // [nq * k]
        uint32_t* n_initial = nullptr;               // [nq]
        fshost_hit* final_out = nullptr;
        uint32_t* n_final = nullptr;
        uint8_t* refinement_failed = nullptr;        // [nq] or null
        float* quality_vectors_out = nullptr;        // [nq, quality dim] or null: the embeddings phase 1 searched with
        float* fast_vectors_out = nullptr;           // [nq, fast dim] or null
    };
    fsgpu_status search_many(const ManyArgs& a, fshost_many_result* result) const;
    // Dynamic batching of concurrent search() callers through the same engine (fshost_two_tier_set_batching): max_chunk = 0 turns it off.
    fsgpu_status set_batching(uint32_t max_chunk, uint32_t max_wait_us);
    void batching_stats(uint64_t* chunks, uint64_t* requests) const;

  private:
    friend class ManyEngine;
    ManyEngine* engine(uint32_t fusion_threads) const;   // created on first use
    void stop_engine();
    void init();
    fsgpu_status tier_hits(const Tier& tier, const std::vector<float>& vec, uint32_t fetch, uint32_t int8_multiplier,
                           std::vector<Hit>* hits, std::string* detail) const;
    fsgpu_status hits_from_rows(const Tier& tier, const uint32_t* rows, const float* scores, uint32_t count, std::vector<Hit>* hits,
                                std::string* detail) const;
    fsgpu_status fuse_initial(const std::vector<Hit>& fast_hits, uint32_t k, const fsgpu_scored_doc* lexical, uint32_t n_lexical,
                              std::vector<fshost_hit>* initial, std::string* detail) const;
    fsgpu_status fuse_final_retrieved(const std::vector<Hit>& fast_hits, const std::vector<Hit>& quality_hits, uint32_t k,
                                      const fsgpu_scored_doc* lexical, uint32_t n_lexical, std::vector<fshost_hit>* final_results,
                                      std::string* detail) const;
    fsgpu_status fuse_final_rescored(const std::vector<Hit>& fast_hits, const float* quality_vec, uint32_t k, const fsgpu_scored_doc* lexical,
                                     uint32_t n_lexical, std::vector<fshost_hit>* final_results, bool* failed, std::string* detail) const;
    fsgpu_status fuse_final_rescored_scores(const std::vector<Hit>& fast_hits, const float* qscores, const uint8_t* qpresent, uint32_t k,
                                            const fsgpu_scored_doc* lexical, uint32_t n_lexical, std::vector<fshost_hit>* final_results,
                                            std::string* detail) const;
    Tier fast_, quality_;
    fsgpu_m2v* m2v_;
    fsgpu_bert* bert_;
    fshost_two_tier_config cfg_;
    uint32_t fast_dim_, quality_dim_, quality_dim_vec_ = 0;
    fsgpu_alignment* alignment_ = nullptr;   // quality_pool == FSHOST_POOL_RESCORED: QualityAlignment of the pair, computed once
    fsgpu_status init_status_ = FSGPU_OK;
    std::string init_detail_;
    mutable std::mutex engine_mu_;
    mutable std::unique_ptr<ManyEngine> engine_;
    std::atomic<bool> batching_{false};
};

// The many-queries engine (two_tier_many.cpp): four stage threads + a fusion pool over the searcher's handles.
class ManyEngine {
  public:
    ManyEngine(const SyncTwoTierSearcher& s, uint32_t fusion_threads);
    ~ManyEngine();
    fsgpu_status run_many(const SyncTwoTierSearcher::ManyArgs& a, fshost_many_result* res);
    void configure_batching(uint32_t max_chunk, uint32_t max_wait_us);
    fsgpu_status search_one(const uint32_t* fast_ids, uint32_t n_fast, const int32_t* quality_ids, uint32_t n_quality, uint32_t k,
                            const fsgpu_scored_doc* lexical, uint32_t n_lexical, Outcome* out, std::string* detail);
    void batching_stats(uint64_t* chunks, uint64_t* requests);

  private:
    struct Query;
    struct Waiter;
    struct Batch;
    struct Chunk;
    struct Task;
    struct Slot {
        float* f = nullptr;   // fast-tier vectors of a chunk (on the fast tier's device)
        float* q = nullptr;   // quality-tier vectors
        uint32_t cap = 0;
    };
    static constexpr int kSlots = 3;   // chunks whose embeddings may sit in device memory at once
    int acquire_slot(uint32_t cap);
    void release_slot(int slot);
    void submit(Chunk* c);
    Chunk* pop(std::deque<Chunk*>& q);
    void embed_stage(bool fast);
    void search_stage(bool fast);
    void rescore_stage();
    fsgpu_status tier_search(const Tier& tier, bool rowlevel, uint32_t int8_mult, const float* vec_dev, const float* vec_host, uint32_t n, uint32_t dim,
                             uint32_t fetch, uint32_t* rows, float* scores, uint32_t* counts, uint32_t* fb, std::string* detail);
    void push_tasks(Chunk* c, bool final);
    void final_part_ready(Chunk* c);
    void fusion_worker();
    void finish_chunk(Chunk* c);
    void collector();
    void lone_lane();

    const SyncTwoTierSearcher& s_;
    uint32_t n_workers_ = 0, fdim_ = 0, qdim_ = 0;
    bool rescored_ = false, fast_rowlevel_ = false, quality_rowlevel_ = false, fast_dev_ok_ = false, quality_dev_ok_ = false;
    int32_t fast_dev_ = -1, quality_dev_ = -1;
    Slot slots_[kSlots];
    // stage queues + free slots
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Chunk*> fe_q_, fs_q_, qe_q_, qs_q_, rs_q_;
    std::deque<int> free_slots_;
    bool stop_ = false;
    // fusion pool
    std::mutex tmu_;
    std::condition_variable tcv_;
    std::deque<Task> tasks_;
    bool pool_stop_ = false;
    // dynamic batching of per-query callers
    std::mutex rmu_;
    std::condition_variable rcv_;
    std::deque<Waiter*> requests_;
    std::chrono::steady_clock::time_point last_arrival_;
    uint32_t max_chunk_ = 0, max_wait_us_ = 0;
    uint64_t server_chunks_ = 0, server_requests_ = 0;
    bool server_stop_ = false;
    std::atomic<uint64_t> in_flight_{0};   // requests handed to the pipeline (or the lone lane) and not yet answered
    std::mutex lmu_;
    std::condition_variable lcv_;
    std::deque<Waiter*> lone_q_;
    bool lone_stop_ = false;
    std::atomic<bool> lone_busy_{false};
    std::thread collector_, lone_thread_;
    std::vector<std::thread> threads_;
};

}  // namespace fshost
