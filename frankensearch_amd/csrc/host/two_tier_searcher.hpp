// two_tier_searcher.hpp — C++ restatement of the reference's synchronous two-phase searcher over the fsgpu C ABI
// (crates/frankensearch-fusion/src/sync_searcher.rs:616-943).  Host code only; see include/fshost.h.
#pragma once

#include <cstdint>
#include <string>
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

  private:
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
    Tier fast_, quality_;
    fsgpu_m2v* m2v_;
    fsgpu_bert* bert_;
    fshost_two_tier_config cfg_;
    uint32_t fast_dim_, quality_dim_, quality_dim_vec_ = 0;
    fsgpu_alignment* alignment_ = nullptr;   // quality_pool == FSHOST_POOL_RESCORED: QualityAlignment of the pair, computed once
    fsgpu_status init_status_ = FSGPU_OK;
    std::string init_detail_;
};

}  // namespace fshost
