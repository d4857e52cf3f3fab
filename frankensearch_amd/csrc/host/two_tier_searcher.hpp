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

  private:
    void init();
    fsgpu_status tier_hits(const Tier& tier, const std::vector<float>& vec, uint32_t fetch, uint32_t int8_multiplier,
                           std::vector<Hit>* hits, std::string* detail) const;
    Tier fast_, quality_;
    fsgpu_m2v* m2v_;
    fsgpu_bert* bert_;
    fshost_two_tier_config cfg_;
    uint32_t fast_dim_, quality_dim_;
    fsgpu_alignment* alignment_ = nullptr;   // quality_pool == FSHOST_POOL_RESCORED: QualityAlignment of the pair, computed once
    fsgpu_status init_status_ = FSGPU_OK;
    std::string init_detail_;
};

}  // namespace fshost
