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
};

class SyncTwoTierSearcher {
  public:
    SyncTwoTierSearcher(fsgpu_index* fast, fsgpu_index* quality, fsgpu_m2v* fast_embedder, fsgpu_bert* quality_embedder,
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
    fsgpu_status tier_hits(fsgpu_index* index, const std::vector<float>& vec, uint32_t fetch, uint32_t int8_multiplier,
                           std::vector<Hit>* hits, std::string* detail) const;
    fsgpu_index* fast_;
    fsgpu_index* quality_;
    fsgpu_m2v* m2v_;
    fsgpu_bert* bert_;
    fshost_two_tier_config cfg_;
    uint32_t fast_dim_, quality_dim_;
    fsgpu_alignment* alignment_ = nullptr;   // quality_pool == FSHOST_POOL_RESCORED: QualityAlignment of the pair, computed once
    fsgpu_status init_status_ = FSGPU_OK;
    std::string init_detail_;
};

}  // namespace fshost
