// two_tier_index.hpp — the pairing of a fast and a quality VectorIndex (crates/frankensearch-index/src/two_tier.rs):
//   QualityAlignment            two_tier.rs:404-409   None / Aligned (fast row i is quality row i) / Mapping (per fast row)
//   the alignment walk          two_tier.rs:750-866   both record tables are sorted by (FNV-1a(doc_id), doc_id): one merge pass,
//                                                     tombstoned rows skipped on either side, Aligned until the first divergence
//   quality_scores_for_hits     two_tier.rs:1566-1631 per fast hit: the quality WAL's latest entry of the doc id, else the aligned
//                                                     quality row (dot_query_at), else the quality index's own row of the doc id
// The dots over main rows run as ONE gather_dot launch on the quality index's device.
#pragma once

#include <cstdint>
#include <functional>
#include <vector>

#include "vector_index.hpp"

namespace fsgpu {

class QualityAlignment {
  public:
    enum Kind : int32_t { kNone = 0, kAligned = 1, kMapping = 2 };
    // Raw slabs (no record table) pair by row: Aligned.  Indexes with record tables take the reference's merge walk.
    SearchError build(const VectorIndex& fast, const VectorIndex& quality);
    // the same walk over the tables of a row-sharded pair: each side is the handle's catalog (record table, doc ids, tombstones;
    // sharded_index.hpp) or null for raw shards, which pair by row like raw slabs
    SearchError build(const VectorIndex* fast_table, uint64_t fast_rows, const VectorIndex* quality_table, uint64_t quality_rows);
    Kind kind() const { return kind_; }
    // quality_index_for_fast_index (two_tier.rs:1975-1981): -1 = none
    int64_t quality_row(uint64_t fast_row) const;
    uint64_t unmatched_quality_docs() const { return unmatched_; }

  private:
    Kind kind_ = kNone;
    uint64_t fast_rows_ = 0;
    std::vector<int64_t> map_;  // kMapping: per fast row
    uint64_t unmatched_ = 0;
};

struct HitRef {
    const char* doc_id;
    uint32_t doc_id_len;
    uint32_t index;  // fast-tier row, or 0xffffffff
};

// The quality tier as the re-scoring sees it: the tables that resolve a doc id (the index itself, a sharded handle's catalog, or
// null for raw slabs) and a gather of dot_query_at over GLOBAL main rows (VectorIndex::gather_dot; ShardedIndex::gather_dot routes
// every row to the shard that owns it).
struct QualityTierView {
    const VectorIndex* table = nullptr;
    uint64_t rows = 0;
    uint32_t dim = 0;
    std::function<SearchError(const float* query, uint32_t query_len, const uint32_t* rows, uint32_t n, float* out)> gather;
};
SearchError quality_scores_for_hits(const VectorIndex* fast_table, uint64_t fast_rows, const QualityTierView& quality,
                                    const QualityAlignment& align, const float* query, uint32_t query_len, const HitRef* hits,
                                    uint32_t n, float* out_scores, uint8_t* out_present);

// out_scores[i] is valid iff out_present[i] != 0.  Takes the quality index's mutex for the gather.
SearchError quality_scores_for_hits(const VectorIndex& fast, VectorIndex& quality, const QualityAlignment& align,
                                    const float* query, uint32_t query_len, const HitRef* hits, uint32_t n, float* out_scores,
                                    uint8_t* out_present);

// quality_scores_for_hits for a chunk of queries (the many-queries two-tier flow): query q owns hits[hit_offsets[q] .. hit_offsets[q + 1]);
// the per-hit resolution is the function above's, the dots over main rows of ALL the queries run as ONE multi-query gather launch.
// out_scores / out_present are flat, aligned with `hits`.
SearchError quality_scores_for_hits_batched(const VectorIndex& fast, VectorIndex& quality, const QualityAlignment& align, const float* queries,
                                            uint32_t nq, uint32_t query_len, const HitRef* hits, const uint32_t* hit_offsets, float* out_scores,
                                            uint8_t* out_present);

}  // namespace fsgpu
