// two_tier_index.cpp — see two_tier_index.hpp.
#include "two_tier_index.hpp"

#include <cstring>
#include <mutex>
#include <string_view>
#include <unordered_map>

#include "../../include/fsgpu.h"

namespace fsgpu {

namespace {
SearchError err(int32_t code, std::string detail) {
    SearchError e;
    e.code = code;
    e.detail = std::move(detail);
    return e;
}
std::string_view doc_of(const VectorIndex& idx, uint64_t row) {
    const char* p = nullptr;
    uint32_t len = 0;
    (void)idx.doc_id_at((uint32_t)row, &p, &len);
    return std::string_view(p, len);
}
}  // namespace

SearchError QualityAlignment::build(const VectorIndex& fast, const VectorIndex& quality) {
    return build(&fast, fast.record_count(), &quality, quality.record_count());
}

SearchError QualityAlignment::build(const VectorIndex* fast_table, uint64_t f_count, const VectorIndex* quality_table, uint64_t q_count) {
    fast_rows_ = f_count;
    map_.clear();
    unmatched_ = 0;
    if (!fast_table || !quality_table || !fast_table->has_doc_ids() || !quality_table->has_doc_ids()) {
        // raw slabs: row i of one tier is row i of the other (the writer emits both tiers from one document order)
        kind_ = kAligned;
        if (q_count < f_count) {   // fast rows past the quality tier have no quality vector
            kind_ = kMapping;
            map_.resize(f_count);
            for (uint64_t i = 0; i < f_count; ++i) map_[i] = i < q_count ? (int64_t)i : -1;
        }
        return SearchError{};
    }
    const VectorIndex& fast = *fast_table;
    const VectorIndex& quality = *quality_table;
    kind_ = kAligned;
    uint64_t f = 0, q = 0;
    auto ensure_mapping = [&](uint64_t upto) {
        if (kind_ == kAligned) {
            kind_ = kMapping;
            map_.resize(upto);
            for (uint64_t i = 0; i < upto; ++i) map_[i] = (int64_t)i;
        }
    };
    auto push_none = [&](uint64_t at) {
        ensure_mapping(at);
        map_.push_back(-1);
    };
    while (f < f_count && q < q_count) {
        if (fast.row_tombstoned(f)) {
            push_none(f);
            ++f;
            continue;
        }
        if (quality.row_tombstoned(q)) {
            ++q;
            continue;
        }
        if (kind_ == kAligned && f != q) ensure_mapping(f);
        const uint64_t fh = fast.doc_hash_at(f), qh = quality.doc_hash_at(q);
        if (fh < qh) {            // fast has the doc, quality lacks it
            push_none(f);
            ++f;
        } else if (fh > qh) {
            ++unmatched_;
            ++q;
        } else {
            const std::string_view fd = doc_of(fast, f), qd = doc_of(quality, q);
            const int c = fd.compare(qd);
            if (c == 0) {
                if (kind_ == kMapping) map_.push_back((int64_t)q);
                ++f;
                ++q;
            } else if (c < 0) {
                push_none(f);
                ++f;
            } else {
                ++unmatched_;
                ++q;
            }
        }
    }
    if (f < f_count) {            // trailing fast docs
        ensure_mapping(f);
        map_.resize(f_count, -1);
    }
    for (; q < q_count; ++q)
        if (!quality.row_tombstoned(q)) ++unmatched_;
    return SearchError{};
}

int64_t QualityAlignment::quality_row(uint64_t fast_row) const {
    if (fast_row >= fast_rows_) return -1;   // score_quality_for_fast_index: fast_idx >= doc_count -> None
    switch (kind_) {
        case kAligned: return (int64_t)fast_row;
        case kMapping: return fast_row < map_.size() ? map_[fast_row] : -1;
        default: return -1;
    }
}

SearchError quality_scores_for_hits(const VectorIndex* fast_table, uint64_t fast_rows, const QualityTierView& quality,
                                    const QualityAlignment& align, const float* query, uint32_t query_len, const HitRef* hits,
                                    uint32_t n, float* out_scores, uint8_t* out_present) {
    if (query_len != quality.dim)
        return err(FSGPU_ERR_DIMENSION_MISMATCH, "expected " + std::to_string(quality.dim) + ", found " + std::to_string(query_len));
    std::vector<uint32_t> rows, slot;
    rows.reserve(n);
    slot.reserve(n);
    const VectorIndex* qt = quality.table;
    const bool ids = qt && qt->has_doc_ids();
    // doc id -> latest resident WAL entry, once per call (two_tier.rs:1583-1595): a forward insert keeps the highest index, which
    // is what a reverse scan per hit would find; an empty WAL (the compacted, common case) costs nothing
    std::unordered_map<std::string_view, size_t> wal_latest;
    if (ids) {
        const size_t w = (size_t)qt->wal_record_count();
        wal_latest.reserve(w);
        for (size_t i = 0; i < w; ++i) wal_latest[qt->wal_doc_id(i)] = i;
    }
    for (uint32_t i = 0; i < n; ++i) {
        out_present[i] = 0;
        out_scores[i] = 0.0f;
        const HitRef& h = hits[i];
        // the quality WAL's latest entry of this document wins (resident f32 vector, host dot in the reference's order)
        if (!wal_latest.empty() && h.doc_id) {
            auto it = wal_latest.find(std::string_view(h.doc_id, h.doc_id_len));
            if (it != wal_latest.end()) {
                out_scores[i] = qt->wal_dot(it->second, query);
                out_present[i] = 1;
                continue;
            }
        }
        int64_t fast_idx = -1;
        if (h.index == 0xffffffffu) {
            if (fast_table && fast_table->has_doc_ids() && h.doc_id) fast_idx = fast_table->find_index_by_doc_id(h.doc_id, h.doc_id_len);
        } else if (h.index < fast_rows) {
            fast_idx = h.index;
        }
        int64_t qrow = fast_idx >= 0 ? align.quality_row((uint64_t)fast_idx) : -1;
        if (qrow >= 0 && (uint64_t)qrow >= quality.rows)   // dot_query_at surfaces this (lib.rs:3229-3239): not a silent None
            return err(FSGPU_ERR_INVALID_CONFIG, "quality row " + std::to_string(qrow) + " of fast row " + std::to_string(fast_idx) +
                                                 " is out of range for dot_query_at (" + std::to_string(quality.rows) + " records)");
        if (qrow < 0 && ids && h.doc_id) qrow = qt->find_index_by_doc_id(h.doc_id, h.doc_id_len);
        if (qrow >= 0) {
            rows.push_back((uint32_t)qrow);
            slot.push_back(i);
        }
    }
    if (!rows.empty()) {
        std::vector<float> dots(rows.size());
        SearchError e = quality.gather(query, query_len, rows.data(), (uint32_t)rows.size(), dots.data());
        if (!e.ok()) return e;
        for (size_t j = 0; j < rows.size(); ++j) {
            out_scores[slot[j]] = dots[j];
            out_present[slot[j]] = 1;
        }
    }
    return SearchError{};
}

SearchError quality_scores_for_hits(const VectorIndex& fast, VectorIndex& quality, const QualityAlignment& align,
                                    const float* query, uint32_t query_len, const HitRef* hits, uint32_t n, float* out_scores,
                                    uint8_t* out_present) {
    std::lock_guard<std::mutex> lock(quality.mutex());
    QualityTierView view;
    view.table = &quality;
    view.rows = quality.record_count();
    view.dim = quality.dimension();
    view.gather = [&quality](const float* q, uint32_t len, const uint32_t* rows, uint32_t cnt, float* out) {
        return quality.gather_dot(q, len, rows, cnt, out);
    };
    return quality_scores_for_hits(&fast, fast.record_count(), view, align, query, query_len, hits, n, out_scores, out_present);
}

SearchError quality_scores_for_hits_batched(const VectorIndex& fast, VectorIndex& quality, const QualityAlignment& align, const float* queries,
                                            uint32_t nq, uint32_t query_len, const HitRef* hits, const uint32_t* hit_offsets, float* out_scores,
                                            uint8_t* out_present) {
    std::lock_guard<std::mutex> lock(quality.mutex());
    // pass 1: every query's resolution (WAL entries scored on the host, alignment, doc-id lookups) with a gather that only RECORDS which
    // rows it was asked for; pass 2: one launch over all of them; pass 3: the dots land where the single-query function would have put them
    std::vector<uint32_t> all_rows, all_q;
    std::vector<uint32_t> first(nq + 1, 0);
    QualityTierView view;
    view.table = &quality;
    view.rows = quality.record_count();
    view.dim = quality.dimension();
    uint32_t cur = 0;
    view.gather = [&](const float*, uint32_t, const uint32_t* rows, uint32_t cnt, float* out) {
        for (uint32_t i = 0; i < cnt; ++i) {
            all_rows.push_back(rows[i]);
            all_q.push_back(cur);
            out[i] = 0.0f;
        }
        return SearchError{};
    };
    for (uint32_t q = 0; q < nq; ++q) {
        cur = q;
        first[q] = (uint32_t)all_rows.size();
        const uint32_t h0 = hit_offsets[q], n = hit_offsets[q + 1] - h0;
        SearchError e = quality_scores_for_hits(&fast, fast.record_count(), view, align, queries + (size_t)q * query_len, query_len, hits + h0, n,
                                                out_scores + h0, out_present + h0);
        if (!e.ok()) return e;
    }
    first[nq] = (uint32_t)all_rows.size();
    if (all_rows.empty()) return SearchError{};
    std::vector<float> dots(all_rows.size());
    SearchError e = quality.gather_dot_batched(queries, nq, query_len, all_rows.data(), all_q.data(), (uint32_t)all_rows.size(), dots.data());
    if (!e.ok()) return e;
    // the single-query function gathers a query's main rows in hit order, skipping the hits it answered from the WAL or found no row
    // for: replay that order (a hit took part in the gather iff it is present and was not scored by the WAL — its score is the 0 the
    // recording gather wrote)
    for (uint32_t q = 0; q < nq; ++q) {
        cur = q;
        uint32_t j = first[q];
        // run the resolution again with a gather that hands out this query's slice of the dots
        view.gather = [&](const float*, uint32_t, const uint32_t*, uint32_t cnt, float* out) {
            for (uint32_t i = 0; i < cnt; ++i) out[i] = dots[j + i];
            return SearchError{};
        };
        const uint32_t h0 = hit_offsets[q], n = hit_offsets[q + 1] - h0;
        SearchError e2 = quality_scores_for_hits(&fast, fast.record_count(), view, align, queries + (size_t)q * query_len, query_len, hits + h0, n,
                                                 out_scores + h0, out_present + h0);
        if (!e2.ok()) return e2;
    }
    return SearchError{};
}

}  // namespace fsgpu
