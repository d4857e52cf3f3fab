// fusion.cpp — host-side rank fusion used after the GPU searches: reciprocal-rank fusion and the two-tier
// score blend.  O(k) work on a few hundred hits, so it stays on the CPU exactly as in the reference:
//   rrf_fuse        crates/frankensearch-fusion/src/rrf.rs:368-560 (cmp_for_ranking :179-198, sanitisers :85-138)
//   blend_two_tier  crates/frankensearch-fusion/src/blend.rs:107-195 (NormBounds :24-75, sanitisers :518-532)
// Exposed through the C ABI (fsgpu_rrf_fuse / fsgpu_blend_two_tier) so the end-to-end two-tier query path can be
// driven without the Rust crate.  No GPU needed.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/fsgpu.h"

namespace {

constexpr double kDefaultRrfK = 60.0;

inline int32_t total_key32(float x) {
    int32_t b;
    std::memcpy(&b, &x, 4);
    return b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1);
}
inline int64_t total_key64(double x) {
    int64_t b;
    std::memcpy(&b, &x, 8);
    return b ^ (int64_t)(((uint64_t)(b >> 63)) >> 1);
}
inline uint64_t fnv1a(std::string_view s) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (unsigned char c : s) {
        h ^= c;
        h *= 0x100000001b3ull;
    }
    return h;
}
inline std::string_view sv(const fsgpu_scored_doc& d) { return std::string_view(d.doc_id, d.doc_id_len); }

struct Scratch {
    std::string_view doc;
    double rrf = 0.0;
    int64_t lexical_rank = -1, semantic_rank = -1;
    uint32_t semantic_index = 0;
    bool has_index = false;
    float lexical_score = 0.f, semantic_score = 0.f;
    bool in_both = false;
};

}  // namespace

extern "C" fsgpu_status fsgpu_rrf_fuse(const fsgpu_scored_doc* lexical, uint32_t n_lexical,
                                        const fsgpu_scored_doc* semantic, uint32_t n_semantic, double k,
                                        double lexical_weight, double semantic_weight, int32_t tiebreak, uint32_t limit,
                                        uint32_t offset, fsgpu_fused_hit* out, uint32_t* out_count) {
    if (!out_count || (n_lexical && !lexical) || (n_semantic && !semantic) || (limit && !out)) return FSGPU_ERR_NULL_ARGUMENT;
    *out_count = 0;
    if (!(std::isfinite(k) && k >= 0.0)) k = kDefaultRrfK;
    if (!(std::isfinite(lexical_weight) && lexical_weight > 0.0)) lexical_weight = 1.0;
    if (!(std::isfinite(semantic_weight) && semantic_weight > 0.0)) semantic_weight = 1.0;
    std::unordered_map<std::string_view, size_t> slot;
    std::vector<Scratch> hits;
    hits.reserve(((size_t)n_lexical + n_semantic) * 3 / 4 + 1);
    for (uint32_t rank = 0; rank < n_lexical; ++rank) {
        const double c = (1.0 / (k + (double)rank + 1.0)) * lexical_weight;
        auto it = slot.find(sv(lexical[rank]));
        if (it == slot.end()) {
            Scratch s;
            s.doc = sv(lexical[rank]);
            s.rrf = c;
            s.lexical_rank = rank;
            s.lexical_score = lexical[rank].score;
            slot.emplace(s.doc, hits.size());
            hits.push_back(s);
        } else {
            Scratch& h = hits[it->second];
            if (h.lexical_rank >= 0) continue;  // first (best) occurrence wins
            h.rrf += c;
            h.lexical_rank = rank;
            h.lexical_score = lexical[rank].score;
            if (h.semantic_rank >= 0) h.in_both = true;
        }
    }
    for (uint32_t rank = 0; rank < n_semantic; ++rank) {
        const double c = (1.0 / (k + (double)rank + 1.0)) * semantic_weight;
        auto it = slot.find(sv(semantic[rank]));
        if (it == slot.end()) {
            Scratch s;
            s.doc = sv(semantic[rank]);
            s.rrf = c;
            s.semantic_rank = rank;
            s.semantic_score = semantic[rank].score;
            s.semantic_index = semantic[rank].index;
            s.has_index = true;
            slot.emplace(s.doc, hits.size());
            hits.push_back(s);
        } else {
            Scratch& h = hits[it->second];
            if (h.semantic_rank >= 0) continue;
            h.rrf += c;
            h.semantic_rank = rank;
            h.semantic_score = semantic[rank].score;
            h.semantic_index = semantic[rank].index;
            h.has_index = true;
            if (h.lexical_rank >= 0) h.in_both = true;
        }
    }
    const size_t window = (size_t)limit + offset;
    if (window == 0) return FSGPU_OK;
    const bool hash_tiebreak = tiebreak == FSGPU_RRF_TIEBREAK_HASH;
    auto before = [&](const Scratch& a, const Scratch& b) {
        const int64_t ka = total_key64(a.rrf), kb = total_key64(b.rrf);
        if (ka != kb) return ka > kb;
        if (a.in_both != b.in_both) return a.in_both;
        if (hash_tiebreak) {
            const uint64_t ha = fnv1a(a.doc), hb = fnv1a(b.doc);
            if (ha != hb) return ha < hb;
        } else {
            const float la = a.lexical_rank >= 0 ? a.lexical_score : -std::numeric_limits<float>::infinity();
            const float lb = b.lexical_rank >= 0 ? b.lexical_score : -std::numeric_limits<float>::infinity();
            const int32_t x = total_key32(la), y = total_key32(lb);
            if (x != y) return x > y;
        }
        return a.doc < b.doc;
    };
    if (window < hits.size()) {
        std::nth_element(hits.begin(), hits.begin() + (window - 1), hits.end(), before);
        hits.resize(window);
    }
    std::sort(hits.begin(), hits.end(), before);
    uint32_t n = 0;
    for (size_t i = offset; i < hits.size() && n < limit; ++i, ++n) {
        const Scratch& h = hits[i];
        fsgpu_fused_hit& o = out[n];
        o.doc_id = h.doc.data();
        o.doc_id_len = (uint32_t)h.doc.size();
        o.rrf_score = h.rrf;
        o.lexical_rank = h.lexical_rank;
        o.semantic_rank = h.semantic_rank;
        o.semantic_index = h.has_index ? h.semantic_index : 0xffffffffu;
        o.lexical_score = h.lexical_score;
        o.semantic_score = h.semantic_score;
        o.in_both_sources = h.in_both ? 1 : 0;
    }
    *out_count = n;
    return FSGPU_OK;
}

extern "C" fsgpu_status fsgpu_blend_two_tier(const fsgpu_scored_doc* fast, uint32_t n_fast,
                                              const fsgpu_scored_doc* quality, uint32_t n_quality, float blend_factor,
                                              fsgpu_scored_doc* out, uint32_t* out_count) {
    if (!out_count || (n_fast && !fast) || (n_quality && !quality) || ((n_fast || n_quality) && !out))
        return FSGPU_ERR_NULL_ARGUMENT;
    *out_count = 0;
    const float alpha = std::isfinite(blend_factor) ? std::min(std::max(blend_factor, 0.0f), 1.0f) : 0.7f;
    struct Bounds {
        float min = std::numeric_limits<float>::infinity(), range = 0.f;
        bool saw = false;
    };
    auto bounds = [](const fsgpu_scored_doc* hits, uint32_t n) {
        Bounds b;
        float mx = -std::numeric_limits<float>::infinity();
        for (uint32_t i = 0; i < n; ++i)
            if (std::isfinite(hits[i].score)) {
                b.min = std::min(b.min, hits[i].score);
                mx = std::max(mx, hits[i].score);
                b.saw = true;
            }
        b.range = mx - b.min;
        return b;
    };
    auto apply = [](const Bounds& b, float s) {
        if (!b.saw || !std::isfinite(s)) return 0.0f;
        const float v = b.range > 1.1920929e-7f ? (s - b.min) / b.range : 1.0f;
        return std::min(std::max(v, 0.0f), 1.0f);
    };
    const Bounds fb = bounds(fast, n_fast), qb = bounds(quality, n_quality);
    struct Pair {
        std::string_view doc;
        float fast = 0.f, quality = 0.f;
        bool has_fast = false, has_quality = false;
        uint32_t index = 0;
    };
    std::unordered_map<std::string_view, size_t> slot;
    std::vector<Pair> merged;
    merged.reserve((size_t)std::max(n_fast, n_quality) * 13 / 10 + 1);
    for (uint32_t i = 0; i < n_fast; ++i) {
        auto it = slot.find(sv(fast[i]));
        if (it == slot.end()) {
            Pair p;
            p.doc = sv(fast[i]);
            p.index = fast[i].index;
            it = slot.emplace(p.doc, merged.size()).first;
            merged.push_back(p);
        }
        Pair& p = merged[it->second];
        if (!p.has_fast) {  // best-first input: keep the first (best) score and its index
            p.fast = apply(fb, fast[i].score);
            p.has_fast = true;
            p.index = fast[i].index;
        }
    }
    for (uint32_t i = 0; i < n_quality; ++i) {
        auto it = slot.find(sv(quality[i]));
        if (it == slot.end()) {
            Pair p;
            p.doc = sv(quality[i]);
            p.index = quality[i].index;
            it = slot.emplace(p.doc, merged.size()).first;
            merged.push_back(p);
        }
        Pair& p = merged[it->second];
        if (!p.has_quality) {
            p.quality = apply(qb, quality[i].score);
            p.has_quality = true;
        }
    }
    std::vector<fsgpu_scored_doc> blended(merged.size());
    for (size_t i = 0; i < merged.size(); ++i) {
        const Pair& p = merged[i];
        float score;
        if (p.has_fast && p.has_quality) score = std::fmaf(alpha, p.quality, (1.0f - alpha) * p.fast);
        else if (p.has_fast) score = p.fast;
        else if (p.has_quality) score = p.quality;
        else score = 0.0f;
        if (!std::isfinite(score)) score = 0.0f;
        blended[i] = fsgpu_scored_doc{p.doc.data(), (uint32_t)p.doc.size(), score, p.index};
    }
    std::sort(blended.begin(), blended.end(), [](const fsgpu_scored_doc& a, const fsgpu_scored_doc& b) {
        const int32_t x = total_key32(a.score), y = total_key32(b.score);
        if (x != y) return x > y;
        return std::string_view(a.doc_id, a.doc_id_len) < std::string_view(b.doc_id, b.doc_id_len);
    });
    std::copy(blended.begin(), blended.end(), out);
    *out_count = (uint32_t)blended.size();
    return FSGPU_OK;
}

// blend_two_tier_aligned (blend.rs:213-294): the same normalisation, merge and order as fsgpu_blend_two_tier with the quality tier
// given as per-position optional scores of the fast hits themselves.
extern "C" fsgpu_status fsgpu_blend_two_tier_aligned(const fsgpu_scored_doc* fast, uint32_t n_fast, const float* quality_scores,
                                                      const uint8_t* quality_present, float blend_factor, fsgpu_scored_doc* out,
                                                      uint32_t* out_count) {
    if (!out_count || (n_fast && (!fast || !out || !quality_scores || !quality_present))) return FSGPU_ERR_NULL_ARGUMENT;
    *out_count = 0;
    const float alpha = std::isfinite(blend_factor) ? std::min(std::max(blend_factor, 0.0f), 1.0f) : 0.7f;
    struct Bounds {
        float min = std::numeric_limits<float>::infinity(), max = -std::numeric_limits<float>::infinity(), range = 0.f;
        bool saw = false;
        void add(float v) {
            if (std::isfinite(v)) {
                min = std::min(min, v);
                max = std::max(max, v);
                saw = true;
            }
        }
        float apply(float s) const {
            if (!saw || !std::isfinite(s)) return 0.0f;
            const float v = range > 1.1920929e-7f ? (s - min) / range : 1.0f;
            return std::min(std::max(v, 0.0f), 1.0f);
        }
    } fb, qb;
    for (uint32_t i = 0; i < n_fast; ++i) {
        fb.add(fast[i].score);
        if (quality_present[i]) qb.add(quality_scores[i]);
    }
    fb.range = fb.max - fb.min;
    qb.range = qb.max - qb.min;
    struct Pair {
        std::string_view doc;
        float fast = 0.f, quality = 0.f;
        bool has_fast = false, has_quality = false;
        uint32_t index = 0;
    };
    std::unordered_map<std::string_view, size_t> slot;
    std::vector<Pair> merged;
    merged.reserve((size_t)n_fast * 13 / 10 + 1);
    for (uint32_t i = 0; i < n_fast; ++i) {
        auto it = slot.find(sv(fast[i]));
        if (it == slot.end()) {
            Pair p;
            p.doc = sv(fast[i]);
            p.index = fast[i].index;
            it = slot.emplace(p.doc, merged.size()).first;
            merged.push_back(p);
        }
        Pair& p = merged[it->second];
        if (!p.has_fast) {   // best-first input: the first (best) score and its index
            p.fast = fb.apply(fast[i].score);
            p.has_fast = true;
            p.index = fast[i].index;
        }
        if (quality_present[i] && !p.has_quality) {
            p.quality = qb.apply(quality_scores[i]);
            p.has_quality = true;
        }
    }
    std::vector<fsgpu_scored_doc> blended(merged.size());
    for (size_t i = 0; i < merged.size(); ++i) {
        const Pair& p = merged[i];
        float score = p.has_quality ? std::fmaf(alpha, p.quality, (1.0f - alpha) * p.fast) : p.fast;
        if (!std::isfinite(score)) score = 0.0f;
        blended[i] = fsgpu_scored_doc{p.doc.data(), (uint32_t)p.doc.size(), score, p.index};
    }
    std::sort(blended.begin(), blended.end(), [](const fsgpu_scored_doc& a, const fsgpu_scored_doc& b) {
        const int32_t x = total_key32(a.score), y = total_key32(b.score);
        if (x != y) return x > y;
        return std::string_view(a.doc_id, a.doc_id_len) < std::string_view(b.doc_id, b.doc_id_len);
    });
    std::copy(blended.begin(), blended.end(), out);
    *out_count = (uint32_t)blended.size();
    return FSGPU_OK;
}
