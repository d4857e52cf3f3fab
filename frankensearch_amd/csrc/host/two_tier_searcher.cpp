// two_tier_searcher.cpp — see two_tier_searcher.hpp.  Calls only functions declared in include/fsgpu.h.
#include "two_tier_searcher.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <future>
#include <unordered_map>

namespace fshost {

namespace {

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

std::vector<fsgpu_scored_doc> view(const std::vector<Hit>& hits) {
    std::vector<fsgpu_scored_doc> v(hits.size());
    for (size_t i = 0; i < hits.size(); ++i)
        v[i] = fsgpu_scored_doc{hits[i].doc_id.data(), (uint32_t)hits[i].doc_id.size(), hits[i].score, hits[i].index};
    return v;
}

fsgpu_status copy_out(const std::vector<fsgpu_fused_hit>& fused, uint32_t n, std::vector<fshost_hit>* out, std::string* detail) {
    out->resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const fsgpu_fused_hit& f = fused[i];
        fshost_hit& h = (*out)[i];
        if (f.doc_id_len > FSHOST_DOC_ID_MAX) {
            *detail = "doc id longer than FSHOST_DOC_ID_MAX";
            return FSGPU_ERR_INVALID_CONFIG;
        }
        std::memcpy(h.doc_id, f.doc_id, f.doc_id_len);
        h.doc_id[f.doc_id_len] = 0;
        h.rrf_score = f.rrf_score;
        h.lexical_rank = f.lexical_rank;
        h.semantic_rank = f.semantic_rank;
        h.semantic_index = f.semantic_index;
        h.lexical_score = f.lexical_score;
        h.semantic_score = f.semantic_score;
        h.in_both_sources = f.in_both_sources;
    }
    return FSGPU_OK;
}

}  // namespace

fsgpu_status Tier::search_rows(const float* query, uint32_t len, uint32_t fetch, uint32_t int8_multiplier, uint32_t* rows, float* scores,
                               uint32_t* count) const {
    if (index)
        return int8_multiplier ? fsgpu_search_topk_int8_two_pass(index, query, len, fetch, int8_multiplier, rows, scores, count)
                               : fsgpu_search_topk(index, query, 1, len, fetch, nullptr, rows, scores, count);
    fsgpu_sharded_request rq{query, 1, len, fetch, int8_multiplier ? FSGPU_SHARDED_INT8_TWO_PASS : FSGPU_SHARDED_EXACT, int8_multiplier, nullptr};
    return fsgpu_sharded_search(sharded, &rq, rows, scores, count, nullptr);
}

fsgpu_status Tier::search_hits(const float* query, uint32_t len, uint32_t fetch, uint32_t* rows, float* scores, uint32_t* count) const {
    return index ? fsgpu_search_hits(index, query, len, fetch, rows, scores, count)
                 : fsgpu_sharded_search_hits(sharded, query, len, fetch, rows, scores, count);
}

fsgpu_status Tier::doc_id(uint32_t row, const char** ptr, uint32_t* len) const {
    return index ? fsgpu_index_doc_id(index, row, ptr, len) : fsgpu_sharded_doc_id(sharded, row, ptr, len);
}

SyncTwoTierSearcher::SyncTwoTierSearcher(fsgpu_index* fast, fsgpu_index* quality, fsgpu_m2v* fast_embedder,
                                         fsgpu_bert* quality_embedder, const fshost_two_tier_config& cfg)
    : m2v_(fast_embedder), bert_(quality_embedder), cfg_(cfg) {
    fast_.index = fast;
    quality_.index = quality;
    init();
}

SyncTwoTierSearcher::SyncTwoTierSearcher(fsgpu_sharded* fast, fsgpu_sharded* quality, fsgpu_m2v* fast_embedder,
                                         fsgpu_bert* quality_embedder, const fshost_two_tier_config& cfg)
    : m2v_(fast_embedder), bert_(quality_embedder), cfg_(cfg) {
    fast_.sharded = fast;
    quality_.sharded = quality;
    init();
}

void SyncTwoTierSearcher::init() {
    fast_dim_ = fast_.dimension();
    quality_dim_ = quality_.dimension();
    quality_dim_vec_ = fsgpu_bert_dimension(bert_);   // a vector has its EMBEDDER's dimension (a mismatch is the index's to report)
    // opt-in: the quality tier's exact search is phase 1's longest leg (one HBM pass over the f16 slab); with the int8 latency
    // path a lone caller's query goes through the int8 filter + exact re-score instead — the same hits from half the bytes.
    // It is a setting of the CALLER's handle (and costs it an int8 copy of the slab): switched off again in the destructor.
    // (A sharded quality tier takes the same switch on every shard: fsgpu_sharded_set_int8_latency.)
    if (cfg_.quality_int8_latency && cfg_.quality_pool == FSHOST_POOL_RETRIEVED) {
        init_status_ = quality_.index ? fsgpu_index_set_int8_latency(quality_.index, FSGPU_INT8_LATENCY_BUILD_NOW) : fsgpu_sharded_set_int8_latency(quality_.sharded, 1);
        if (init_status_ != FSGPU_OK) init_detail_ = fsgpu_last_error();
    }
    if (init_status_ == FSGPU_OK && cfg_.quality_pool == FSHOST_POOL_RESCORED) {
        // two_tier.rs:750-866, once per pair
        init_status_ = fast_.index ? fsgpu_alignment_create(fast_.index, quality_.index, &alignment_)
                                   : fsgpu_sharded_alignment_create(fast_.sharded, quality_.sharded, &alignment_);
        if (init_status_ != FSGPU_OK) init_detail_ = fsgpu_last_error();
    }
}

SyncTwoTierSearcher::~SyncTwoTierSearcher() {
    stop_engine();   // the engine's threads use the handles below
    if (alignment_) fsgpu_alignment_destroy(alignment_);
    if (cfg_.quality_int8_latency && cfg_.quality_pool == FSHOST_POOL_RETRIEVED) {
        if (quality_.index) (void)fsgpu_index_set_int8_latency(quality_.index, 0);
        else (void)fsgpu_sharded_set_int8_latency(quality_.sharded, 0);
    }
}

// VectorIndex::search_top_k -> Vec<VectorHit> with doc ids resolved (search.rs:192-206, 1503-1558).
fsgpu_status SyncTwoTierSearcher::tier_hits(const Tier& tier, const std::vector<float>& vec, uint32_t fetch,
                                            uint32_t int8_multiplier, std::vector<Hit>* hits, std::string* detail) const {
    std::vector<uint32_t> rows(fetch);
    std::vector<float> scores(fetch);
    uint32_t count = 0;
    // an index with a doc-id table (doc_id_mode 0) goes through search_hits = search_top_k + scan_wal + resolve_hits
    // (resident WAL entries, shadowing, post-top-k doc-id dedup, search.rs:1493-1558), as the int8 branch does inside the
    // library; a raw slab (synthetic doc ids) has neither a WAL nor duplicate ids, so the row-level search is the same thing
    fsgpu_status st = int8_multiplier || cfg_.doc_id_mode != 0
                          ? tier.search_rows(vec.data(), (uint32_t)vec.size(), fetch, int8_multiplier, rows.data(), scores.data(), &count)
                          : tier.search_hits(vec.data(), (uint32_t)vec.size(), fetch, rows.data(), scores.data(), &count);
    if (st != FSGPU_OK) {
        *detail = fsgpu_last_error();
        return st;
    }
    return hits_from_rows(tier, rows.data(), scores.data(), count, hits, detail);
}

// Vec<VectorHit> of one tier's row-level answer: doc ids resolved (search.rs:1503-1558) or synthesised (doc_id_mode 1)
fsgpu_status SyncTwoTierSearcher::hits_from_rows(const Tier& tier, const uint32_t* rows, const float* scores, uint32_t count,
                                                 std::vector<Hit>* hits, std::string* detail) const {
    fsgpu_status st = FSGPU_OK;
    hits->clear();
    hits->reserve(count);
    char buf[32];
    for (uint32_t i = 0; i < count; ++i) {
        Hit h;
        h.score = scores[i];
        h.index = rows[i];
        if (cfg_.doc_id_mode == 1) {
            const int n = std::snprintf(buf, sizeof buf, "doc-%08u", rows[i]);
            h.doc_id.assign(buf, (size_t)n);
        } else {
            const char* p = nullptr;
            uint32_t len = 0;
            st = tier.doc_id(rows[i], &p, &len);
            if (st != FSGPU_OK) {
                *detail = fsgpu_last_error();
                return st;
            }
            h.doc_id.assign(p, len);
        }
        hits->push_back(std::move(h));
    }
    return FSGPU_OK;
}

fsgpu_status SyncTwoTierSearcher::search(const uint32_t* fast_ids, uint32_t n_fast, const int32_t* quality_ids,
                                         uint32_t n_quality, uint32_t k, const fsgpu_scored_doc* lexical, uint32_t n_lexical,
                                         Outcome* out, std::string* detail) const {
    if (init_status_ != FSGPU_OK) {
        *detail = init_detail_;
        return init_status_;
    }
    // dynamic batching (fshost_two_tier_set_batching): this caller's query rides the many-queries engine with whoever else is calling
    if (batching_.load(std::memory_order_relaxed))
        return engine(0)->search_one(fast_ids, n_fast, quality_ids, n_quality, k, lexical, n_lexical, out, detail);
    return search_unbatched(fast_ids, n_fast, quality_ids, n_quality, k, lexical, n_lexical, out, detail);
}

fsgpu_status SyncTwoTierSearcher::search_unbatched(const uint32_t* fast_ids, uint32_t n_fast, const int32_t* quality_ids,
                                                   uint32_t n_quality, uint32_t k, const fsgpu_scored_doc* lexical, uint32_t n_lexical,
                                                   Outcome* out, std::string* detail) const {
    using clock = std::chrono::steady_clock;
    if (init_status_ != FSGPU_OK) {
        *detail = init_detail_;
        return init_status_;
    }
    const uint32_t mult = std::max<uint32_t>(cfg_.candidate_multiplier, 1);
    // candidate_count (rrf.rs:113-115): limit.saturating_mul(multiplier).max(limit)
    const uint64_t wide_fetch = (uint64_t)k * mult;
    const uint32_t fetch = std::max<uint32_t>(wide_fetch > 0xffffffffull ? 0xffffffffu : (uint32_t)wide_fetch, k);
    fshost_metrics& m = out->metrics;
    out->refinement_failed = false;
    out->skip_reason.clear();
    const auto t0 = clock::now();
    // quality embedding: needed by phase 1 only, optionally computed while phase 0 runs
    // (a vector has its EMBEDDER's dimension; an index of another dimension answers DimensionMismatch — search.rs:1602-1610 — which
    // fails phase 0 on the fast tier and is a RefinementFailed outcome on the quality tier)
    std::vector<float> quality_vec(fsgpu_bert_dimension(bert_));
    const uint32_t q_off[2] = {0, n_quality};
    std::string quality_err;
    auto embed_quality = [&]() -> fsgpu_status {
        const fsgpu_status s = fsgpu_bert_embed(bert_, quality_ids, q_off, 1, quality_vec.data());
        if (s != FSGPU_OK) quality_err = fsgpu_last_error();  // thread-local: read on the thread that made the call
        return s;
    };
    // prefetch_quality_embed == 2: the helper goes on to the quality tier's search as well (it depends on nothing phase 0
    // produces): both tiers' scans then share the GPU, phase 0 is delivered a little later and phase 1 much earlier
    std::vector<Hit> quality_hits;  // the `Retrieved` pool (sync_searcher.rs:810-813)
    std::string quality_search_err;
    bool quality_searched = false, quality_search_failed = false;
    auto embed_and_search_quality = [&]() -> fsgpu_status {
        fsgpu_status s = embed_quality();
        if (s != FSGPU_OK) return s;   // the embedding's failure is the search's (embed_sync's error propagates)
        s = tier_hits(quality_, quality_vec, fetch, 0, &quality_hits, &quality_search_err);
        quality_searched = s == FSGPU_OK;
        quality_search_failed = s != FSGPU_OK;   // the pool's failure is a RefinementFailed outcome, decided in phase 1
        return FSGPU_OK;
    };
    std::future<fsgpu_status> quality_future;  // declared after what the task touches: joined first on every return path
    const bool rescored = cfg_.quality_pool == FSHOST_POOL_RESCORED;   // (its quality scores need phase 0's hits: only the embedding can run ahead)
    if (cfg_.prefetch_quality_embed >= 2 && !rescored) quality_future = std::async(std::launch::async, embed_and_search_quality);
    else if (cfg_.prefetch_quality_embed) quality_future = std::async(std::launch::async, embed_quality);
    // ---- phase 0 / Initial ----
    std::vector<float> fast_vec(fsgpu_m2v_dimension(m2v_));
    const uint32_t fast_off[2] = {0, n_fast};
    fsgpu_status st = fsgpu_m2v_embed(m2v_, fast_ids, fast_off, 1, fast_vec.data());
    if (st != FSGPU_OK) {
        *detail = fsgpu_last_error();
        return st;
    }
    m.fast_embed_ms = ms_since(t0);
    const auto t1 = clock::now();
    std::vector<Hit> fast_hits;
    st = tier_hits(fast_, fast_vec, fetch, cfg_.fast_tier_int8_multiplier, &fast_hits, detail);
    if (st != FSGPU_OK) return st;
    m.fast_search_ms = ms_since(t1);
    st = fuse_initial(fast_hits, k, lexical, n_lexical, &out->initial, detail);
    if (st != FSGPU_OK) return st;
    m.phase1_total_ms = ms_since(t0);
    // ---- phase 1 / Refined ----
    const auto t3 = clock::now();
    // The quality pool failing — scoring, gathering, the quality tier's search — does not fail the search: the reference records
    // SearchPhase::RefinementFailed / skip_reason and returns the phase-0 results as final_results (sync_searcher.rs:820-839)
    auto refinement_failed = [&](Outcome* o, clock::time_point started) -> fsgpu_status {
        o->refinement_failed = true;
        o->skip_reason = quality_search_failed ? quality_search_err : std::string(fsgpu_last_error());
        o->final_results = o->initial;
        o->metrics.phase2_total_ms = ms_since(started);
        return FSGPU_OK;
    };
    st = quality_future.valid() ? quality_future.get() : embed_quality();
    if (st != FSGPU_OK) {
        *detail = quality_err;
        return st;
    }
    m.quality_embed_ms = ms_since(t3);
    const auto t4 = clock::now();
    if (rescored) {
        // SyncQualityPool::RescoredFastPool (sync_searcher.rs:814-818): quality_scores_for_hits over the fast pool, then the
        // aligned blend (:862-866)
        bool failed = false;
        st = fuse_final_rescored(fast_hits, quality_vec.data(), k, lexical, n_lexical, &out->final_results, &failed, detail);
        if (failed) return refinement_failed(out, t3);
        if (st != FSGPU_OK) return st;
        m.quality_search_ms = ms_since(t4);
        m.phase2_total_ms = ms_since(t3);
        return FSGPU_OK;
    }
    if (quality_search_failed) return refinement_failed(out, t3);
    if (!quality_searched) {
        st = tier_hits(quality_, quality_vec, fetch, 0, &quality_hits, detail);
        if (st != FSGPU_OK) return refinement_failed(out, t3);
    }
    m.quality_search_ms = ms_since(t4);
    const auto t5 = clock::now();
    st = fuse_final_retrieved(fast_hits, quality_hits, k, lexical, n_lexical, &out->final_results, detail);
    if (st != FSGPU_OK) return st;
    m.blend_ms = ms_since(t5);
    m.phase2_total_ms = ms_since(t3);
    return FSGPU_OK;
}

// Phase 1's fusion for SyncQualityPool::RescoredFastPool (sync_searcher.rs:814-818, 862-866): quality_scores_for_hits over the fast
// pool (a gather on the quality tier), blend_two_tier_aligned, RRF.  *failed: the quality scores could not be produced — a
// RefinementFailed outcome, not an error (sync_searcher.rs:820-839).
fsgpu_status SyncTwoTierSearcher::fuse_final_rescored(const std::vector<Hit>& fast_hits, const float* quality_vec, uint32_t k,
                                                      const fsgpu_scored_doc* lexical, uint32_t n_lexical, std::vector<fshost_hit>* final_results,
                                                      bool* failed, std::string* detail) const {
    *failed = false;
    const std::vector<fsgpu_scored_doc> fast_view = view(fast_hits);
    std::vector<float> qscores(fast_view.size() + 1);
    std::vector<uint8_t> qpresent(fast_view.size() + 1);
    const fsgpu_status st = fast_.index ? fsgpu_quality_scores_for_hits(fast_.index, quality_.index, alignment_, quality_vec, quality_dim_vec_,
                                                                        fast_view.data(), (uint32_t)fast_view.size(), qscores.data(), qpresent.data())
                                        : fsgpu_sharded_quality_scores_for_hits(fast_.sharded, quality_.sharded, alignment_, quality_vec, quality_dim_vec_,
                                                                                fast_view.data(), (uint32_t)fast_view.size(), qscores.data(), qpresent.data());
    if (st != FSGPU_OK) {
        *failed = true;
        return st;
    }
    return fuse_final_rescored_scores(fast_hits, qscores.data(), qpresent.data(), k, lexical, n_lexical, final_results, detail);
}

// ... the second half: blend_two_tier_aligned of the fast pool with its quality scores, RRF (sync_searcher.rs:862-918)
fsgpu_status SyncTwoTierSearcher::fuse_final_rescored_scores(const std::vector<Hit>& fast_hits, const float* qscores, const uint8_t* qpresent, uint32_t k,
                                                             const fsgpu_scored_doc* lexical, uint32_t n_lexical, std::vector<fshost_hit>* final_results,
                                                             std::string* detail) const {
    const std::vector<fsgpu_scored_doc> fast_view = view(fast_hits);
    fsgpu_status st = FSGPU_OK;
    std::vector<fsgpu_scored_doc> blended(fast_view.size() + 1);
    uint32_t nb = 0, n = 0;
    st = fsgpu_blend_two_tier_aligned(fast_view.data(), (uint32_t)fast_view.size(), qscores, qpresent, cfg_.quality_weight,
                                      blended.data(), &nb);
    if (st != FSGPU_OK) {
        *detail = fsgpu_last_error();
        return st;
    }
    std::vector<fsgpu_fused_hit> fused(k ? k : 1);
    st = fsgpu_rrf_fuse(lexical, n_lexical, blended.data(), nb, cfg_.rrf_k, 1.0, 1.0, FSGPU_RRF_TIEBREAK_LEXICAL_THEN_ID, k, 0,
                        fused.data(), &n);
    if (st != FSGPU_OK) {
        *detail = fsgpu_last_error();
        return st;
    }
    return copy_out(fused, n, final_results, detail);
}

// Phase 0's fusion (sync_searcher.rs:700-760): RRF of the lexical list with the fast tier's hits, k results.
fsgpu_status SyncTwoTierSearcher::fuse_initial(const std::vector<Hit>& fast_hits, uint32_t k, const fsgpu_scored_doc* lexical,
                                               uint32_t n_lexical, std::vector<fshost_hit>* initial, std::string* detail) const {
    const std::vector<fsgpu_scored_doc> fast_view = view(fast_hits);
    std::vector<fsgpu_fused_hit> fused(k ? k : 1);
    uint32_t n = 0;
    const fsgpu_status st = fsgpu_rrf_fuse(lexical, n_lexical, fast_view.data(), (uint32_t)fast_view.size(), cfg_.rrf_k, 1.0, 1.0,
                                           FSGPU_RRF_TIEBREAK_LEXICAL_THEN_ID, k, 0, fused.data(), &n);
    if (st != FSGPU_OK) {
        *detail = fsgpu_last_error();
        return st;
    }
    return copy_out(fused, n, initial, detail);
}

// Phase 1's fusion for SyncQualityPool::Retrieved (sync_searcher.rs:840-943): blend_two_tier of the two tiers' hits, the blended
// hits carrying the fast-tier row of their doc, RRF with the lexical list again.
fsgpu_status SyncTwoTierSearcher::fuse_final_retrieved(const std::vector<Hit>& fast_hits, const std::vector<Hit>& quality_hits, uint32_t k,
                                                       const fsgpu_scored_doc* lexical, uint32_t n_lexical, std::vector<fshost_hit>* final_results,
                                                       std::string* detail) const {
    const std::vector<fsgpu_scored_doc> fast_view = view(fast_hits);
    const std::vector<fsgpu_scored_doc> quality_view = view(quality_hits);
    std::vector<fsgpu_scored_doc> blended(fast_view.size() + quality_view.size() + 1);
    uint32_t nb = 0, n = 0;
    fsgpu_status st = fsgpu_blend_two_tier(fast_view.data(), (uint32_t)fast_view.size(), quality_view.data(), (uint32_t)quality_view.size(),
                                           cfg_.quality_weight, blended.data(), &nb);
    if (st != FSGPU_OK) {
        *detail = fsgpu_last_error();
        return st;
    }
    // blended hits carry the fast-tier row of their doc (sync_searcher.rs:880-891)
    std::unordered_map<std::string, uint32_t> fast_index_of;
    fast_index_of.reserve(fast_hits.size() * 2);
    for (const Hit& h : fast_hits) fast_index_of.emplace(h.doc_id, h.index);
    for (uint32_t i = 0; i < nb; ++i) {
        auto it = fast_index_of.find(std::string(blended[i].doc_id, blended[i].doc_id_len));
        blended[i].index = it == fast_index_of.end() ? 0xffffffffu : it->second;
    }
    std::vector<fsgpu_fused_hit> fused(k ? k : 1);
    st = fsgpu_rrf_fuse(lexical, n_lexical, blended.data(), nb, cfg_.rrf_k, 1.0, 1.0, FSGPU_RRF_TIEBREAK_LEXICAL_THEN_ID, k, 0,
                        fused.data(), &n);
    if (st != FSGPU_OK) {
        *detail = fsgpu_last_error();
        return st;
    }
    return copy_out(fused, n, final_results, detail);
}

}  // namespace fshost
