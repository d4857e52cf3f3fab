// fshost_api.cpp — extern "C" surface of libfshost.so (include/fshost.h).
#include <cstring>
#include <exception>
#include <string>

#include "two_tier_searcher.hpp"

namespace fshost {
fsgpu_status run_load(const SyncTwoTierSearcher& searcher, const fshost_load_config& cfg, fshost_load_result* res);
fsgpu_status run_load_many(const SyncTwoTierSearcher& searcher, const fshost_load_config& cfg, uint32_t chunk, fshost_many_result* res);
fsgpu_status embed_search_stream(fsgpu_bert* const* encoders, uint32_t n_encoders, fsgpu_index* index, fsgpu_sharded* sharded, const int32_t* ids,
                                 const uint32_t* offsets, uint32_t batch, uint32_t n_batches, uint32_t group, uint32_t k, bool overlap,
                                 bool host_handoff, uint32_t* out_rows, float* out_scores, uint32_t* out_counts, fshost_stream_result* result);
}

struct fshost_two_tier {
    fshost::SyncTwoTierSearcher impl;
    fshost_two_tier(fsgpu_index* f, fsgpu_index* q, fsgpu_m2v* m, fsgpu_bert* b, const fshost_two_tier_config& c) : impl(f, q, m, b, c) {}
    fshost_two_tier(fsgpu_sharded* f, fsgpu_sharded* q, fsgpu_m2v* m, fsgpu_bert* b, const fshost_two_tier_config& c) : impl(f, q, m, b, c) {}
};

namespace {
template <class IndexHandle>
fsgpu_status create_searcher(IndexHandle* fast_index, IndexHandle* quality_index, fsgpu_m2v* fast_embedder, fsgpu_bert* quality_embedder,
                             const fshost_two_tier_config* config, fshost_two_tier** out) {
    if (!fast_index || !quality_index || !fast_embedder || !quality_embedder || !config || !out) return FSGPU_ERR_NULL_ARGUMENT;
    try {
        auto* s = new fshost_two_tier(fast_index, quality_index, fast_embedder, quality_embedder, *config);
        if (s->impl.init_status() != FSGPU_OK) {   // the alignment of a re-scored pair / the int8 latency switch failed
            const fsgpu_status st = s->impl.init_status();
            delete s;
            return st;
        }
        *out = s;
    } catch (const std::exception&) {
        return FSGPU_ERR_DEVICE;
    }
    return FSGPU_OK;
}
}  // namespace

extern "C" {

fsgpu_status fshost_two_tier_create(fsgpu_index* fast_index, fsgpu_index* quality_index, fsgpu_m2v* fast_embedder,
                                    fsgpu_bert* quality_embedder, const fshost_two_tier_config* config,
                                    fshost_two_tier** out) {
    return create_searcher(fast_index, quality_index, fast_embedder, quality_embedder, config, out);
}

fsgpu_status fshost_two_tier_create_sharded(fsgpu_sharded* fast_index, fsgpu_sharded* quality_index, fsgpu_m2v* fast_embedder,
                                            fsgpu_bert* quality_embedder, const fshost_two_tier_config* config,
                                            fshost_two_tier** out) {
    return create_searcher(fast_index, quality_index, fast_embedder, quality_embedder, config, out);
}

void fshost_two_tier_destroy(fshost_two_tier* s) { delete s; }

fsgpu_status fshost_two_tier_search(fshost_two_tier* s, const uint32_t* fast_token_ids, uint32_t n_fast_ids,
                                    const int32_t* quality_token_ids, uint32_t n_quality_ids, uint32_t k,
                                    const fsgpu_scored_doc* lexical, uint32_t n_lexical, fshost_hit* initial_out,
                                    uint32_t* n_initial, fshost_hit* final_out, uint32_t* n_final, fshost_metrics* metrics) {
    if (!s || !n_initial || !n_final || (k && (!initial_out || !final_out))) return FSGPU_ERR_NULL_ARGUMENT;
    try {
        fshost::Outcome out;
        std::string detail;
        const fsgpu_status st = s->impl.search(fast_token_ids, n_fast_ids, quality_token_ids, n_quality_ids, k, lexical,
                                               n_lexical, &out, &detail);
        if (st != FSGPU_OK) return st;
        *n_initial = (uint32_t)out.initial.size();
        *n_final = (uint32_t)out.final_results.size();
        if (!out.initial.empty()) std::memcpy(initial_out, out.initial.data(), out.initial.size() * sizeof(fshost_hit));
        if (!out.final_results.empty())
            std::memcpy(final_out, out.final_results.data(), out.final_results.size() * sizeof(fshost_hit));
        if (metrics) {
            *metrics = out.metrics;
            metrics->refinement_failed = out.refinement_failed ? 1 : 0;
        }
        return FSGPU_OK;
    } catch (const std::exception&) {
        return FSGPU_ERR_DEVICE;
    }
}

fsgpu_status fshost_two_tier_search_many(fshost_two_tier* s, const uint32_t* fast_token_ids, const uint32_t* fast_offsets,
                                         const int32_t* quality_token_ids, const uint32_t* quality_offsets, uint32_t nq, uint32_t k,
                                         const fsgpu_scored_doc* lexical, const uint32_t* lexical_offsets, uint32_t chunk,
                                         uint32_t fusion_threads, fshost_hit* initial_out, uint32_t* n_initial, fshost_hit* final_out,
                                         uint32_t* n_final, uint8_t* refinement_failed_out, float* fast_vectors_out,
                                         float* quality_vectors_out, fshost_many_result* result) {
    if (!s || !result) return FSGPU_ERR_NULL_ARGUMENT;
    if (nq && (!fast_offsets || !quality_offsets || !n_initial || !n_final || (k && (!initial_out || !final_out)))) return FSGPU_ERR_NULL_ARGUMENT;
    if ((lexical == nullptr) != (lexical_offsets == nullptr)) return FSGPU_ERR_NULL_ARGUMENT;
    try {
        fshost::SyncTwoTierSearcher::ManyArgs a;
        a.fast_ids = fast_token_ids;
        a.fast_offsets = fast_offsets;
        a.quality_ids = quality_token_ids;
        a.quality_offsets = quality_offsets;
        a.nq = nq;
        a.k = k;
        a.lexical = lexical;
        a.lexical_offsets = lexical_offsets;
        a.chunk = chunk;
        a.fusion_threads = fusion_threads;
        a.initial_out = initial_out;
        a.n_initial = n_initial;
        a.final_out = final_out;
        a.n_final = n_final;
        a.refinement_failed = refinement_failed_out;
        a.fast_vectors_out = fast_vectors_out;
        a.quality_vectors_out = quality_vectors_out;
        return s->impl.search_many(a, result);
    } catch (const std::exception&) {
        return FSGPU_ERR_DEVICE;
    }
}

fsgpu_status fshost_two_tier_set_batching(fshost_two_tier* s, uint32_t max_chunk, uint32_t max_wait_us) {
    if (!s) return FSGPU_ERR_NULL_ARGUMENT;
    try {
        return s->impl.set_batching(max_chunk, max_wait_us);
    } catch (const std::exception&) {
        return FSGPU_ERR_DEVICE;
    }
}

fsgpu_status fshost_two_tier_batching_stats(fshost_two_tier* s, uint64_t* chunks, uint64_t* requests) {
    if (!s || !chunks || !requests) return FSGPU_ERR_NULL_ARGUMENT;
    s->impl.batching_stats(chunks, requests);
    return FSGPU_OK;
}

fsgpu_status fshost_run_load(fshost_two_tier* s, const fshost_load_config* config, fshost_load_result* result) {
    if (!s || !config || !result) return FSGPU_ERR_NULL_ARGUMENT;
    try {
        return fshost::run_load(s->impl, *config, result);
    } catch (const std::exception&) {
        return FSGPU_ERR_DEVICE;
    }
}

fsgpu_status fshost_run_load_many(fshost_two_tier* s, const fshost_load_config* config, uint32_t chunk, fshost_many_result* result) {
    if (!s || !config || !result) return FSGPU_ERR_NULL_ARGUMENT;
    try {
        return fshost::run_load_many(s->impl, *config, chunk, result);
    } catch (const std::exception&) {
        return FSGPU_ERR_DEVICE;
    }
}

fsgpu_status fshost_embed_search_stream(fsgpu_bert* encoder, fsgpu_index* index, fsgpu_sharded* sharded, const int32_t* ids,
                                        const uint32_t* offsets, uint32_t batch, uint32_t n_batches, uint32_t group, uint32_t k,
                                        int32_t overlap, uint32_t* out_rows, float* out_scores, uint32_t* out_counts,
                                        fshost_stream_result* result) {
    if (!encoder || !ids || !offsets || !result || (index == nullptr) == (sharded == nullptr)) return FSGPU_ERR_NULL_ARGUMENT;
    if (batch == 0 || group == 0 || k == 0) return FSGPU_ERR_INVALID_CONFIG;
    try {
        return fshost::embed_search_stream(&encoder, 1, index, sharded, ids, offsets, batch, n_batches, group, k, (overlap & 1) != 0,
                                           (overlap & 2) != 0, out_rows, out_scores, out_counts, result);
    } catch (const std::exception&) {
        return FSGPU_ERR_DEVICE;
    }
}

fsgpu_status fshost_embed_search_stream_dp(fsgpu_bert* const* encoders, uint32_t n_encoders, fsgpu_sharded* sharded, const int32_t* ids,
                                           const uint32_t* offsets, uint32_t batch, uint32_t n_batches, uint32_t group, uint32_t k,
                                           int32_t overlap, uint32_t* out_rows, float* out_scores, uint32_t* out_counts,
                                           fshost_stream_result* result) {
    if (!encoders || n_encoders == 0 || !sharded || !ids || !offsets || !result) return FSGPU_ERR_NULL_ARGUMENT;
    for (uint32_t e = 0; e < n_encoders; ++e)
        if (!encoders[e]) return FSGPU_ERR_NULL_ARGUMENT;
    if (batch == 0 || group == 0 || k == 0) return FSGPU_ERR_INVALID_CONFIG;
    try {
        return fshost::embed_search_stream(encoders, n_encoders, nullptr, sharded, ids, offsets, batch, n_batches, group, k, (overlap & 1) != 0,
                                           false, out_rows, out_scores, out_counts, result);
    } catch (const std::exception&) {
        return FSGPU_ERR_DEVICE;
    }
}

}  // extern "C"
