/* fshost.h — host-side mirror of the reference's two-phase searcher, written in C++ over the fsgpu C ABI.
 *
 * The reference's host is Rust and no Rust toolchain exists in this environment, so the code that would sit above
 * libfsgpu.so in a real deployment (crates/frankensearch-fusion/src/sync_searcher.rs:616-943, the SyncTwoTierSearcher
 * flow; fsfs shape crates/frankensearch-fsfs/src/runtime.rs:8185-8355) is restated here in C++, calling ONLY the
 * functions declared in fsgpu.h — exactly what the Rust shim of INTEGRATION.md would call.  It exists so that the
 * end-to-end metric (queries/sec and phase-0 / phase-1 latency with many concurrent callers) can be measured from
 * native threads, the way a multi-threaded Rust host drives the library.  libfshost.so has no GPU code of its own.
 */
#ifndef FSHOST_H
#define FSHOST_H

#include "fsgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fshost_two_tier fshost_two_tier;

/* TwoTierConfig defaults: crates/frankensearch-core/src/config.rs:169-176. */
typedef struct fshost_two_tier_config {
    float quality_weight;          /* 0.7 */
    double rrf_k;                  /* 60 */
    uint32_t candidate_multiplier; /* 3 */
    int32_t doc_id_mode;           /* 0 = the indexes' FSVI doc-id tables (fsgpu_index_doc_id); 1 = "doc-%08u" of the row */
    uint32_t fast_tier_int8_multiplier; /* 0 = exact f16 scan of the fast tier; n = search_top_k_int8_two_pass(query, fetch, n),
                                         * the reference's default with n = FAST_TIER_MULT = 3 (two_tier.rs:1318-1337,
                                         * sync_searcher::search_fast_hits) */
    int32_t prefetch_quality_embed;     /* != 0: the quality-tier (MiniLM) embedding of the query is started on a helper
                                         * thread when the search begins and joined at the start of phase 1, so it
                                         * overlaps the fast tier's scan (same results; for latency with few callers —
                                         * with many, coalescing fills the GPU and the extra thread only costs).
                                         * 2: the helper also runs the quality tier's search (it needs nothing phase 0
                                         * produces): the two scans share the GPU, phase 0 arrives a little later and
                                         * phase 1 much earlier */
    int32_t quality_pool;               /* FSHOST_POOL_RETRIEVED (0): the quality tier is searched independently and blended with
                                         * blend_two_tier — the reference's branch for an ATTESTED quality space (an admitted FSVI
                                         * v2 artifact), sync_searcher.rs:810-813.  FSHOST_POOL_RESCORED (1): the reference's
                                         * branch for every unattested pair — all FSVI v1 artifacts —: the fast pool is re-scored
                                         * on the quality tier (TwoTierIndex::quality_scores_for_hits, two_tier.rs:1566-1631: a
                                         * gather of ~k*multiplier rows instead of a scan) and blended with blend_two_tier_aligned
                                         * (sync_searcher.rs:814-818,862-866).  The alignment is computed at create. */
    int32_t quality_int8_latency;       /* != 0: a lone caller's quality-tier search goes through the int8 filter + exact re-score
                                         * (fsgpu_index_set_int8_latency on the quality handle while the searcher lives: the same
                                         * hits from half the bytes, at +50 % device memory for the int8 copy); 0 leaves the
                                         * caller's handle as it is */
} fshost_two_tier_config;
#define FSHOST_POOL_RETRIEVED 0
#define FSHOST_POOL_RESCORED 1

#define FSHOST_DOC_ID_MAX 63
/* FusedHit (crates/frankensearch-core/src/types.rs:3892-3925) with the doc id copied out. */
typedef struct fshost_hit {
    char doc_id[FSHOST_DOC_ID_MAX + 1]; /* NUL-terminated */
    double rrf_score;
    int64_t lexical_rank, semantic_rank; /* -1 when absent */
    uint32_t semantic_index;             /* 0xffffffff when absent */
    float lexical_score, semantic_score;
    uint8_t in_both_sources;
} fshost_hit;

/* TwoTierMetrics (crates/frankensearch-core/src/config.rs:465-480); milliseconds. */
typedef struct fshost_metrics {
    double fast_embed_ms, fast_search_ms, phase1_total_ms; /* library name of the Initial stage */
    double quality_embed_ms, quality_search_ms, blend_ms, phase2_total_ms;
    int32_t refinement_failed; /* SearchPhase::RefinementFailed (sync_searcher.rs:820-839): the quality pool could not be produced
                                * (a failing quality-tier search / quality_scores_for_hits); the search still returns FSGPU_OK and
                                * final_out holds the initial results, as the reference's final_results does */
} fshost_metrics;

/* The handles stay owned by the caller and must outlive the searcher. */
fsgpu_status fshost_two_tier_create(fsgpu_index *fast_index, fsgpu_index *quality_index, fsgpu_m2v *fast_embedder,
                                    fsgpu_bert *quality_embedder, const fshost_two_tier_config *config,
                                    fshost_two_tier **out);
/* The same searcher over two ROW-SHARDED tiers (fsgpu_sharded handles over the GPUs of one node, SURVEY 8e: the fast and quality
 * slabs shard identically).  Fast tier: fsgpu_sharded_search in INT8_TWO_PASS mode (the corpus-wide candidate set of
 * search_top_k_int8_two_pass) or EXACT; quality tier: EXACT (Retrieved) or fsgpu_sharded_quality_scores_for_hits — a gather routed to
 * the shards that own the rows (RescoredFastPool); doc ids from the handles' catalogs (doc_id_mode 0) or synthetic (1).  Fused
 * results equal the unsharded searcher's over the same rows.  quality_int8_latency switches every shard of the quality tier (fsgpu_sharded_set_int8_latency). */
fsgpu_status fshost_two_tier_create_sharded(fsgpu_sharded *fast_index, fsgpu_sharded *quality_index, fsgpu_m2v *fast_embedder,
                                            fsgpu_bert *quality_embedder, const fshost_two_tier_config *config,
                                            fshost_two_tier **out);
void fshost_two_tier_destroy(fshost_two_tier *s);

/* SyncTwoTierSearcher::search (sync_searcher.rs:616-943): phase 0 = fast embed -> fast-tier top-(k*mult) -> RRF with
 * the lexical list; phase 1 = quality embed -> quality-tier top-(k*mult) -> blend_two_tier -> RRF again.
 * initial_out / final_out hold k entries each.  Thread-safe: any number of concurrent callers. */
fsgpu_status fshost_two_tier_search(fshost_two_tier *s, const uint32_t *fast_token_ids, uint32_t n_fast_ids,
                                    const int32_t *quality_token_ids, uint32_t n_quality_ids, uint32_t k,
                                    const fsgpu_scored_doc *lexical, uint32_t n_lexical, fshost_hit *initial_out,
                                    uint32_t *n_initial, fshost_hit *final_out, uint32_t *n_final,
                                    fshost_metrics *metrics);

/* The same two-phase flow for MANY queries in one call — the throughput form (the reference's many-queries shapes:
 * crates/frankensearch-index/benches/batched_query_scan.rs, crates/frankensearch-embed/src/batch_coalescer.rs:18-23; the per-query
 * flow it must equal: sync_searcher.rs:616-943).  Query q owns fast_token_ids[fast_offsets[q] .. fast_offsets[q + 1]),
 * quality_token_ids[quality_offsets[q] .. quality_offsets[q + 1]) and the lexical list lexical[lexical_offsets[q] ..
 * lexical_offsets[q + 1]) (lexical / lexical_offsets may be NULL: no lexical source).  The queries run in chunks of `chunk` (0 = 1,024:
 * two 512-query passes of the matrix-core scan) through a pipeline: Model2Vec batch -> fast tier batched (int8 two-pass with
 * fast_tier_int8_multiplier, else the batched exact search) | MiniLM batch -> quality tier batched exact search, both tiers side by
 * side on the GPU and one chunk ahead of each other's embeddings, the embeddings handed over in device memory; the per-query RRF /
 * blend / RRF of a chunk runs on `fusion_threads` host threads (0 = automatic) as soon as its tiers have answered, with the functions
 * fshost_two_tier_search runs.  initial_out / final_out: [nq * k] (query q's hits at q * k), n_initial / n_final: [nq];
 * refinement_failed_out (may be NULL): [nq], 1 where the quality pool failed and final = initial (sync_searcher.rs:820-839).
 * quality_vectors_out / fast_vectors_out (may be NULL): [nq, dim] — the embeddings the tiers were searched with (asking for them
 * keeps that tier's vectors on the host path).
 * Results: those of fshost_two_tier_search on the same tier answers; the tier answers are bit-identical to the per-query searches'
 * for identical query vectors; the Model2Vec vectors are bit-identical whatever the batch; a text's MiniLM vector is within the
 * encoder's tolerance (cos >= 0.999, 2e-3) of its single-text embedding (the encoder picks kernels by batch shape).
 * doc_id_mode 0 (FSVI doc-id tables): exact tier searches go through fsgpu_search_hits query by query, as the per-query flow does.
 * quality_pool FSHOST_POOL_RESCORED: phase 1 is the gather of quality_scores_for_hits per query on the fusion threads. */
typedef struct fshost_many_result {
    double wall_seconds, queries_per_sec;
    double mean_fast_embed_ms, mean_fast_search_ms, mean_quality_embed_ms, mean_quality_search_ms; /* per chunk */
    double fusion_busy_ms_per_chunk;   /* summed over the fusion threads */
    double first_chunk_initial_ms, first_chunk_refined_ms; /* call start -> the first chunk's fast / quality tier answered */
    uint64_t queries, chunks, chunk_queries, fusion_threads;
    uint64_t refinement_failed, fast_fallbacks, quality_fallbacks;
    uint64_t device_resident_handoff;  /* bit 0: fast-tier vectors stayed in HBM, bit 1: quality-tier vectors */
    uint64_t queries_with_k_initial_and_refined_hits; /* fshost_run_load_many only */
    char error_detail[256];
} fshost_many_result;
fsgpu_status fshost_two_tier_search_many(fshost_two_tier *s, const uint32_t *fast_token_ids, const uint32_t *fast_offsets,
                                         const int32_t *quality_token_ids, const uint32_t *quality_offsets, uint32_t nq, uint32_t k,
                                         const fsgpu_scored_doc *lexical, const uint32_t *lexical_offsets, uint32_t chunk,
                                         uint32_t fusion_threads, fshost_hit *initial_out, uint32_t *n_initial, fshost_hit *final_out,
                                         uint32_t *n_final, uint8_t *refinement_failed_out, float *fast_vectors_out,
                                         float *quality_vectors_out, fshost_many_result *result);

/* Dynamic batching of CONCURRENT fshost_two_tier_search callers through the many-queries pipeline above: a caller's query is queued,
 * a collector turns whatever is queued into the next chunk (up to max_chunk queries of the same k; taken as soon as the pipeline has
 * room, so chunks grow with the arrival rate and a lone caller's query leaves at once; with fewer than max_chunk queued it lingers
 * while requests are still arriving, at most max_wait_us past the oldest), and the caller is woken ONCE, when its Refined list is
 * written (the per-stage coalescers of libfsgpu wake it four times per query).  Results as fshost_two_tier_search_many's.  In this
 * mode fshost_metrics carries phase1_total_ms (call -> Initial results written) and phase2_total_ms only.  max_chunk = 0 turns it
 * off (the default).  Replaces batch_coalescer.rs:18-23's role for the whole two-phase flow. */
fsgpu_status fshost_two_tier_set_batching(fshost_two_tier *s, uint32_t max_chunk, uint32_t max_wait_us);
fsgpu_status fshost_two_tier_batching_stats(fshost_two_tier *s, uint64_t *chunks, uint64_t *requests);

/* Closed-loop load generator: `threads` native threads each issue fshost_two_tier_search calls back to back on
 * synthetic queries (SURVEY §8d config 5 shapes: fast ids uniform in [0, fast_vocab), 4-23 tokens; quality ids
 * [CLS] + uniform [1000, quality_vocab) + [SEP], 8-32 tokens; stub lexical list of 3k "doc-%08u" ids), the way a
 * multi-threaded host drives the per-query API. */
typedef struct fshost_load_config {
    uint32_t threads;
    uint32_t queries;        /* timed queries in total */
    uint32_t warmup_queries; /* untimed, in total */
    uint32_t k;
    uint32_t fast_vocab;
    uint32_t quality_vocab;  /* quality ids are uniform in [1000, quality_vocab); 30000 for MiniLM's WordPiece table */
    uint64_t corpus_rows;    /* for the stub lexical list */
    uint64_t seed;
} fshost_load_config;

typedef struct fshost_load_result {
    double wall_seconds, queries_per_sec;
    double phase0_p50_ms, phase0_p95_ms, phase0_p99_ms; /* query start -> initial results */
    double phase1_p50_ms, phase1_p95_ms, phase1_p99_ms; /* query start -> refined results */
    double mean_fast_embed_ms, mean_fast_search_ms, mean_quality_embed_ms, mean_quality_search_ms, mean_fusion_ms;
    uint64_t completed, failed;
    char first_error[160];   /* detail of the first failed query, if any */
} fshost_load_result;

fsgpu_status fshost_run_load(fshost_two_tier *s, const fshost_load_config *config, fshost_load_result *result);
/* The load generator's synthetic queries (config->queries of them, after one untimed call over config->warmup_queries) through ONE
 * fshost_two_tier_search_many call: the throughput of the two-phase flow when the host has a queue of requests instead of a
 * thread per request.  config->threads = fusion threads (0 = automatic). */
fsgpu_status fshost_run_load_many(fshost_two_tier *s, const fshost_load_config *config, uint32_t chunk, fshost_many_result *result);

/* BASELINE config 5's serving loop: batches of token-id queries -> MiniLM forward on the GPU (fsgpu_bert_embed) ->
 * batched exact top-k of the embeddings (fsgpu_search_topk_batched on an index, or fsgpu_sharded_search in BATCHED mode on a
 * row-sharded handle; exactly one of `index` / `sharded` is non-null).  overlap != 0: a second host thread encodes group g + 1
 * while group g is searched (what a Rust host's rayon::join of the two stages does; the encoder's stream has the higher
 * priority, so its short kernels are not queued behind the chip-filling scan launches); overlap == 0 runs the two stages in
 * turn on the calling thread.  `overlap` is a bit set: 1 = overlapped, 2 = keep the embeddings on the HOST path (fsgpu_bert_embed ->
 * host vectors -> host-pointer search).  Without bit 2, an encoder that shares its device with the index (a sharded handle's root
 * shard) hands the vectors over in device memory: fsgpu_bert_embed_device -> fsgpu_search_topk_batched_device_queries /
 * fsgpu_sharded_request::queries_dev (the other shards fetch them from the root peer to peer).  Text t owns ids[offsets[t] .. offsets[t + 1]); the texts are taken batch by batch, `group`
 * batches per search (1: one search per encoder batch; 2: two encoder batches share each pass over the slab).  out_* hold
 * [n_batches * batch, k] rows / scores and [n_batches * batch] counts (any may be NULL). */
typedef struct fshost_stream_result {
    double wall_seconds, queries_per_sec;
    double mean_encode_ms, mean_search_ms; /* per group */
    uint64_t queries, groups, exact_fallbacks;
    uint64_t device_resident_handoff; /* 1: the embeddings went from the encoder to the search in device memory */
    uint64_t encoders;                /* encoder handles that shared every group's texts (1: the single-encoder form) */
    char error_detail[256];           /* fsgpu_last_error of the call that failed, whichever thread made it ("" on success):
                                       * fsgpu_last_error is thread-local and the encoders run on threads of their own */
} fshost_stream_result;
fsgpu_status fshost_embed_search_stream(fsgpu_bert *encoder, fsgpu_index *index, fsgpu_sharded *sharded, const int32_t *ids,
                                        const uint32_t *offsets, uint32_t batch, uint32_t n_batches, uint32_t group, uint32_t k,
                                        int32_t overlap, uint32_t *out_rows, float *out_scores, uint32_t *out_counts,
                                        fshost_stream_result *result);

/* The same loop with DATA-PARALLEL encoders over a sharded handle ("Encoders: data-parallel over the query batch"): one
 * encoder handle per device (any subset of the sharded handle's devices); encoder e embeds the e-th contiguous slice of every
 * group's texts on its own device, the vectors stay there, and the search fetches each device's slice of its query group peer to
 * peer (fsgpu_sharded_search_parts).  overlap bit 1 as above (bit 2 has no meaning here: there is no host path). */
fsgpu_status fshost_embed_search_stream_dp(fsgpu_bert *const *encoders, uint32_t n_encoders, fsgpu_sharded *sharded, const int32_t *ids,
                                           const uint32_t *offsets, uint32_t batch, uint32_t n_batches, uint32_t group, uint32_t k,
                                           int32_t overlap, uint32_t *out_rows, float *out_scores, uint32_t *out_counts,
                                           fshost_stream_result *result);

#ifdef __cplusplus
}
#endif
#endif /* FSHOST_H */
