/*
 * fsgpu.h — C ABI of libfsgpu.so, the MI355X (gfx950) semantic tier for frankensearch.
 *
 * The reference has no FFI for this path: its seams are Rust traits and one concrete method
 * set.  Each entry point below names the reference interface it replaces
 * (file:line relative to the reference repo) so a ~100-line Rust shim can bind it
 * (see INTEGRATION.md for that shim).
 *
 * Conventions
 *   - every function returns an fsgpu_status (0 = OK); no exception crosses the boundary;
 *   - fsgpu_last_error() returns the calling thread's last failure detail (UTF-8);
 *   - all pointers are caller-owned unless stated; "_dev" pointers are HIP device pointers;
 *   - handles may be used from many host threads.  fsgpu_search_topk calls (row-level searches, filtered or not) from
 *     different threads run side by side on up to four lanes of an index (own HIP stream and workspaces each) — the
 *     reference's scan is `&self` and lock-free (search.rs:192); every other call on a handle serialises internally, and
 *     calls that change an index (tombstones, WAL, live bitmap) wait for the searches in flight;
 *   - row ids are GLOBAL physical row ids (row_base + local row), u32 like VectorHit.index
 *     (crates/frankensearch-core/src/types.rs:88-95);
 *   - result order is the reference's: score descending with NaN ranked as -inf, f32 total
 *     order (-0.0 < +0.0), ties by ascending row (crates/frankensearch-index/src/search.rs:1655-1686);
 *   - scores are computed in the reference's exact accumulation order (simd.rs:398-446), so they
 *     are bit-identical to the CPU path's scores for a given FSGPU_HREDUCE_* mode.
 */
#ifndef FSGPU_H
#define FSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t fsgpu_status;

/* Status codes <-> SearchError variants (crates/frankensearch-core/src/error.rs:57-176). */
#define FSGPU_OK 0
#define FSGPU_ERR_DIMENSION_MISMATCH 1     /* SearchError::DimensionMismatch{expected,found} */
#define FSGPU_ERR_INVALID_CONFIG 2         /* SearchError::InvalidConfig{field,value,reason} */
#define FSGPU_ERR_INDEX_CORRUPTED 3        /* SearchError::IndexCorrupted{path,detail} */
#define FSGPU_ERR_INDEX_VERSION_MISMATCH 4 /* SearchError::IndexVersionMismatch{expected,found} */
#define FSGPU_ERR_IO 5                     /* SearchError::Io */
#define FSGPU_ERR_DEVICE 6                 /* HIP runtime failure (no reference analogue) */
#define FSGPU_ERR_NO_DEVICE 7              /* no gfx950 device visible */
#define FSGPU_ERR_NULL_ARGUMENT 8
#define FSGPU_ERR_EMBEDDING_FAILED 9       /* SearchError::EmbeddingFailed */
#define FSGPU_ERR_MODEL_LOAD_FAILED 10     /* SearchError::ModelLoadFailed{path,source} */

/* ZeroSignalReason of search_top_k_classified (search.rs:227-261). */
#define FSGPU_ZERO_SIGNAL_NONE 0
#define FSGPU_ZERO_SIGNAL_CALLER_REQUESTED_ZERO_K 1
#define FSGPU_ZERO_SIGNAL_ZERO_NORM_QUERY 2
/* empty_result_reason (crates/frankensearch-core/src/config.rs:696-740) for a well-formed search that came back empty.  The
 * classified entry point takes no filter, so FILTER_ELIMINATED_ALL and NO_USABLE_VECTORS (an exact scan returns every live
 * row, whatever its score) are listed for callers that classify filtered searches themselves. */
#define FSGPU_ZERO_SIGNAL_FILTER_ELIMINATED_ALL 3
#define FSGPU_ZERO_SIGNAL_NEWLY_CREATED_EMPTY 4      /* no main records, no WAL entries */
#define FSGPU_ZERO_SIGNAL_ALL_TOMBSTONED 5           /* every main record tombstoned, no WAL entry */
#define FSGPU_ZERO_SIGNAL_WAL_ONLY_NO_LIVE_RECORDS 6 /* no live main record; the WAL entries produced no usable hit */
#define FSGPU_ZERO_SIGNAL_NO_USABLE_VECTORS 7

/* Order of the final 8-lane horizontal add (`wide::f32x8::reduce_add`, simd.rs:439,563).
 * SSE2 is the reference's default build (no +avx2 in .cargo/config.toml). */
#define FSGPU_HREDUCE_SSE2 0 /* ((v0+v2)+(v1+v3)) + ((v4+v6)+(v5+v7)) */
#define FSGPU_HREDUCE_AVX 1  /* ((v0+v4)+(v2+v6)) + ((v1+v5)+(v3+v7)) */
#define FSGPU_HREDUCE_SEQ 2  /* (((v0+v1)+v2)+v3) + (((v4+v5)+v6)+v7): f32x8 as a.reduce_add() + b.reduce_add() over a
                              * left-to-right f32x4 sum (the fallback shape `wide` has shipped); INTEGRATION.md shows how a
                              * maintainer finds out which of the three the Rust build uses */

typedef struct fsgpu_index fsgpu_index;     /* device-resident VectorIndex (lib.rs:819) */
typedef struct fsgpu_m2v fsgpu_m2v;         /* Model2VecEmbedder (embed/src/model2vec_embedder.rs:55) */
typedef struct fsgpu_bert fsgpu_bert;       /* NativeEmbedder (rerank/src/native_embedder.rs:40-50) */

/* BERT shape (all-MiniLM-L6-v2: vocab 30522, hidden 384, layers 6, heads 12, inter 1536, max_pos 512,
 * ln_eps 1e-12; crates/frankensearch-rerank/src/native.rs:36-45).  heads*32 must equal hidden. */
typedef struct fsgpu_bert_config {
    uint32_t vocab, hidden, layers, heads, inter, max_pos;
    float ln_eps;
} fsgpu_bert_config;
/* One encoder layer in HuggingFace tensor layout ([out, in] row-major f32), the keys parse_weights reads
 * (native.rs:1359-1602): attention.self.{query,key,value}, attention.output.{dense,LayerNorm},
 * intermediate.dense, output.{dense,LayerNorm}. */
typedef struct fsgpu_bert_layer_weights {
    const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b;
    const float *ao_w, *ao_b, *ln1_w, *ln1_b;
    const float *i_w, *i_b, *o_w, *o_b, *ln2_w, *ln2_b;
} fsgpu_bert_layer_weights;
typedef struct fsgpu_bert_weights {
    const float *word_emb, *pos_emb, *type_emb, *emb_ln_w, *emb_ln_b;
    const fsgpu_bert_layer_weights *layers; /* [config.layers] */
} fsgpu_bert_weights;

/* ---- library ---- */
const char *fsgpu_version(void);
int32_t fsgpu_device_count(void);
const char *fsgpu_last_error(void);

/* ---- index lifecycle ---- */
/* Replaces VectorIndex::open + the mmap of the slab (lib.rs:1747-1816): copies `nrows` x `dim`
 * little-endian f16 rows from host memory to device `device`.  live_bitmap (bit r = row r is live,
 * i.e. RECORD_FLAG_TOMBSTONE clear, lib.rs:171-173) may be NULL = all live.  row_base is added to
 * every reported row id (shard offset). */
fsgpu_status fsgpu_index_create(int32_t device, uint32_t dim, uint64_t nrows, const void *slab_f16_le,
                                const uint64_t *live_bitmap, uint64_t row_base, fsgpu_index **out);
/* Zero-copy variant: adopts (does not free) a device-resident slab / bitmap. */
fsgpu_status fsgpu_index_create_device(int32_t device, uint32_t dim, uint64_t nrows, const void *slab_f16_dev,
                                       const uint64_t *live_bitmap_dev, uint64_t row_base, fsgpu_index **out);
/* VectorIndex::open for an FSVI v1 / F16 file (lib.rs:4049-4144 header+CRC, :3510-3537 record table):
 * uploads the slab, builds the live bitmap from record flags, keeps the doc-id table on the host. */
fsgpu_status fsgpu_index_open_fsvi(const char *path, int32_t device, fsgpu_index **out);
void fsgpu_index_destroy(fsgpu_index *idx);

uint64_t fsgpu_index_record_count(const fsgpu_index *idx); /* VectorIndex::record_count */
uint32_t fsgpu_index_dimension(const fsgpu_index *idx);    /* VectorIndex::dimension */
fsgpu_status fsgpu_index_set_hreduce(fsgpu_index *idx, int32_t mode);
/* doc id of a global row (FSVI-opened indexes only; pointer valid until destroy; not NUL-terminated). */
fsgpu_status fsgpu_index_doc_id(const fsgpu_index *idx, uint32_t row, const char **ptr, uint32_t *len);
/* VectorIndex::soft_delete = soft_delete_batch(&[id]) > 0 (lib.rs:2303-2397): clears the live bit of every main row with
 * that doc id AND drops its resident WAL entries; *deleted = 1 if anything went live -> deleted. */
fsgpu_status fsgpu_index_soft_delete(fsgpu_index *idx, const char *doc_id, uint32_t doc_id_len, int32_t *deleted);
/* SearchFilter::candidate_hashes -> rows (gather_positions_for_hashes / hash_range, search.rs:1146-1198): sets, in
 * allow_bitmap_out (ceil(record_count / 64) words, cleared first), the bit of every main row whose record hash
 * (FNV-1a 64 of the doc id, lib.rs:6120-6127) is one of `hashes` — a binary search per hash in the (hash, doc_id)-
 * sorted record table.  The bitmap is what fsgpu_search_topk takes as its filter (tombstoned rows are masked there).
 * FSVI-opened indexes only.  *rows_matched (may be NULL) = bits set. */
fsgpu_status fsgpu_index_allow_bitmap_for_hashes(const fsgpu_index *idx, const uint64_t *hashes, uint32_t n,
                                                 uint64_t *allow_bitmap_out, uint64_t *rows_matched);
/* VectorIndex::append (lib.rs:2532-2720): a resident WAL entry (f32 vector, immediately searchable through
 * fsgpu_search_hits with the reference's scan_wal / shadowing rules, search.rs:1449-1475,1503-1558); supersedes a
 * resident entry with the same doc id and tombstones the first live main row with that doc id.  The WAL *file*
 * (durability) stays with the caller.  FSVI-opened indexes only. */
fsgpu_status fsgpu_index_wal_append(fsgpu_index *idx, const char *doc_id, uint32_t doc_id_len, const float *vector,
                                    uint32_t vector_len);
uint64_t fsgpu_index_wal_record_count(const fsgpu_index *idx); /* VectorIndex::wal_record_count */
/* Replace the live bitmap (host words, ceil(nrows/64)); NULL = all live. */
fsgpu_status fsgpu_index_set_live_bitmap(fsgpu_index *idx, const uint64_t *live_bitmap);

/* ---- search ---- */
/* Replaces VectorIndex::search_top_k(query, limit, None) (search.rs:192-206 -> :426-494) for nq queries
 * at once: tombstones via the live bitmap, optional per-call allow bitmap (a precomputed SearchFilter,
 * filter.rs:19-56), collect-all when k >= nrows.  query_len must equal the index dimension
 * (else FSGPU_ERR_DIMENSION_MISMATCH, search.rs:1602-1610).  Outputs are [nq,k] row-major, best first;
 * out_counts[q] <= k entries are valid.  No doc-id dedup (see fsgpu_search_hits). */
fsgpu_status fsgpu_search_topk(fsgpu_index *idx, const float *queries, uint32_t nq, uint32_t query_len,
                               uint32_t k, const uint64_t *allow_bitmap, uint32_t *out_rows,
                               float *out_scores, uint32_t *out_counts);
/* The same search on the exact f16 kernels, whatever quantised copies the index holds.  fsgpu_search_topk itself answers a LONE
 * unfiltered query (nq = 1, k <= 32) of an index that already holds the int8 copy of its slab — a batched search built it — with ONE
 * certified pass over that copy + an exact re-score of the rows within the proven margin: the rows and f32 score bits of these kernels
 * from half the bytes (a query the certificate does not cover goes to these kernels).  This entry is the checked kernels alone
 * (no coalescing, no int8 pass): what the parity tests compare everything else with. */
fsgpu_status fsgpu_search_topk_exact(fsgpu_index *idx, const float *queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                     const uint64_t *allow_bitmap, uint32_t *out_rows, float *out_scores, uint32_t *out_counts);
/* A SearchFilter is typically built once and reused by many queries (filter.rs:19-56; the filtered search paths search.rs:1114-1255
 * take `Option<&dyn SearchFilter>` per call).  fsgpu_allow_bitmap is the precomputed filter made RESIDENT on the index's device:
 * uploaded once (1.25 MB at 10M rows), then any number of fsgpu_search_topk_filtered / _batched_filtered calls use it without a
 * per-call copy — same hits as fsgpu_search_topk{,_batched} with the same words, same 1/50 selective-gather rule, tombstones still
 * masked at search time.  Concurrent single-query callers that pass the SAME handle share a coalesced batch.  The bitmap belongs to
 * the index it was created for (device and record count are checked). */
typedef struct fsgpu_allow_bitmap fsgpu_allow_bitmap;
fsgpu_status fsgpu_allow_bitmap_create(fsgpu_index *idx, const uint64_t *allow_bitmap, fsgpu_allow_bitmap **out);
void fsgpu_allow_bitmap_destroy(fsgpu_allow_bitmap *filter);
uint64_t fsgpu_allow_bitmap_allowed_rows(const fsgpu_allow_bitmap *filter);
fsgpu_status fsgpu_search_topk_filtered(fsgpu_index *idx, const float *queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                        const fsgpu_allow_bitmap *filter, uint32_t *out_rows, float *out_scores,
                                        uint32_t *out_counts);
fsgpu_status fsgpu_search_topk_batched_filtered(fsgpu_index *idx, const float *queries, uint32_t nq, uint32_t query_len,
                                                uint32_t k, const fsgpu_allow_bitmap *filter, uint32_t *out_rows,
                                                float *out_scores, uint32_t *out_counts, uint32_t *out_fallbacks);
/* Same, all pointers device-resident, enqueued on `hip_stream` (a hipStream_t), no host sync. */
fsgpu_status fsgpu_search_topk_device(fsgpu_index *idx, const float *queries_dev, uint32_t nq,
                                      uint32_t query_len, uint32_t k, const uint64_t *allow_bitmap_dev,
                                      uint32_t *out_rows_dev, float *out_scores_dev, uint32_t *out_counts_dev,
                                      void *hip_stream);
/* Throughput form of fsgpu_search_topk for large query batches: groups of 64 queries share ONE pass over the
 * slab on the matrix cores (f16-rounded queries, v_mfma_f32_16x16x32_f16); rows that could reach the top k under a
 * proven error bound are re-scored in the reference's exact order, so rows and score bits are IDENTICAL to
 * fsgpu_search_topk.  Queries the batched path cannot certify (k > 64, unsupported dimension, margin overflow) are
 * answered by the exact kernels; *out_fallbacks (optional) counts them.  The _device form synchronises hip_stream. */
/* The filter those passes score with is chosen per index (FSGPU_FILTER_AUTO): batches of 16 queries and more are
 * filtered on an int8 copy of the slab (built on first use, half the slab's size again; v_mfma_i32_16x16x64_i8: half the
 * bytes and half the matrix instructions per row) under a bound measured from the slab and each query (mfma_scan.hip,
 * prepare_queries_i8_filter_kernel); queries whose margin lets too many rows through are re-filtered on the f16 slab, and an
 * index where that happens to more than 1/8 of a batch twice in a row (or that has no room for the copy) stays with the
 * f16 filter.  Either way the emitted rows and score bits are the exact search's. */
#define FSGPU_FILTER_AUTO 0
#define FSGPU_FILTER_F16 1
#define FSGPU_FILTER_INT8 2
fsgpu_status fsgpu_index_set_batched_filter(fsgpu_index *idx, int32_t filter);
/* The int8 filter's copy of a slab WITH OUTLIER CHANNELS (a few dimensions carrying most of every row's norm, as trained embedding
 * models have) is built from ROTATED rows: one corpus-wide int8 scale (simd.rs:1865-1886) spends the int8 range on those channels, and
 * the filter's proven margin — fixed in integer units — is then several times wider in cosine units than it has to be.  Dot products
 * are invariant under an orthogonal map: the filter scores quantised R x against quantised R q (R fixed, random, applied in f64 and
 * rounded once to f32; what that adds to the proven bound is 2^-24-sized), and the exact re-score that decides rows and score bits never
 * sees the rotation.  AUTO (default): rotate when the slab's largest |element| x sqrt(dim) exceeds 9 x its largest row norm — decided
 * once, when the copy is built (first batched search).  The reference has no counterpart (its int8 slab serves the two-pass search,
 * which keeps the reference's own quantisation here too).  Takes effect for a copy that is not built yet. */
#define FSGPU_ROTATION_AUTO 0
#define FSGPU_ROTATION_OFF 1
#define FSGPU_ROTATION_ON 2
fsgpu_status fsgpu_index_set_filter_rotation(fsgpu_index *idx, int32_t mode);
int32_t fsgpu_index_filter_rotated(fsgpu_index *idx); /* 1: the filter's copy of this index is the rotated one */
/* Latency form of the same idea: with this set, unfiltered fsgpu_search_topk calls of up to 16 queries (k <= 64) are answered
 * through the int8 filter + exact re-score as well — the pass streams half the bytes, rows and score bits unchanged; the int8 copy is
 * built at the first such call.  A lone query (k <= 32) takes ONE certified pass: every block of the int8 scan keeps its 32 best entries,
 * the rows within the proven margin of the k-th are re-scored from the f16 slab, and the answer stands when no block can have dropped a
 * row within that margin (one query at 10M x 384: p50 0.67 ms against 1.26 ms; 1M x 384: 0.126 against 0.161); an uncertified query
 * takes the staged path.  Off by default: fsgpu_search_topk then takes the certified pass only for a lone query of an index whose int8
 * copy a batched search has already built (see fsgpu_search_topk_exact above) and never builds the copy itself; the two-tier host
 * (libfshost) sets it on the quality tier.
 * enabled = FSGPU_INT8_LATENCY_BUILD_NOW: on, and the int8 copy + its statistics are built before the call returns (index-build work,
 * ~4 ms per GB of slab), so that the latency of the first lone query does not depend on what was searched before it. */
#define FSGPU_INT8_LATENCY_BUILD_NOW 2
fsgpu_status fsgpu_index_set_int8_latency(fsgpu_index *idx, int32_t enabled);
/* Queries the int8 filter has taken so far, how many of them it handed on to the f16 filter, and whether the index still
 * uses it (any pointer may be null). */
fsgpu_status fsgpu_index_batched_filter_stats(fsgpu_index *idx, uint64_t *int8_queries, uint64_t *refiltered_f16,
                                              int32_t *int8_active);
/* The int8 filter's certificate, for inspection (tests/test_gpu_int8_filter.py checks it against float64 arithmetic): per
 * query the bound out_delta[q] >= |int8 score - exact score * slab scale * query scale| over every row (< 0: the query
 * cannot be certified and goes to the f16 filter), the scales 127 / max|q| and 127 / max|x|, the quantised queries
 * [nq, dim] and the int8 slab [rows, dim] (any output may be null).  Builds the int8 copy if it does not exist yet. */
fsgpu_status fsgpu_index_int8_filter_bound(fsgpu_index *idx, const float *queries, uint32_t nq, uint32_t query_len,
                                           float *out_delta, float *out_query_scale, float *out_slab_scale,
                                           int8_t *out_queries_i8, int8_t *out_slab_i8);
fsgpu_status fsgpu_search_topk_batched(fsgpu_index *idx, const float *queries, uint32_t nq, uint32_t query_len,
                                       uint32_t k, const uint64_t *allow_bitmap, uint32_t *out_rows, float *out_scores,
                                       uint32_t *out_counts, uint32_t *out_fallbacks);
fsgpu_status fsgpu_search_topk_batched_device(fsgpu_index *idx, const float *queries_dev, uint32_t nq,
                                              uint32_t query_len, uint32_t k, const uint64_t *allow_bitmap_dev,
                                              uint32_t *out_rows_dev, float *out_scores_dev, uint32_t *out_counts_dev,
                                              void *hip_stream, uint32_t *out_fallbacks);
/* Shard-local search for the multi-GPU path (the reference partitions the same way per
 * rayon chunk, search.rs:1020-1035): like fsgpu_search_topk_device but the result stays PACKED,
 * out_packed_dev[q*k+i] = (f32 score bits << 32) | global row, best first, ~0ull padding — one
 * 8-byte word per hit, ready for an RCCL all-gather.  k <= 256, dim % 8 == 0. */
fsgpu_status fsgpu_search_topk_packed_device(fsgpu_index *idx, const float *queries_dev, uint32_t nq,
                                             uint32_t query_len, uint32_t k, const uint64_t *allow_bitmap_dev,
                                             uint64_t *out_packed_dev, void *hip_stream);
/* Batched (matrix-core) form of fsgpu_search_topk_packed_device: same packed output, same exact results, 64 queries
 * per pass.  Synchronises hip_stream. */
fsgpu_status fsgpu_search_topk_batched_packed_device(fsgpu_index *idx, const float *queries_dev, uint32_t nq,
                                                     uint32_t query_len, uint32_t k, const uint64_t *allow_bitmap_dev,
                                                     uint64_t *out_packed_dev, void *hip_stream, uint32_t *out_fallbacks);
/* The batched searches in two halves, for a caller that keeps its GPU fed: _begin enqueues the WHOLE search on hip_stream and returns
 * a ticket without waiting (0 / 1: two may be outstanding per index); _end waits for that search's last kernel — an event, not the
 * stream, which may already hold the caller's next search —, reads the per-query verdicts and answers the rare uncertified query with
 * the exact kernels, exactly as the blocking forms do at their end.  Outputs as in fsgpu_search_topk_batched_device (rows / scores /
 * counts, each may be null) plus out_packed_dev as in the packed form (may be null); the queries and every output stay the caller's
 * until _end.  Between a _begin and its _end the index may begin ONE more search; blocking calls on the same index are allowed:
 * a batched search on another stream than an outstanding ticket's is ordered behind that ticket's last kernel on the device (all
 * batched searches of an index share one set of workspaces), on the same stream it queues behind by itself.  Calls that CHANGE
 * the index (fsgpu_index_set_live_bitmap, _soft_delete, _wal_append) are refused with FSGPU_ERR_INVALID_CONFIG while a ticket is
 * outstanding.  The per-rank loop of `bench.py --gpus N` (frankensearch_amd/sharded.py) runs on these: the ~50 us between
 * one blocking call's wake-up and the next call's first kernel were a tenth of a 1.25M-row shard's step.  No reference counterpart
 * (the reference's scan is synchronous CPU code, search.rs:1013-1080). */
fsgpu_status fsgpu_search_topk_batched_device_begin(fsgpu_index *idx, const float *queries_dev, uint32_t nq, uint32_t query_len,
                                                    uint32_t k, const uint64_t *allow_bitmap_dev, uint32_t *out_rows_dev,
                                                    float *out_scores_dev, uint32_t *out_counts_dev, uint64_t *out_packed_dev,
                                                    void *hip_stream, int32_t *out_ticket);
fsgpu_status fsgpu_search_topk_batched_device_end(fsgpu_index *idx, int32_t ticket, uint32_t *out_fallbacks);
/* ... the same, also reporting how many queries were ANSWERED IN THE END HALF — by the exact kernels (*out_fallbacks) or by the f16
 * filter after the int8 filter handed them on (re-filtered: certified, not a fallback).  Their hits are written by work enqueued on
 * hip_stream inside this call: a caller that chained work to the last kernel of _begin (a shard's all-gather behind an event recorded
 * right after _begin returned) must chain it again when *out_late_answers != 0. */
fsgpu_status fsgpu_search_topk_batched_device_end_late(fsgpu_index *idx, int32_t ticket, uint32_t *out_fallbacks,
                                                       uint32_t *out_late_answers);
/* One-shot hook for the NEXT batched search on this handle: `fn(ctx)` is called on the calling thread once every kernel of
 * that search is enqueued and before the call blocks on its stream (it has to read the certificate flags back).  A launcher
 * that pipelines steps uses the window to enqueue the PREVIOUS step's exchange (all-gather + merge on another stream), so
 * that host work runs under the scan instead of between two scans.  The hook must not call into this handle.  A search that
 * leaves the matrix-core path (odd shapes) returns without calling it; the hook is cleared either way. */
typedef void (*fsgpu_after_enqueue_fn)(void *ctx);
fsgpu_status fsgpu_index_set_after_enqueue_hook(fsgpu_index *idx, fsgpu_after_enqueue_fn fn, void *ctx);
/* Merge of gathered packed lists = merge_partial_heaps + resolve_hits sort (search.rs:1704-1720,1493-1501)
 * across shards: entry (q, l, i) is lists_dev[q*q_stride + l*l_stride + i]; selects the k best per query
 * under the reference order.  For an all-gather result laid out [nlists][nq][list_len]:
 * q_stride = list_len, l_stride = nq*list_len. */
fsgpu_status fsgpu_merge_topk_device(int32_t device, const uint64_t *lists_dev, uint32_t nq, uint32_t nlists,
                                     uint32_t list_len, uint64_t q_stride, uint64_t l_stride, uint32_t k,
                                     uint32_t *out_rows_dev, float *out_scores_dev, uint32_t *out_counts_dev,
                                     void *hip_stream);
/* search_top_k_classified (search.rs:227-261): validates the query (non-finite -> INVALID_CONFIG),
 * reports the ZeroSignalReason, then searches one query (through fsgpu_search_hits when the index has a doc-id
 * table, so resident WAL entries count; row-level otherwise). */
fsgpu_status fsgpu_search_topk_classified(fsgpu_index *idx, const float *query, uint32_t query_len, uint32_t k,
                                          uint32_t *out_rows, float *out_scores, uint32_t *out_count,
                                          int32_t *zero_signal);
/* search_top_k + scan_wal + resolve_hits (search.rs:426-494, 1449-1475, 1493-1558) for FSVI-opened indexes:
 * GPU top-k of the main rows, merge of the resident WAL entries (dot_product_f32_f32 on the host, as in the
 * reference), WAL shadowing, post-top-k doc-id dedup (first = best wins).  WAL hits report the virtual row
 * record_count + wal index.  out_* hold up to k entries. */
fsgpu_status fsgpu_search_hits(fsgpu_index *idx, const float *query, uint32_t query_len, uint32_t k,
                               uint32_t *out_rows, float *out_scores, uint32_t *out_count);
/* VectorIndex::search_top_k_int8_two_pass(query, k, candidate_multiplier) (search.rs:514-661) — the reference's
 * production default for the fast tier (two_tier.rs:1332-1337, multiplier 3): pass 1 scans a lazily built int8
 * slab (one corpus-wide max-abs scale, simd.rs:1865-1886) with the integer-exact dot and keeps the top
 * max(min(k*mult, N), min(k, N)) rows under (int score desc, row asc); pass 2 re-scores those rows with the exact
 * f16 dot and selects k under the usual order; doc-id dedup when the index has a doc-id table.  Falls back to the
 * exact search when a WAL is resident.  Halves the HBM bytes of a pass (N*dim). */
fsgpu_status fsgpu_search_topk_int8_two_pass(fsgpu_index *idx, const float *query, uint32_t query_len, uint32_t k,
                                             uint32_t candidate_multiplier, uint32_t *out_rows, float *out_scores,
                                             uint32_t *out_count);
/* Batched search_top_k_int8_two_pass: nq queries share each pass over the int8 slab (int8 MFMA, exact integer
 * scores: the k*candidate_multiplier candidates of every query are exactly the reference's), then the exact f16 rescore
 * and the best-first selection of k.  Outputs as fsgpu_search_topk ([nq, k] rows / scores, [nq] counts).  Row-level
 * results: indexes with a doc-id table, a resident WAL, k = 0 or shapes the matrix-core kernel does not cover are
 * answered query by query through fsgpu_search_topk_int8_two_pass (counted in *out_fallbacks, may be NULL). */
fsgpu_status fsgpu_search_topk_int8_two_pass_batched(fsgpu_index *idx, const float *queries, uint32_t nq, uint32_t query_len,
                                                     uint32_t k, uint32_t candidate_multiplier, uint32_t *out_rows,
                                                     float *out_scores, uint32_t *out_counts, uint32_t *out_fallbacks);

/* VectorIndex::search_top_k_4bit_two_pass (crates/frankensearch-index/src/search.rs:876-946): pass 1 over a packed
 * signed-4-bit slab (dim/2 bytes per vector, one corpus-wide scale 7/max_abs, simd.rs:2153-2215; exact integer nibble
 * dot, simd.rs:1338-1556) keeps the top k*candidate_multiplier, pass 2 re-scores them with the exact f16 dot.  Same
 * arguments, outputs and fallbacks (WAL resident, k = 0, empty index -> exact search) as the int8 variant. */
fsgpu_status fsgpu_search_topk_4bit_two_pass(fsgpu_index *idx, const float *query, uint32_t query_len, uint32_t k,
                                             uint32_t candidate_multiplier, uint32_t *out_rows, float *out_scores,
                                             uint32_t *out_count);
/* Batched search_top_k_4bit_two_pass: as fsgpu_search_topk_int8_two_pass_batched, with the reference's 4-bit quantisers
 * (slab scale 7/max_abs, pack_4bit_query).  The nibble dot is an exact integer, so the levels are kept one per byte (built on
 * first use, rows x dim bytes) and run through the same int8 matrix-core pass — which is bound by matrix instructions, not
 * by bytes — giving exactly the reference's k*candidate_multiplier candidates; then the exact f16 rescore. */
fsgpu_status fsgpu_search_topk_4bit_two_pass_batched(fsgpu_index *idx, const float *queries, uint32_t nq, uint32_t query_len,
                                                     uint32_t k, uint32_t candidate_multiplier, uint32_t *out_rows,
                                                     float *out_scores, uint32_t *out_counts, uint32_t *out_fallbacks);
/* VectorIndex::dot_query_at (lib.rs:3229-3239) over a row list, as used by
 * TwoTierIndex::quality_scores_for_hits (two_tier.rs:1566-1631).  rows are global ids. */
fsgpu_status fsgpu_gather_dot(fsgpu_index *idx, const float *query, uint32_t query_len, const uint32_t *rows,
                              uint32_t n, float *out_scores);

/* ---- row-sharded index over the GPUs of one node ---- */
/* The reference's only partitioning is scan_parallel's contiguous row chunks merged by merge_partial_heaps
 * (crates/frankensearch-index/src/search.rs:1013-1036,1704-1720).  A sharded handle is the same shape across GPUs, inside the
 * library, so that the host makes ONE call per search: shard r (on devices[r]) owns the contiguous rows
 * [r*ceil(N/ndev), ...) and reports global row ids; queries are replicated; each shard scans on its own stream; one
 * ncclAllGather (RCCL over xGMI, nq*k*8 bytes per shard; issued for all shards inside one ncclGroupStart/End from the calling
 * thread, on per-shard exchange streams behind an event of the scan — no host wait between scan, all-gather and merge) gathers
 * the packed per-shard top-k lists and the root device merges them under the reference order.  Results are identical to an
 * unsharded index over the same rows, for every entry point below. */
typedef struct fsgpu_sharded fsgpu_sharded;
#define FSGPU_EXCHANGE_AUTO 0      /* RCCL when the devices are distinct and librccl.so.1 loads, else peer copies */
#define FSGPU_EXCHANGE_RCCL 1      /* ncclCommInitAll + ncclAllGather; creation fails if RCCL cannot be used */
#define FSGPU_EXCHANGE_PEER_COPY 2 /* device-to-device copies into the root's gather buffer (also: several shards on ONE
                                      device, a rehearsal of the N-way path on fewer GPUs — RCCL wants one rank per device) */
/* VectorIndex::open for a host slab of nrows x dim little-endian f16 rows, split over devices[0..ndev).  live_bitmap as in
 * fsgpu_index_create (bit r = global row r is live; NULL = all live). */
fsgpu_status fsgpu_sharded_create(const int32_t *devices, uint32_t ndev, uint32_t dim, uint64_t nrows, const void *slab_f16_le,
                                  const uint64_t *live_bitmap, int32_t exchange, fsgpu_sharded **out);
/* Zero-copy variant: shard r adopts (does not free) shard_rows[r] rows already resident on devices[r] (shard_live_dev and its
 * entries may be NULL); global row ids follow the order of the shards. */
fsgpu_status fsgpu_sharded_create_device(const int32_t *devices, uint32_t ndev, uint32_t dim, const uint64_t *shard_rows,
                                         const void *const *shard_slabs_dev, const uint64_t *const *shard_live_dev,
                                         int32_t exchange, fsgpu_sharded **out);
/* Hybrid layout (round 5): ndev = query_groups x row shards.  Device r holds row shard r % (ndev / query_groups) — every row shard is
 * resident once per group (288 GB per GPU hold a 10M x 384 slab many times over) — and scans it for the queries of group
 * r / (ndev / query_groups), a contiguous 1/query_groups of the batch.  A shard's step has a fixed part that does not shrink with its
 * rows (sample, selections, launches): 2 groups x 4 row shards pay it over 2.5M rows and half the queries per device where 8 row
 * shards pay it over 1.25M rows and every query.  Same results as the unsharded index; ONE all-gather over all devices, one merge per
 * group.  The reference partitions rows only (scan_parallel, search.rs:1013-1036); splitting the QUERY batch as well has no
 * counterpart there because its callers issue one query at a time.  query_groups = 1 is fsgpu_sharded_create{,_device} /
 * fsgpu_sharded_open_fsvi.  _create_device_grouped: shard_rows / slabs / live per DEVICE; shard_rows[r] must equal
 * shard_rows[r % row shards]. */
fsgpu_status fsgpu_sharded_create_grouped(const int32_t *devices, uint32_t ndev, uint32_t query_groups, uint32_t dim, uint64_t nrows,
                                          const void *slab_f16_le, const uint64_t *live_bitmap, int32_t exchange, fsgpu_sharded **out);
fsgpu_status fsgpu_sharded_create_device_grouped(const int32_t *devices, uint32_t ndev, uint32_t query_groups, uint32_t dim,
                                                 const uint64_t *shard_rows, const void *const *shard_slabs_dev,
                                                 const uint64_t *const *shard_live_dev, int32_t exchange, fsgpu_sharded **out);
uint32_t fsgpu_sharded_query_groups(const fsgpu_sharded *idx);
uint32_t fsgpu_sharded_row_shards(const fsgpu_sharded *idx);
void fsgpu_sharded_destroy(fsgpu_sharded *idx);
uint64_t fsgpu_sharded_record_count(const fsgpu_sharded *idx);
uint32_t fsgpu_sharded_dimension(const fsgpu_sharded *idx);
uint32_t fsgpu_sharded_shard_count(const fsgpu_sharded *idx); /* devices: query groups x row shards */
int32_t fsgpu_sharded_exchange_mode(const fsgpu_sharded *idx); /* FSGPU_EXCHANGE_RCCL or FSGPU_EXCHANGE_PEER_COPY, as chosen */
int32_t fsgpu_sharded_device(const fsgpu_sharded *idx, uint32_t shard); /* the HIP device of a shard (shard 0 = the root: merges, takes queries_dev) */
fsgpu_status fsgpu_sharded_shard_range(const fsgpu_sharded *idx, uint32_t shard, uint64_t *row_lo, uint64_t *row_hi);
fsgpu_status fsgpu_sharded_set_hreduce(fsgpu_sharded *idx, int32_t mode);
/* VectorIndex::search_top_k(query, limit, None) for nq host queries over all shards (exact kernels); outputs as
 * fsgpu_search_topk.  k <= 256 and dim % 8 == 0 (the packed lists of the fused tiers are what travels). */
fsgpu_status fsgpu_sharded_search_topk(fsgpu_sharded *idx, const float *queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                       uint32_t *out_rows, float *out_scores, uint32_t *out_counts);
/* Same answers through the matrix-core batched path of every shard (fsgpu_search_topk_batched); *out_fallbacks (optional) sums
 * the per-shard exact fallbacks. */
fsgpu_status fsgpu_sharded_search_topk_batched(fsgpu_sharded *idx, const float *queries, uint32_t nq, uint32_t query_len,
                                               uint32_t k, uint32_t *out_rows, float *out_scores, uint32_t *out_counts,
                                               uint32_t *out_fallbacks);
/* The general form: VectorIndex::search_top_k(query, limit, filter) (search.rs:192-206) and the two-pass searches
 * (search.rs:514-661, 876-946) on one object.
 *   mode  FSGPU_SHARDED_EXACT / _BATCHED: the exact kernels / the matrix-core batched path of every shard; allow_bitmap (optional)
 *         is the index-wide precomputed SearchFilter (bit r = global row r may be returned, filter.rs:19-56), split per shard
 *         like the live bitmap.
 *         FSGPU_SHARDED_INT8_TWO_PASS / _4BIT_TWO_PASS: search_top_k_int8_two_pass / search_top_k_4bit_two_pass for the batch.
 *         Every shard quantises with the ONE corpus-wide scale of the reference (simd.rs:1865-1886: max-abs over the whole slab —
 *         an ncclAllReduce(max) of 4 bytes over the shards, once per index) and hands its k*candidate_multiplier pass-1
 *         candidates to the root as (pass-1 entry, exact entry) pairs; the root takes the corpus-wide k*candidate_multiplier
 *         best by the pass-1 order — exactly the unsharded candidate set — and the k best of those by the exact order.
 *         k*candidate_multiplier <= 256, shards*k*candidate_multiplier <= 1024, no filter.
 * fsgpu_sharded_search = begin + end, both on the calling thread (no worker threads inside the handle): begin ENQUEUES every
 * shard's scan on that shard's stream — the batched and two-pass searches through their begin halves, so no host wait sits between
 * one search's last kernel and the next one's first — and the exchange + merge behind them; end waits for that one search, reads the
 * shards' verdicts and, only if a shard had to answer an uncertified query late, sends the corrected lists through the exchange again.
 * Two searches may be in flight per handle (end them in order), so the exchange and merge of one run underneath the scan of the next.
 * A LONE query (nq = 1, host pointer, no filter, mode EXACT or a two-pass mode) takes the shards' latency lanes instead (certified
 * int8 pass with fsgpu_sharded_set_int8_latency, two-pass lane): every shard of one query group answers into its own pinned block,
 * all begun before any is waited for, and the calling thread merges the short lists — merge_partial_heaps on the host, where the
 * reference runs it (search.rs:1704-1720); no collective, no merge launch, no D2H copy. */
#define FSGPU_SHARDED_EXACT 0
#define FSGPU_SHARDED_BATCHED 1
#define FSGPU_SHARDED_INT8_TWO_PASS 2
#define FSGPU_SHARDED_4BIT_TWO_PASS 3
typedef struct fsgpu_sharded_request {
    const float *queries;          /* [nq, query_len] host */
    uint32_t nq, query_len, k;
    int32_t mode;
    uint32_t candidate_multiplier; /* two-pass modes (0 counts as 1, as in the reference) */
    const uint64_t *allow_bitmap;  /* ceil(N/64) words or NULL */
    const float *queries_dev;      /* NULL, or the queries already RESIDENT on devices[0] (an encoder's device output,
                                    * fsgpu_bert_embed_device): `queries` is then ignored — no staging, no H2D copy; the other shards
                                    * fetch them from the root peer to peer (xGMI).  Must stay unchanged until the search has ended. */
} fsgpu_sharded_request;
fsgpu_status fsgpu_sharded_search(fsgpu_sharded *idx, const fsgpu_sharded_request *request, uint32_t *out_rows, float *out_scores,
                                  uint32_t *out_counts, uint32_t *out_fallbacks);
/* The batch's queries resident in PARTS on several devices — data-parallel encoders ("Encoders: data-parallel over the
 * query batch"): part p holds part_counts[p] consecutive queries on device part_devices[p] (fsgpu_bert_embed_device of that device's
 * encoder); request->queries / queries_dev are ignored, the counts add up to request->nq.  Every device fetches the slice of its query
 * group peer to peer over xGMI (in place when the slice is one part on that very device).  Blocking (begin + end). */
fsgpu_status fsgpu_sharded_search_parts(fsgpu_sharded *idx, const fsgpu_sharded_request *request, const float *const *parts_dev,
                                        const uint32_t *part_counts, const int32_t *part_devices, uint32_t n_parts, uint32_t *out_rows,
                                        float *out_scores, uint32_t *out_counts, uint32_t *out_fallbacks);
/* fsgpu_index_set_int8_latency for every shard: lone EXACT queries take ONE certified pass over the shard's int8 copy (built here,
 * from the corpus-wide scale) + exact re-score instead of the exact kernel's pass over the f16 rows — same rows and score bits, half
 * the bytes.  Off by default. */
fsgpu_status fsgpu_sharded_set_int8_latency(fsgpu_sharded *idx, int32_t enabled);
fsgpu_status fsgpu_sharded_search_begin(fsgpu_sharded *idx, const fsgpu_sharded_request *request, uint64_t *out_ticket);
fsgpu_status fsgpu_sharded_search_end(fsgpu_sharded *idx, uint64_t ticket, uint32_t *out_rows, float *out_scores,
                                      uint32_t *out_counts, uint32_t *out_fallbacks);
/* Dynamic batching of concurrent callers on a sharded handle (as fsgpu_index_set_coalescing): fsgpu_sharded_search calls with
 * nq = 1, k <= 64, no allow bitmap and mode EXACT or INT8_TWO_PASS that are in flight together ride ONE search of the shards (more
 * than four exact callers take the BATCHED mode — the same rows and score bits; two-pass callers one two-pass batch with their
 * multiplier).  max_batch = 0 turns it off (the default). */
fsgpu_status fsgpu_sharded_set_coalescing(fsgpu_sharded *idx, uint32_t max_batch, uint32_t max_wait_us);
fsgpu_status fsgpu_sharded_coalescing_stats(fsgpu_sharded *idx, uint64_t *batches, uint64_t *requests);
/* the corpus-wide max-abs the shards' int8 / 4-bit copies are built from (0 before the first two-pass search) */
float fsgpu_sharded_quant_scale_max(const fsgpu_sharded *idx);
/* VectorIndex::open (lib.rs:1747-1909) of an FSVI v1 file with an F16 slab, rows split over the devices.  The handle keeps the
 * record table, the doc-id strings and the tombstone flags, so the doc-id level calls below work as on fsgpu_index. */
fsgpu_status fsgpu_sharded_open_fsvi(const char *path, const int32_t *devices, uint32_t ndev, int32_t exchange, fsgpu_sharded **out);
fsgpu_status fsgpu_sharded_open_fsvi_grouped(const char *path, const int32_t *devices, uint32_t ndev, uint32_t query_groups,
                                             int32_t exchange, fsgpu_sharded **out);
/* index-wide tombstone bitmap (bit r = global row r is live; NULL = all live), split per shard */
fsgpu_status fsgpu_sharded_set_live_bitmap(fsgpu_sharded *idx, const uint64_t *live_bitmap);
/* soft_delete / append / doc ids / search_top_k with the resident WAL, shadowing and doc-id dedup (fsgpu_index_soft_delete,
 * fsgpu_index_wal_append, fsgpu_index_doc_id, fsgpu_search_hits): handles opened with fsgpu_sharded_open_fsvi */
fsgpu_status fsgpu_sharded_soft_delete(fsgpu_sharded *idx, const char *doc_id, uint32_t doc_id_len, int32_t *out_deleted);
fsgpu_status fsgpu_sharded_wal_append(fsgpu_sharded *idx, const char *doc_id, uint32_t doc_id_len, const float *vector,
                                      uint32_t vector_len);
uint64_t fsgpu_sharded_wal_record_count(const fsgpu_sharded *idx);
fsgpu_status fsgpu_sharded_doc_id(const fsgpu_sharded *idx, uint32_t row, const char **out_ptr, uint32_t *out_len);
fsgpu_status fsgpu_sharded_search_hits(fsgpu_sharded *idx, const float *query, uint32_t query_len, uint32_t k, uint32_t *out_rows,
                                       float *out_scores, uint32_t *out_count);
/* dot_query_at over global row ids, each routed to the shard that owns it (fsgpu_gather_dot) */
fsgpu_status fsgpu_sharded_gather_dot(fsgpu_sharded *idx, const float *query, uint32_t query_len, const uint32_t *rows, uint32_t n,
                                      float *out_scores);

/* ---- index build helpers ---- */
/* VectorIndexWriter::write_record + finish for FSVI v1 (crates/frankensearch-index/src/lib.rs:3637-3672, 3752-3943): every
 * vector must be finite with a usable norm and every doc id fit in u16 bytes (else FSGPU_ERR_INVALID_CONFIG, nothing
 * written); records are stably sorted by (FNV-1a(doc_id), doc_id) and written as header (CRC32) | 16-byte records | string
 * table | pad to 64 | little-endian f16 slab (f32 -> f16 round-to-nearest-even on `device`).  The bytes equal the
 * reference writer's for the same input.  doc_id_lens may be NULL (NUL-terminated ids).  n may exceed the GPU grid limit
 * only in theory (n < 2^31). */
fsgpu_status fsgpu_fsvi_write(const char *path, const char *embedder_id, const char *embedder_revision, uint32_t dim,
                              uint64_t n, const char *const *doc_ids, const uint32_t *doc_id_lens, const float *vectors,
                              uint8_t compaction_gen, int32_t device);

/* The same writer for either Quantization (lib.rs:203-208): quantization 1 = F16 (as above), 0 = F32 (rows stored as raw
 * little-endian f32, write_vector_slab lib.rs:6017-6024).  F32 files open and search like F16 ones (dot_product_f32_bytes_f32,
 * simd.rs:581-702), through the general path only: F16 is the reference's default and the accelerated format. */
fsgpu_status fsgpu_fsvi_write_quant(const char *path, const char *embedder_id, const char *embedder_revision, uint32_t dim,
                                    uint64_t n, const char *const *doc_ids, const uint32_t *doc_id_lens,
                                    const float *vectors, uint8_t compaction_gen, int32_t device, uint8_t quantization);

/* encode_f32_to_f16_extend (simd.rs:2245-2305): f32 -> f16 round-to-nearest-even on the GPU. */
fsgpu_status fsgpu_encode_f32_to_f16(int32_t device, const float *src, uint64_t n, uint16_t *dst);
/* f16 -> f32 widen (simd.rs:63-94), exposed for the exhaustive 65,536-pattern parity test. */
fsgpu_status fsgpu_widen_f16_to_f32(int32_t device, const uint16_t *src, uint64_t n, float *dst);

/* ---- Model2Vec (potion) ---- */
/* Model2VecEmbedder (embed/src/model2vec_embedder.rs:55-58): table is [vocab,dim] f32 row-major (host). */
fsgpu_status fsgpu_m2v_create(int32_t device, const float *table, uint32_t vocab, uint32_t dim, fsgpu_m2v **out);
void fsgpu_m2v_destroy(fsgpu_m2v *m);
uint32_t fsgpu_m2v_dimension(const fsgpu_m2v *m); /* Embedder::dimension (crates/frankensearch-core/src/traits.rs:220-370) */
/* embed_batch_sync over token ids (model2vec_embedder.rs:310-335,409-419,435-451): text i owns
 * ids[offsets[i]..offsets[i+1]); out is [n,dim].  Empty / all-OOV texts give zeros. */
fsgpu_status fsgpu_m2v_embed(fsgpu_m2v *m, const uint32_t *ids, const uint32_t *offsets, uint32_t n, float *out);

/* ---- MiniLM-class BERT embedder ---- */
/* NativeEmbedder::load (native_embedder.rs:60-116): copies the weights to the GPU (linears as f16). */
fsgpu_status fsgpu_bert_create(int32_t device, const fsgpu_bert_config *config, const fsgpu_bert_weights *weights,
                               fsgpu_bert **out);
/* The same from the model file itself — NativeEmbedder::load -> parse_weights (native.rs:1359-1602): `blob` is a safetensors file
 * (8-byte little-endian header length, JSON header, tensor bytes) in HuggingFace key layout.  F32 tensors only are read (I64
 * position_ids etc. are skipped); bare `embeddings.*` / `encoder.*` keys count as `bert.`-prefixed; pooler / classifier tensors
 * are ignored.  The model's shape comes from the tensors (vocab x hidden, layers by counting, inter, heads = hidden / 32, max_pos =
 * min(rows of the position table, 512)); ln_eps <= 0 means 1e-12.  A malformed blob / a missing or mis-shaped tensor is
 * FSGPU_ERR_MODEL_LOAD_FAILED with the reference's wording in fsgpu_last_error(); the blob is validated BEFORE a device is looked
 * for.  The tensor data must be 4-byte aligned in `blob` (it is when the file is read into a buffer malloc returned: the header
 * is padded to 8 bytes). */
fsgpu_status fsgpu_bert_create_safetensors(int32_t device, const void *blob, uint64_t blob_len, float ln_eps, fsgpu_bert **out);
void fsgpu_bert_destroy(fsgpu_bert *m);
uint32_t fsgpu_bert_dimension(const fsgpu_bert *m); /* Embedder::dimension: the model's hidden size */
/* embed_batch_sync over token ids (native_embedder.rs:218-255 -> Model::embed_forward native.rs:1142-1236):
 * text i owns ids[offsets[i]..offsets[i+1]) (already tokenised WITH special tokens and truncated to <= 512,
 * no padding); every returned token is mean-pooled, then L2-normalised (zeros for empty / zero-norm,
 * fastembed_embedder.rs:416-426).  out is [n, hidden]. */
fsgpu_status fsgpu_bert_embed(fsgpu_bert *m, const int32_t *ids, const uint32_t *offsets, uint32_t n, float *out);

/* ---- device-resident hand-offs: encoder -> search without the vectors crossing PCIe ---- */
/* The reference's seam between the two is a host Vec<f32> (traits.rs:401-582 -> search.rs:192); on the GPU both ends live in HBM.
 * fsgpu_bert_embed_device / fsgpu_m2v_embed_device are fsgpu_bert_embed / fsgpu_m2v_embed with the [n, dim] output left in device
 * memory `out_dev` (on the embedder's device; complete when the call returns).  It feeds fsgpu_search_topk_device /
 * fsgpu_search_topk_batched_device, fsgpu_search_topk_batched_device_queries (device queries, host results: what a serving loop
 * wants), or a sharded search through fsgpu_sharded_request::queries_dev.  fsgpu_device_malloc / _free give a host that links
 * nothing but this library the buffers for it. */
fsgpu_status fsgpu_bert_embed_device(fsgpu_bert *m, const int32_t *ids, const uint32_t *offsets, uint32_t n, float *out_dev);
fsgpu_status fsgpu_m2v_embed_device(fsgpu_m2v *m, const uint32_t *ids, const uint32_t *offsets, uint32_t n, float *out_dev);
fsgpu_status fsgpu_search_topk_batched_device_queries(fsgpu_index *idx, const float *queries_dev, uint32_t nq, uint32_t query_len,
                                                      uint32_t k, uint32_t *out_rows, float *out_scores, uint32_t *out_counts,
                                                      uint32_t *out_fallbacks);
/* fsgpu_search_topk_int8_two_pass_batched (search.rs:514-661 for a batch) with the queries already in device memory — the fast tier
 * of a many-queries two-tier flow takes the Model2Vec batch where fsgpu_m2v_embed_device left it (sync_searcher.rs:652-700 runs
 * embed -> search_fast_hits back to back; here neither the vectors nor a wait crosses PCIe in between). */
fsgpu_status fsgpu_search_topk_int8_two_pass_batched_device_queries(fsgpu_index *idx, const float *queries_dev, uint32_t nq,
                                                                    uint32_t query_len, uint32_t k, uint32_t candidate_multiplier,
                                                                    uint32_t *out_rows, float *out_scores, uint32_t *out_counts,
                                                                    uint32_t *out_fallbacks);
fsgpu_status fsgpu_device_malloc(int32_t device, uint64_t bytes, void **out);
fsgpu_status fsgpu_device_free(int32_t device, void *ptr);
int32_t fsgpu_bert_device(const fsgpu_bert *m);
int32_t fsgpu_m2v_device(const fsgpu_m2v *m);
int32_t fsgpu_index_device(const fsgpu_index *idx);

/* ---- host-side rank fusion (O(k), CPU, no GPU needed) ---- */
/* One ranked hit: ScoredResult / VectorHit as the fusion code reads them
 * (crates/frankensearch-core/src/types.rs:88-134): doc id (not NUL-terminated), score, vector row index. */
typedef struct fsgpu_scored_doc {
    const char *doc_id;
    uint32_t doc_id_len;
    float score;
    uint32_t index;
} fsgpu_scored_doc;
/* FusedHit (types.rs:3892-3925); ranks are -1 when absent, semantic_index 0xffffffff when absent;
 * doc_id points into the caller's input arrays. */
typedef struct fsgpu_fused_hit {
    const char *doc_id;
    uint32_t doc_id_len;
    double rrf_score;
    int64_t lexical_rank, semantic_rank;
    uint32_t semantic_index;
    float lexical_score, semantic_score;
    uint8_t in_both_sources;
} fsgpu_fused_hit;
#define FSGPU_RRF_TIEBREAK_LEXICAL_THEN_ID 0 /* RrfTiebreak::LexicalThenId (default, rrf.rs:52-66) */
#define FSGPU_RRF_TIEBREAK_HASH 1            /* RrfTiebreak::Hash */
/* rrf_fuse (crates/frankensearch-fusion/src/rrf.rs:368-560): score = sum over lanes of weight/(k+rank+1);
 * order (rrf desc, in_both desc, lexical score desc | doc-id hash, doc_id asc); window = offset..offset+limit.
 * Non-finite / negative k -> 60; non-finite / non-positive weights -> 1.  out holds `limit` entries. */
fsgpu_status fsgpu_rrf_fuse(const fsgpu_scored_doc *lexical, uint32_t n_lexical, const fsgpu_scored_doc *semantic,
                            uint32_t n_semantic, double k, double lexical_weight, double semantic_weight,
                            int32_t tiebreak, uint32_t limit, uint32_t offset, fsgpu_fused_hit *out,
                            uint32_t *out_count);
/* blend_two_tier (crates/frankensearch-fusion/src/blend.rs:107-195): per-list min-max normalisation,
 * alpha*quality + (1-alpha)*fast (single-source docs keep their normalised score), order (score desc, doc_id asc).
 * out holds n_fast + n_quality entries. */
fsgpu_status fsgpu_blend_two_tier(const fsgpu_scored_doc *fast, uint32_t n_fast, const fsgpu_scored_doc *quality,
                                  uint32_t n_quality, float blend_factor, fsgpu_scored_doc *out, uint32_t *out_count);
/* blend_two_tier_aligned (blend.rs:213-294; the vector-index specialisation :296-358 gives the same output for unique doc ids):
 * the quality tier as per-position scores of the SAME hits — quality_scores[i] belongs to fast[i] and counts only where
 * quality_present[i] != 0 (the Option<f32> of quality_scores_for_hits).  Fast bounds over all fast scores, quality bounds over
 * the present scores, first occurrence of a doc id wins both slots.  out holds n_fast entries. */
fsgpu_status fsgpu_blend_two_tier_aligned(const fsgpu_scored_doc *fast, uint32_t n_fast, const float *quality_scores,
                                          const uint8_t *quality_present, float blend_factor, fsgpu_scored_doc *out,
                                          uint32_t *out_count);

/* ---- TwoTierIndex: the pairing of a fast and a quality index (crates/frankensearch-index/src/two_tier.rs) ---- */
/* QualityAlignment (two_tier.rs:404-409), computed once when the pair is opened (two_tier.rs:750-866): both record tables are
 * sorted by (FNV-1a(doc_id), doc_id), so one merge pass — tombstoned rows skipped on either side — maps every fast row to its
 * quality row; the result stays ALIGNED (fast row i = quality row i) until the first divergence and becomes a per-row MAPPING
 * after it.  Raw slabs (no record table) pair by row.  The indexes stay owned by the caller and must outlive the alignment;
 * rebuild it after a tombstone update of either index (the reference computes it at open). */
typedef struct fsgpu_alignment fsgpu_alignment;
#define FSGPU_ALIGNMENT_NONE 0
#define FSGPU_ALIGNMENT_ALIGNED 1
#define FSGPU_ALIGNMENT_MAPPING 2
fsgpu_status fsgpu_alignment_create(fsgpu_index *fast, fsgpu_index *quality, fsgpu_alignment **out);
void fsgpu_alignment_destroy(fsgpu_alignment *a);
int32_t fsgpu_alignment_kind(const fsgpu_alignment *a);
/* quality_index_for_fast_index (two_tier.rs:1975-1981): -1 when the fast row has no quality row */
int64_t fsgpu_alignment_quality_row(const fsgpu_alignment *a, uint64_t fast_row);
uint64_t fsgpu_alignment_unmatched_quality_docs(const fsgpu_alignment *a);
/* TwoTierIndex::quality_scores_for_hits (two_tier.rs:1566-1631): the phase-2 scoring of an UNATTESTED quality tier — every
 * FSVI v1 artifact (sync_searcher.rs:810-818).  Per fast hit: the quality WAL's latest resident entry of the doc id
 * (dot_product_f32_f32 on the host), else the aligned quality row (hit.index == 0xffffffff looks the fast row up by doc id
 * first), else the quality index's own live row of that doc id; main rows are scored by ONE gather launch of
 * dot_product_f16_bytes_f32 in the scan's operation order (dot_query_at).  out_present[i] == 0 is the reference's None.
 * hits[i].doc_id may be null for raw-slab pairs.  FSGPU_ERR_DIMENSION_MISMATCH when query_len is not the quality dimension. */
fsgpu_status fsgpu_quality_scores_for_hits(fsgpu_index *fast, fsgpu_index *quality, const fsgpu_alignment *alignment,
                                           const float *query, uint32_t query_len, const fsgpu_scored_doc *hits, uint32_t n,
                                           float *out_scores, uint8_t *out_present);
/* quality_scores_for_hits for a CHUNK of queries (the many-queries two-tier flow, fshost_two_tier_search_many with
 * FSHOST_POOL_RESCORED): query q owns hits[hit_offsets[q] .. hit_offsets[q + 1]) and queries[q * query_len ..); every hit is
 * resolved as above (two_tier.rs:1566-1631), and the dots over the quality tier's main rows of ALL the queries run as ONE gather
 * launch.  out_scores / out_present are aligned with `hits`.  Same scores, bit for bit, as nq calls of the function above. */
fsgpu_status fsgpu_quality_scores_for_hits_batched(fsgpu_index *fast, fsgpu_index *quality, const fsgpu_alignment *alignment,
                                                   const float *queries, uint32_t nq, uint32_t query_len,
                                                   const fsgpu_scored_doc *hits, const uint32_t *hit_offsets, float *out_scores,
                                                   uint8_t *out_present);

/* TwoTierIndex over two row-sharded handles (fast and quality slabs shard identically).  The alignment walk reads the
 * handles' catalogs (fsgpu_sharded_open_fsvi; raw shards pair by row); quality_scores_for_hits resolves WAL entries and doc ids
 * there and gathers dot_query_at on the shards that own the quality rows (one gather launch per shard touched).  Same outputs
 * as fsgpu_quality_scores_for_hits over unsharded indexes with the same rows. */
fsgpu_status fsgpu_sharded_alignment_create(fsgpu_sharded *fast, fsgpu_sharded *quality, fsgpu_alignment **out);
fsgpu_status fsgpu_sharded_quality_scores_for_hits(fsgpu_sharded *fast, fsgpu_sharded *quality, const fsgpu_alignment *alignment,
                                                   const float *query, uint32_t query_len, const fsgpu_scored_doc *hits, uint32_t n,
                                                   float *out_scores, uint8_t *out_present);

/* ---- MRL: truncated scan + full-dimension rescore ---- */
/* MrlSearchStats (crates/frankensearch-index/src/mrl.rs:122-139). */
typedef struct fsgpu_mrl_stats {
    uint32_t scan_dims, rescore_dims, candidates_rescored;
    uint64_t records_scanned;
    int32_t fell_back_to_full;
} fsgpu_mrl_stats;
/* VectorIndex::mrl_search_with_stats (mrl.rs:241-395) with MrlConfig{search_dims, rescore_dims, rescore_top_k}
 * (:55-115; 0 = full dimension / 3*k): phase 1 scores only the first search_dims dimensions of every live row (the
 * kernel reads that prefix of each row: N*search_dims*2 bytes instead of N*dim*2) and keeps the top rescore_top_k,
 * resident WAL entries join with their truncated f32 dot; phase 2 re-scores the candidates over rescore_dims and
 * returns the best k, best first (WAL hits at the virtual index record_count + i; no doc-id dedup, as the reference).
 * search_dims >= dimension falls back to the standard search; search_dims == 0 is FSGPU_ERR_INVALID_CONFIG.
 * out_rows / out_scores hold k entries; stats may be NULL. */
fsgpu_status fsgpu_search_mrl(fsgpu_index *idx, const float *query, uint32_t query_len, uint32_t k, uint32_t search_dims,
                              uint32_t rescore_dims, uint32_t rescore_top_k, uint32_t *out_rows, float *out_scores,
                              uint32_t *out_count, fsgpu_mrl_stats *stats);

/* fsgpu_search_mrl for nq queries at once (row-level results, no stats): the truncated scan of phase 1 runs on the matrix cores
 * for the whole batch (mfma_scan.hip / mfma_wide.hip over the strided prefix of every row: N*search_dims*2 bytes per 256-384
 * queries), phase 2 re-scores every query's rescore_top_k candidates in one launch.  Outputs as fsgpu_search_topk ([nq, k]
 * rows / scores, [nq] counts); hits equal fsgpu_search_mrl's for each query.  Resident WAL entries, rescore_top_k > 64,
 * k > 64 or unsupported search_dims (not 64 / 128 / 256) are answered query by query (*out_fallbacks, may be NULL). */
fsgpu_status fsgpu_search_mrl_batched(fsgpu_index *idx, const float *queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                      uint32_t search_dims, uint32_t rescore_dims, uint32_t rescore_top_k, uint32_t *out_rows,
                                      float *out_scores, uint32_t *out_counts, uint32_t *out_fallbacks);

/* ---- dynamic batching of concurrent callers ---- */
/* The reference's seams are per-query calls made by many host threads at once (VectorIndex::search_top_k takes &self,
 * crates/frankensearch-index/src/search.rs:192; SyncEmbed::embed_sync, crates/frankensearch-core/src/traits.rs:401-582;
 * the MiniLM backends serialise callers on a mutex, crates/frankensearch-rerank/src/native_embedder.rs:40-50).  With
 * coalescing enabled, single-item calls that are in flight together (fsgpu_search_topk with nq = 1 and k <= 64 — filtered
 * callers share a batch when they pass the SAME allow bitmap, i.e. the same pointer; fsgpu_m2v_embed / fsgpu_bert_embed
 * with n = 1) are gathered into one batched launch: up to max_batch
 * items, waiting at most max_wait_us for the batch to fill.  Results are bit-identical to the unbatched calls.
 * max_batch = 0 turns it off (the default).  No threads are created: the first waiting caller runs the batch. */
fsgpu_status fsgpu_index_set_coalescing(fsgpu_index *idx, uint32_t max_batch, uint32_t max_wait_us);
fsgpu_status fsgpu_index_coalescing_stats(fsgpu_index *idx, uint64_t *batches, uint64_t *requests);
fsgpu_status fsgpu_m2v_set_coalescing(fsgpu_m2v *m, uint32_t max_batch, uint32_t max_wait_us);
fsgpu_status fsgpu_bert_set_coalescing(fsgpu_bert *m, uint32_t max_batch, uint32_t max_wait_us);

/* Environment switches (read once, at first use).
 * A default build reads five:
 *   FSGPU_WIDE=0|2|3            batched main pass: 0 = queries in LDS (128 per pass), 2 / 3 = queries in registers (256 / 384 on f16 rows)
 *   FSGPU_FILTER=f16|i8         pin the candidate filter of the exact batched search (as fsgpu_index_set_batched_filter does per index)
 *   FSGPU_DEBUG_BATCHED         one line per batched search on stderr (fallback census)
 *   FSGPU_BERT_NO_GRAPH         the query-sized encoder calls launch eagerly instead of replaying a captured hipGraph
 *   FSGPU_DEBUG_GRAPH           say why a graph capture failed
 * Everything else is a tuning / A-B switch of the lab and exists only in builds with -DFSGPU_EXPERIMENTS
 * (FSGPU_BUILD_DEFS="-DFSGPU_EXPERIMENTS" python -m frankensearch_amd.build; scripts/exp_*, scripts/r03/): FSGPU_WIDE_OPT and
 * FSGPU_WIDE_DBG (options and timing skeletons of the wide main pass — skeleton answers are NOT valid), FSGPU_USE_160,
 * FSGPU_MFMA_SHAPE{,_I8}, FSGPU_RA, FSGPU_RB, FSGPU_ROUND, FSGPU_NO_SKIP_B, FSGPU_NO_REVERSE, FSGPU_I8F_GROWTH, FSGPU_WIDE_MAX,
 * FSGPU_SLOTS_B, FSGPU_SLOTS_MAIN, FSGPU_NO_WIDE_B, FSGPU_NO_ANCHOR, FSGPU_NO_BIG_POOL, FSGPU_NO_HEUR_B, FSGPU_HEUR_RANK,
 * FSGPU_RB_PCT, FSGPU_NO_GROUP_SAMPLE, FSGPU_WIDE_OPT, FSGPU_WIDE_DBG,
 * FSGPU_GRID_BLOCKS, FSGPU_I8_PER_CU, FSGPU_SELECT_SORT_ABOVE, FSGPU_BERT_GEMM_SHAPE, FSGPU_BERT_ATTN, FSGPU_BERT_NO_FUSED_LN,
 * FSGPU_BERT_NO_QUERY_PATH, FSGPU_BERT_GEMM_V1, FSGPU_BERT_SPLIT_FFN, FSGPU_BERT_SPLIT_AO, FSGPU_BERT_PACKED_MIN_TOKENS,
 * FSGPU_BERT_EMBED_V1. */

/* Bench fixtures, kernel timers and A/B switches (fsgpu_bench_fixture_device, fsgpu_index_set_profiling / _scan_time / _scan_stats /
 * _filter_stats, fsgpu_last_main_pass_kernel, fsgpu_index_set_variant) are not part of the drop-in surface: include/fsgpu_lab.h. */

#ifdef __cplusplus
}
#endif
#endif /* FSGPU_H */
