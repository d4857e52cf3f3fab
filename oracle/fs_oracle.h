/*
 * fs_oracle.h — CPU restatement ("oracle") of frankensearch's semantic hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product (frankensearch_amd/,
 * libfsgpu.so) links, imports or calls this file.  Only tests/, the smoke check in
 * __graft_entry__.smoke() and the `cpu_baseline` leg of bench.py may use it, and only
 * as the checker / the CPU comparison baseline.
 *
 * The reference (Dicklesworthstone/frankensearch) is a Rust workspace that cannot be
 * built in this environment (no cargo/rustc, crates.io dependencies), so this is a
 * from-scratch C restatement of the algorithm, every function citing the reference
 * file:line it follows (paths relative to /root/reference/).
 *
 * Parity pin status:
 *   - PINNED on the known-answer / invariant tests the reference itself carries for
 *     this path (restated in tests/test_oracle_*.py): FNV-1a offsets, f16 widen over
 *     all 65,536 patterns, f32->f16 RNE vs IEEE (numpy + F16C), scan ordering / ties /
 *     NaN / k>N / collect-all / tombstones, FSVI v1 byte layout.
 *   - NOT pinned: the horizontal order of `wide::f32x8::reduce_add` (third-party crate
 *     `wide` 1.6.1, not vendored).  Two orders are provided (FSO_HREDUCE_SSE2 — the
 *     order of the crate's SSE2 build, the reference's default build — and
 *     FSO_HREDUCE_AVX); scores can differ by 1-2 ulp between them.  Against real Rust
 *     output the score contract therefore stays the north-star's 1e-3, while GPU vs
 *     oracle is held to bit-exact.
 */
#ifndef FS_ORACLE_H
#define FS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Horizontal-reduce order of the final 8-lane vector (see header comment). */
#define FSO_HREDUCE_SSE2 0 /* ((v0+v2)+(v1+v3)) + ((v4+v6)+(v5+v7)) */
#define FSO_HREDUCE_AVX 1  /* ((v0+v4)+(v2+v6)) + ((v1+v5)+(v3+v7)) */
#define FSO_HREDUCE_SEQ 2  /* (((v0+v1)+v2)+v3) + (((v4+v5)+v6)+v7): f32x8 = a.reduce_add() + b.reduce_add(), sequential f32x4 */

/* Status codes mirror SearchError variants (crates/frankensearch-core/src/error.rs:57-176). */
#define FSO_OK 0
#define FSO_ERR_DIMENSION_MISMATCH 1
#define FSO_ERR_INVALID_CONFIG 2
#define FSO_ERR_INDEX_CORRUPTED 3
#define FSO_ERR_INDEX_VERSION_MISMATCH 4
#define FSO_ERR_IO 5

/* ---- f16 <-> f32 (half 2.7.1 semantics; simd.rs:63-94, 2245-2305) ---- */
float fso_f16_to_f32(uint16_t h);
uint16_t fso_f32_to_f16(float f);
void fso_encode_f32_to_f16(const float *src, size_t n, uint16_t *dst);

/* ---- dot products ---- */
/* dot_product_f16_bytes_f32 (simd.rs:361-446 == 532-571), scalar emulation of the 8-lane path. */
float fso_dot_f16_f32(const uint8_t *row_le, const float *q, size_t dim, int hreduce);
/* Same arithmetic, AVX2+F16C intrinsics when the CPU has them (bit-identical). */
float fso_dot_f16_f32_fast(const uint8_t *row_le, const float *q, size_t dim, int hreduce);
int fso_has_avx2_f16c(void);

/* dot_product_f32_f32 (simd.rs:134-222): groups of 32 into four accumulators, (a0+a1)+(a2+a3), leftover
 * 8-chunks added to the sum, horizontal reduce, scalar tail with separate multiply and add. */
float fso_dot_f32_f32(const float *a, const float *b, size_t n, int hreduce);

/* ---- ordering (search.rs:91-126, 1655-1686) ---- */
/* 1 if (row_a,score_a) ranks strictly before (row_b,score_b) in best-first order. */
int fso_ranks_before(uint64_t row_a, float score_a, uint64_t row_b, float score_b);

/* ---- exact brute-force top-k over a raw f16 slab (search.rs:426-494, 1013-1036, 1257-1327, 1704-1720) ----
 * live: optional bitmap, bit r set = row r is live (tombstone flag clear); NULL = all live.
 * chunk_size/parallel_threshold/parallel_enabled: as SearchParams (search.rs:42-61).
 * nthreads: worker threads used for the chunk heaps (result is order-independent).
 * Returns number of hits written (<= k), best first. No doc-id dedup (see fso_fsvi_search). */
size_t fso_search_top_k(const uint8_t *slab, uint64_t nrows, uint32_t dim, const uint64_t *live,
                        const float *q, size_t k, size_t parallel_threshold, size_t chunk_size,
                        int parallel_enabled, int nthreads, int hreduce, uint32_t *out_rows,
                        float *out_scores);

/* dot_product_f32_bytes_f32 (simd.rs:581-702) and the scan over a Quantization::F32 slab. */
float fso_dot_f32_bytes_f32(const uint8_t *row, const float *q, size_t dim, int hreduce);
size_t fso_search_top_k_f32(const uint8_t *slab, uint64_t nrows, uint32_t dim, const uint64_t *live,
                            const float *q, size_t k, int nthreads, int hreduce, uint32_t *out_rows,
                            float *out_scores);

/* search_top_k_classified's input validation (search.rs:227-261):
 * returns FSO_OK, FSO_ERR_DIMENSION_MISMATCH or FSO_ERR_INVALID_CONFIG;
 * *zero_signal: 0 none, 1 CallerRequestedZeroK, 2 ZeroNormQuery. */
int fso_classify_query(const float *q, size_t qlen, uint32_t dim, size_t k, int *zero_signal);

/* ---- int8 two-pass (search.rs:514-661; simd.rs:757-1286,1865-1886) ---- */
/* quantize_f16_le_bytes_to_i8_generic: ONE corpus-wide max-abs scale, round() half away from zero, clamp +-127. */
void fso_quantize_slab_i8(const uint8_t *slab_f16_le, uint64_t n_values, int8_t *out);
/* quantize_i8_query (search.rs:1616-1626): the query's own max-abs scale. */
void fso_quantize_query_i8(const float *q, size_t dim, int8_t *out);
int32_t fso_dot_i8_i8(const int8_t *a, const int8_t *b, size_t n);
/* search_top_k_int8_two_pass without WAL / doc-id resolution: pass 1 keeps the top
 * candidate_count = max(min(k*mult, n), min(k, n)) rows by (int score desc [as f32 when dim > 1040], row asc),
 * pass 2 re-scores them with the exact f16 dot and selects the top k under the usual order.
 * slab_i8 is the output of fso_quantize_slab_i8 for the same slab.  Returns the hit count. */
size_t fso_search_int8_two_pass(const uint8_t *slab, const int8_t *slab_i8, uint64_t nrows, uint32_t dim,
                                const uint64_t *live, const float *q, size_t k, size_t candidate_multiplier,
                                int hreduce, uint32_t *out_rows, float *out_scores);

/* gather-dot: VectorIndex::dot_query_at (lib.rs:3229-3239) for a list of rows. */
void fso_gather_dot(const uint8_t *slab, uint32_t dim, const float *q, const uint32_t *rows,
                    size_t n, int hreduce, float *out);

/* ---- hashes / checksums (lib.rs:6115-6127) ---- */
uint64_t fso_fnv1a64(const uint8_t *bytes, size_t n);
uint32_t fso_crc32(const uint8_t *bytes, size_t n);
uint64_t fso_align_up(uint64_t value, uint64_t alignment);

/* vector_signal_usable (lib.rs:6133-6142). */
int fso_vector_signal_usable(const float *v, size_t n);
/* l2_normalize (core/src/traits.rs:590-618): zeros if norm^2 non-finite or < f32::EPSILON. */
void fso_l2_normalize(float *v, size_t n);

/* ---- FSVI v1 writer / reader (lib.rs:6-43, 3607-3943, 4049-4144, 5714-5768, 5979-6026) ---- */
typedef struct fso_fsvi fso_fsvi;

/* Writes an F16 FSVI v1 file: records sorted (stable) by (fnv1a(doc_id), doc_id). */
int fso_fsvi_write(const char *path, const char *embedder_id, const char *embedder_revision,
                   uint32_t dim, uint64_t n, const char *const *doc_ids, const float *vectors,
                   uint8_t compaction_gen);
int fso_fsvi_write_quant(const char *path, const char *embedder_id, const char *embedder_revision,
                         uint32_t dim, uint64_t n, const char *const *doc_ids, const float *vectors,
                         uint8_t compaction_gen, uint8_t quantization /* 1 = F16, 0 = F32 */);
uint8_t fso_fsvi_quantization(const fso_fsvi *idx);
int fso_fsvi_open(const char *path, fso_fsvi **out);
void fso_fsvi_close(fso_fsvi *idx);
uint64_t fso_fsvi_record_count(const fso_fsvi *idx);
uint32_t fso_fsvi_dimension(const fso_fsvi *idx);
uint64_t fso_fsvi_vectors_offset(const fso_fsvi *idx);
const uint8_t *fso_fsvi_slab(const fso_fsvi *idx);
/* doc id of physical row (not NUL-terminated); returns length. */
uint32_t fso_fsvi_doc_id(const fso_fsvi *idx, uint64_t row, const char **ptr);
uint16_t fso_fsvi_flags(const fso_fsvi *idx, uint64_t row);
void fso_fsvi_set_flags(fso_fsvi *idx, uint64_t row, uint16_t flags); /* in-memory soft delete */
/* VectorIndex::append (lib.rs:2532-2720): validates (FSO_ERR_DIMENSION_MISMATCH / FSO_ERR_INVALID_CONFIG),
 * supersedes a resident WAL entry with the same doc id, tombstones the first live main row with that doc id,
 * and makes the f32 vector immediately searchable (scan_wal, search.rs:1449-1475). */
int fso_fsvi_append(fso_fsvi *idx, const char *doc_id, const float *vector, size_t len);
/* VectorIndex::soft_delete_batch (lib.rs:2313-2397) for one id: tombstones the live main records, drops the resident WAL
 * entries; returns how many records went live -> deleted. */
size_t fso_fsvi_soft_delete(fso_fsvi *idx, const char *doc_id);
uint64_t fso_fsvi_wal_count(const fso_fsvi *idx);
/* doc id of a WAL entry (virtual row record_count + i). */
uint32_t fso_fsvi_wal_doc_id(const fso_fsvi *idx, uint64_t i, const char **ptr);
/* search_top_k + resolve_hits incl. tombstone skip and post-top-k doc-id dedup (search.rs:1493-1558). */
size_t fso_fsvi_search(const fso_fsvi *idx, const float *q, size_t k, int hreduce,
                       uint32_t *out_rows, float *out_scores);

/* ---- synthetic fixtures ---- */
/* hash-mix fixture (search.rs:1823-1834): value(i,j). */
float fso_fixture_hashmix(uint64_t i, uint64_t j);
/* xorshift raw_vector / clustered corpus (frankensearch/benches/fsvi_4bit_vs_incumbent.rs:56-101,344-365). */
void fso_raw_vector(uint64_t seed, uint32_t dim, float *out);
void fso_normalize_bench(float *v, uint32_t dim);
/* rows [row0,row0+n) of the clustered corpus as f16 LE; centroids = `clusters` normalised raw vectors. */
void fso_clustered_corpus_f16(uint64_t row0, uint64_t n, uint32_t dim, uint32_t clusters,
                              float noise, uint16_t *out);
void fso_clustered_query(uint64_t q, uint32_t dim, uint32_t clusters, float noise, float *out);

/* ---- Model2Vec (embed/src/model2vec_embedder.rs:280-335,435-451; embed/src/simd.rs:74-113) ---- */
/* ids: token ids of one text; table: [vocab,dim] f32 row-major. out: [dim]. */
void fso_m2v_embed(const float *table, uint32_t vocab, uint32_t dim, const uint32_t *ids,
                   size_t n_ids, float *out);

/* 4-bit two-pass (search.rs:860-1000; simd.rs:1286-1556, 2153-2215). */
void fso_pack_slab_4bit(const uint8_t *slab_f16, uint64_t count, uint32_t dim, uint8_t *out);
void fso_pack_query_4bit(const float *q, uint32_t dim, uint8_t *out);
int32_t fso_dot_4bit(const uint8_t *stored, const uint8_t *query, size_t nbytes);
size_t fso_search_4bit_two_pass(const uint8_t *slab, const uint8_t *slab_4bit, uint64_t nrows, uint32_t dim,
                                const uint64_t *live, const float *q, size_t k, size_t candidate_multiplier, int hreduce,
                                uint32_t *out_rows, float *out_scores);
/* mrl.rs:241-395 — truncated scan + rescore (see fs_oracle.c). */
size_t fso_mrl_search(const uint8_t *slab, uint64_t nrows, uint32_t dim, const uint64_t *live, const float *const *wal_vecs,
                      size_t wal_len, const float *q, size_t limit, size_t search_dims, size_t rescore_dims,
                      size_t rescore_top_k, int hreduce, uint32_t *out_rows, float *out_scores);

/* ---- MiniLM-class encoder, f32 (bert_oracle_c.c; crates/frankensearch-rerank/src/native.rs:587-626,1142-1236) ---- */
typedef struct {
    const float *wqkv, *bqkv;     /* [3H, H] (query, key, value stacked, native.rs:1359-1602), [3H] */
    const float *wo, *bo;         /* attention.output.dense [H, H], [H] */
    const float *ln1_w, *ln1_b;   /* attention.output.LayerNorm */
    const float *w1, *b1;         /* intermediate.dense [I, H], [I] */
    const float *w2, *b2;         /* output.dense [H, I], [H] */
    const float *ln2_w, *ln2_b;   /* output.LayerNorm */
} fso_bert_layer;
typedef struct {
    int32_t vocab, hidden, layers, inter, max_pos;
    float eps;
    const float *word, *pos, *type0, *emb_ln_w, *emb_ln_b;
    const fso_bert_layer *layer;  /* [layers] */
} fso_bert_weights;
/* ids: all documents' token ids back to back; offsets: [n_docs + 1] into ids; out: [n_docs, hidden] unit vectors (zeros for
 * an empty document).  nthreads workers share the batch in blocks of ~128 tokens.  0 on success. */
int fso_bert_forward(const fso_bert_weights *w, const int32_t *ids, const uint32_t *offsets, uint32_t n_docs, int nthreads,
                     float *out);

#ifdef __cplusplus
}
#endif
#endif /* FS_ORACLE_H */
