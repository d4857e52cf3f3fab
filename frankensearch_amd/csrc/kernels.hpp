// kernels.hpp — argument blocks and host launchers of the fsgpu HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace fsgpu {

typedef unsigned long long u64;

struct ScanArgs {
    const void* slab;       // [nrows, dim] little-endian f16, row-major (FSVI slab, lib.rs:36-41)
    const u64* live;        // bit r set = row r live (tombstone flag clear); nullptr = all live
    const u64* allow;       // per-call filter bitmap; nullptr = no filter
    const float* queries;   // [nq, dim] f32 (device)
    u64* partial;           // [nq, grid, k] packed best-first lists (device)
    uint32_t nrows;
    uint32_t dim;
    uint32_t k;
    uint32_t row_base;      // added to local rows (shard offset)
    int32_t hreduce;        // FSGPU_HREDUCE_*
    uint32_t row_stride;    // bytes between rows (>= dim*2); the MRL truncated scan reads a prefix of every row
};

struct MergeArgs {
    const u64* lists;       // packed entries; element (q, l, i) at lists[q*q_stride + l*l_stride + i]
    uint64_t q_stride;
    uint64_t l_stride;
    uint32_t nlists;
    uint32_t list_len;
    uint32_t k;             // entries to select per query
    uint32_t out_stride;    // row stride of the outputs (>= k); slots beyond the count are padded
    uint32_t* out_rows;     // [nq, out_stride]
    float* out_scores;      // [nq, out_stride]
    uint32_t* out_counts;   // [nq]
    u64* out_packed;        // optional [nq, out_stride] packed copy of the result (kEmpty padded)
    uint32_t lists_sorted = 1;  // 0: the lists are NOT best-first (disables the head/tail pruning bounds)
};

size_t scan_lds_bytes(int dim, int nq, int kcap);
int scan_occupancy_blocks_per_cu(int dim, int nq, int kcap, bool force_runtime_dim);
hipError_t launch_scan_topk(const ScanArgs& args, int nq, int kcap, int grid, hipStream_t stream,
                            bool force_runtime_dim, bool plain_loads);
// scan_mq_kernel.hip: 4 / 8 queries per pass (dim 128/256/384); occupancy != nullptr only queries residency.
bool scan_mq_supported(int dim, int nq, int kcap);
hipError_t launch_scan_mq(const ScanArgs& args, int nq, int kcap, int grid, hipStream_t stream, int* occupancy);
hipError_t launch_merge_topk(const MergeArgs& args, int nq, hipStream_t stream);
// one query read from HOST memory at launch time (it travels in the kernel's argument block: no H2D copy); dim <= 512, kcap 64 / 256
constexpr int kKernargQueryDims = 512;
bool scan_kernarg_query_supported(int dim, int kcap);
hipError_t launch_scan_topk_host_query(const ScanArgs& args, const float* query_host, int kcap, int grid, hipStream_t stream);
hipError_t launch_score_rows(const ScanArgs& args, u64* out_packed, int q_index, int grid, hipStream_t stream);
hipError_t launch_packed_to_sortkey(u64* data, size_t n, hipStream_t stream);
hipError_t launch_sorted_keys_to_rows(const u64* keys, uint32_t k, uint32_t* out_rows, uint32_t* out_count,
                                      hipStream_t stream, u64* out_packed = nullptr);
hipError_t launch_gather_dot(const ScanArgs& args, const uint32_t* rows, uint32_t n, float* out, hipStream_t stream);
// (rows[i], qidx[i]) items against args.queries = [nq, dim]: one launch for a chunk of queries (dim % 8 == 0, f16 rows)
hipError_t launch_gather_dot_mq(const ScanArgs& args, const uint32_t* rows, const uint32_t* qidx, uint32_t n, float* out, hipStream_t stream);
hipError_t launch_gather_queries(const float* src, const uint32_t* idx, uint32_t n, uint32_t dim, uint32_t src_stride, float* dst,
                                 hipStream_t stream);
hipError_t launch_scatter_hits(const uint32_t* idx, uint32_t n, uint32_t k, const uint32_t* src_rows,
                               const float* src_scores, const uint32_t* src_counts, uint32_t* dst_rows,
                               float* dst_scores, uint32_t* dst_counts, u64* dst_packed, hipStream_t stream);
hipError_t launch_encode_f16(const float* src, size_t n, unsigned short* dst, hipStream_t stream);
// The reference's bench corpus / queries (frankensearch/benches/fsvi_4bit_vs_incumbent.rs:56-101,344-365) written straight into
// device memory: vectors first .. first+n with seeds seed_base + index, as f16 rows (out_f16) or f32 queries (out_f32).
// centroid_scratch holds clusters * dim floats.
hipError_t launch_bench_fixture(uint64_t first, uint64_t n, uint32_t dim, uint32_t clusters, float noise, uint64_t seed_base,
                                float* centroid_scratch, unsigned short* out_f16, float* out_f32, hipStream_t stream);
hipError_t launch_encode_rows_f16(const float* src, const uint32_t* perm, uint64_t n, uint32_t dim, unsigned short* dst,
                                  hipStream_t stream);
hipError_t launch_widen_f16(const unsigned short* src, size_t n, float* dst, hipStream_t stream);

// f32_kernels.hip — Quantization::F32 slabs: every row's packed (score, row) entry, and the gathered dot
// fused scan + top-k over an F32 slab (one query per pass; lists to args.partial like launch_scan_topk); kcap 64 / 256
hipError_t launch_scan_topk_f32(const ScanArgs& args, int kcap, int nq, int grid, hipStream_t stream, int* occupancy);
int scan_f32_queries_per_pass(int dim, int kcap, int want);   // 4, 2 or 1: what the LDS holds
hipError_t launch_score_rows_f32(const ScanArgs& args, u64* out_packed, int q_index, hipStream_t stream);
hipError_t launch_gather_dot_f32(const ScanArgs& args, const uint32_t* rows, uint32_t n, float* out, hipStream_t stream);

// sort_general.hip (own LSD radix sort, descending u64 keys) — the large-k / collect-all path.
// varying_bits: the key bits that can differ between two keys of the input (digits without any are skipped).
hipError_t sort_keys_desc_temp_bytes(size_t n, size_t* temp_bytes);
hipError_t sort_keys_desc(void* temp, size_t temp_bytes, const u64* keys_in, u64* keys_out, size_t n,
                          hipStream_t stream, u64 varying_bits = ~0ull);

// m2v_kernels.hip
hipError_t launch_m2v_embed(const float* table, uint32_t vocab, uint32_t dim, const uint32_t* ids,
                            const uint32_t* offsets, uint32_t n, float* out, hipStream_t stream);

// mfma_scan.hip — batched approximate scan on the matrix cores + exact re-score helpers
struct MfmaScanArgs {
    const void* slab;          // [nrows, dim] f16
    const u64* live;           // may be null
    const u64* allow;          // may be null
    const void* queries;       // [nq_pad, dim] f16 (rows >= nq are zero)
    const float* tau;          // [nq_pad] candidate threshold per query (ignored in dense mode)
    u64* cand;                 // [nq_pad, gridDim.x, slots] packed approximate candidates, one list per block
    u64* spill;                // [nq_pad, spill_cap] candidates that did not fit their block's list
    uint32_t* spill_count;     // [nq_pad * kMfmaSpillCountStride] append counters, one cache line apart
    uint32_t spill_cap;
    uint32_t* overflow;        // [nq_pad] set when a query's spill area overflows too
    u64* dense;                // stage 0: [nq_pad, group_count * 64] packed approximate scores of the sample
    uint32_t nrows;            // rows in the slab
    uint32_t stage;            // 0 = dense sample, 1 = thresholded sample, 2 = main pass (everything stage 1 skipped),
                               // 3 = (mfma_wide.hip, int8 rows) sample that writes each block's four best GROUPS per query into cand
                               //     ([q][block][4]: best approximate score | first row of the group's 8 rows) — no thresholds, no lists
    uint32_t group_stride, group_count;  // the sample, in 64-row groups: {j * group_stride : j < group_count}
    uint32_t dim, slots, row_base;       // slots <= kMfmaMaxSlots
    uint32_t elem_bytes;                 // 2 = f16 slab / f16 queries (0 means 2), 1 = int8 slab / int8 queries
    uint32_t reverse;                    // main pass: walk the slab from its end (alternate passes re-read what the
                                         // previous pass left in the Infinity Cache)
    uint32_t row_stride;                 // bytes between rows; 0 = dense (dim * elem_bytes).  MRL prefix views scan the first
                                         // dim dimensions of rows that are row_stride bytes apart
    uint32_t groups;                     // sample stages: query groups answered by one launch (gridDim.y; 0 means 1) —
                                         // group g's queries/tau/cand/spill/overflow/dense follow group g-1's
    uint32_t* cand_count;                // mfma_wide.hip: [nq_pad, gridDim.x] entries in each (query, block) list, clamped to
                                         // slots (may be null: the lists are then padded with kEmpty instead)
};

// select_kernel (mfma_scan.hip): per query, the k-th best of the packed approximate entries without sorting them
// (k rounds of wave arg-max per wave, k more over the 16 waves' winners) -> tau = a_k - 2 delta; the entries at or
// above tau are the query's candidates.  Two uses:
//   threshold step (slab == null): tau_out for the next scan stage, candidates optionally kept as a pool;
//   finish step (slab != null): the <= kSelectPool candidates are re-scored with the exact-order dot right in the block (one
//   quad per candidate) and the best k_out exact entries are emitted best first.
struct SelectArgs {
    const u64* lists;          // [nq][nlists][list_len] (strides below), kEmpty = hole
    uint64_t q_stride;         // entries between queries
    uint32_t l_stride, nlists, list_len;
    const uint32_t* list_counts;  // [nq][nlists] valid entries at the head of each list (may be null: every slot is read and
                                  // kEmpty marks the holes) — the register-resident-query scan writes counts instead of padding
    const u64* extra;          // [nq, extra_len] more entries per query (may be null)
    uint32_t extra_len;
    const u64* spill;          // [nq, spill_cap] spilled entries (may be null); valid prefix = min(count, spill_cap)
    const uint32_t* spill_count;  // [nq * kMfmaSpillCountStride]
    uint32_t spill_cap;
    uint32_t k;                // 1..kSelectMaxK
    uint32_t* spill_reset;     // [nq * kMfmaSpillCountStride] (may be null) the query's spill counter is ZEROED once this selection has
                               // read what it needs: the next scan stage appends from 0 again — no memset launch between the stages
    const float* delta;        // [nq] error bound; < 0 = query is skipped (tau = +inf, overflow set)
    float* tau_out;            // [nq] (may be null)
    u64* pool_out;             // [nq, kSelectPool] candidates, kEmpty padded (may be null)
    uint32_t* cand_counts;     // [nq] number of candidates, unclamped (may be null)
    uint32_t* overflow;        // [nq] set when there are more than kSelectPool candidates
    uint32_t take_topk;        // != 0: the candidates are the k best entries themselves (exact pass-1 scores)
    // threshold step with the finish fields set and anchor_unit != null (int8 filter): the candidates are re-scored exactly and
    // tau_out = max(a_k - 2 delta, S_k * anchor_unit[q] - delta), S_k = the k-th best EXACT score among them — k real rows score
    // at least S_k, so every true top-k row's approximate score is at least S_k in filter units minus ONE delta
    const float* anchor_unit;  // [nq] filter-score units per exact-score unit (slab scale x query scale); null = off
    // A sample stage whose survivors only ANCHOR the next threshold (the wide main pass visits every row anyway) may emit any
    // subset of rows — k real rows with exact scores bound the final k-th best whichever rows they are: heur_rank != 0 makes
    // tau_out the approximate score at that rank with NO margin (a few dozen rows of the next, larger sample pass instead of
    // thousands), tau_floor_out keeps the proven threshold, and the next selection falls back to it (tau_floor_in) when fewer
    // than k rows came through
    uint32_t heur_rank;
    float* tau_floor_out;      // [nq] (may be null)
    const float* tau_floor_in; // [nq] (may be null)
    uint32_t big_pool;         // finish step: re-score up to 8,192 candidates per query (sorted variant) instead of kSelectPool
    uint32_t* pool_flag;       // [nq] device memory (may be null).  big_pool == 0: a query with more than kSelectPool candidates
                               // sets its flag INSTEAD of `overflow`; big_pool != 0: only flagged queries are processed (second
                               // chance of the finish, launched right behind the first over the same lists)
    // finish step
    const void* slab;          // [nrows, dim] f16
    const float* queries;      // [nq, q_stride_f] f32 (the first dim of each are used)
    uint32_t dim, nrows, row_base;
    uint32_t row_stride;       // bytes between slab rows; 0 = dim * 2
    uint32_t query_stride;     // floats between queries; 0 = dim
    int hreduce;
    uint32_t k_out, out_stride;
    uint32_t* out_rows;        // [nq, out_stride] (may be null; 0xffffffff padding)
    float* out_scores;         // [nq, out_stride] (may be null)
    u64* out_packed;           // [nq, out_stride] (may be null; kEmpty padding)
    uint32_t* out_counts;      // [nq] (may be null)
    // finish step with take_topk (the two-pass searches): the k candidates themselves, for a row-sharded index whose root
    // repeats the selection over all shards' candidates — position i of both lists is the same row: its pass-1 entry
    // (integer score as f32 bits) and its exact entry; kEmpty beyond the candidates.  [nq, k] each, may be null.
    u64* cand_approx_out;
    u64* cand_exact_out;
    uint32_t cand_out_stride;  // entries between queries in both (>= k)
    uint32_t valid_queries;       // blocks q >= this are padding slots whose query lies past the caller's array (0 = every block's query exists)
    unsigned long long* stamps;   // lab builds (FSGPU_EXPERIMENTS): shader clocks of block 0's phases; null otherwise
};
// select_groups_kernel (mfma_scan.hip): the selection behind a group-maxima sample (MfmaScanArgs::stage == 3).  Per query: the
// kGroupsTaken best of its nentries groups (packed best approximate score | first row of the group's 8 rows: rows g + {0..3} and
// g + 16 + {0..3}) -> their live, allowed rows re-scored from the f16 slab in the reference's operation order -> S_k = the k-th best
// exact score among them (k distinct real rows: a lower bound on the final k-th best) ->
//   tau_out = max(a_k - 2 delta, S_k * anchor_unit - delta) in filter units (a_k: the k-th best group maximum),
// exactly what select_kernel's exact-anchor step emits; the query's spill counter is reset for the scan stage that follows.
// delta < 0 (padding / zero / non-finite query): tau = +inf and overflow[q] = 1 (the exact path answers it).
struct GroupSelectArgs {
    const u64* groups;         // [nq][nentries]
    uint32_t nentries;         // <= 1024
    uint32_t k;                // 1..64
    const float* delta;        // [nq]
    const float* anchor_unit;  // [nq] filter-score units per exact-score unit
    float* tau_out;            // [nq]
    uint32_t* overflow;        // [nq] (may be null)
    uint32_t* spill_reset;     // [nq * kMfmaSpillCountStride] (may be null)
    const void* slab;          // [nrows, dim] f16
    const u64* live;           // may be null
    const u64* allow;          // may be null
    const float* queries;      // [nq, query_stride] f32
    uint32_t dim, nrows, row_base, query_stride;
    int hreduce;
    uint32_t valid_queries;    // blocks q >= this are padding slots (0 = every block's query exists)
    uint32_t rank_only;        // != 0 (the int8 two-pass, whose pass-1 scores are the reference's own): no re-score — tau_out = the k-th
                               // best group maximum itself (k <= 64, delta = 0: k distinct rows score at least that); slab may be null
};
constexpr uint32_t kGroupsTaken = 24;      // groups whose rows are re-scored for ranks up to 24 (192 rows) ...
constexpr uint32_t kGroupsTakenMax = 32;   // ... and up to 32 (256 rows: the two-tier flow's fetch of 3 x 10)
constexpr uint32_t kGroupsRankMax = 128;   // rank-only form (int8 / 4-bit two-pass: k x multiplier candidates, 3 x 30 = 90)
hipError_t launch_select_groups(const GroupSelectArgs& args, int nq, hipStream_t stream);
constexpr uint32_t kSelectPool = 1024;
constexpr uint32_t kSelectMaxK = 128;   // largest rank a selection can anchor on (k, or k * multiplier in int8 mode)
hipError_t launch_select(const SelectArgs& args, int nq, hipStream_t stream);
// out[0] = the largest score at slot list_len - 1 of nlists best-first lists (kEmpty = the list is not full), -inf when no list is full:
// no row outside the lists scores above it (the certified lone-query search, VectorIndex::certified_i8_lone_query)
hipError_t launch_list_cut(const u64* lists, uint32_t nlists, uint32_t list_len, float* out, hipStream_t stream);
// Root of a row-sharded two-pass search (search.rs:514-661 over W shards): per query, the W x cc candidate pairs (pass-1 entry,
// exact entry; shard s's lists [nq][cc] start shard_pitch entries after shard s-1's, kEmpty padded) -> the cc best by the pass-1
// order (integer score desc, row asc: exactly the unsharded candidate set, whose members are each in their own shard's top cc)
// -> the k best of those by the exact order.
hipError_t launch_two_pass_merge(const u64* approx_lists, const u64* exact_lists, uint32_t nshards, uint64_t shard_pitch, uint32_t nq,
                                 uint32_t cc, uint32_t k, uint32_t out_stride, uint32_t* out_rows, float* out_scores, uint32_t* out_counts,
                                 hipStream_t stream);
// bits = 8: quantize_i8_query (scale 127/max when max > 0); bits = 4: pack_4bit_query's levels (scale 7/max when max > 1e-9), one per byte
hipError_t launch_prepare_queries_i8(const float* q, uint32_t nq, uint32_t nq_pad, uint32_t dim, void* qi8, float* delta,
                                     hipStream_t stream, int bits = 8);
// the same quantiser for the int8 FILTER of the exact search: delta = a proven bound on |int8 score - exact score / (row
// scale x query scale)| from the slab statistics of launch_i8_slab_stats (see mfma_scan.hip)
hipError_t launch_prepare_queries_i8_filter(const float* q, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t q_stride,
                                            const unsigned int* slab_max_bits, const unsigned int* slab_stats, void* qi8,
                                            float* delta, hipStream_t stream, float* unit_out = nullptr, double extra_coeff = 0.0);

constexpr uint32_t kMfmaMaxSlots = 32;           // candidate slots per (block, query) staged in LDS
constexpr uint32_t kMfmaSpillCountStride = 16;  // uint32 counters 64 bytes apart
constexpr uint32_t kWideSlots = 32;               // ... by the register-resident-query main pass (mfma_wide.hip)
bool scan_mfma_supported(int dim);
// mfma_wide.hip: main pass with the queries in registers and the row tiles in an LDS-DMA ring; query_tiles 2 = 256, 3 = 384
// queries per launch; one candidate list of args.slots <= kWideSlots entries per (query, block)
bool scan_wide_supported(int dim, int elem_bytes);
bool scan_wide_group_maxima_supported(int dim, int query_tiles);   // MfmaScanArgs::stage == 3 (int8 rows): see select_groups
int scan_wide_max_query_tiles(int dim, int elem_bytes);  // 3 for f16 rows of 384 dimensions, 5 for their int8 form
hipError_t launch_scan_wide(const MfmaScanArgs& args, int query_tiles, int grid, hipStream_t stream, int* occupancy);
void note_main_pass_kernel(const char* name);  // remembers the instantiation the last main pass ran (last_main_pass_kernel)
const char* last_main_pass_kernel();  // template instantiation of the last batched main pass launched, as rocprofv3 names it
// shape: see mfma_scan.hip (0 = 64 queries; 1..3 = 128 queries with different row tiling / buffering)
int scan_mfma_waves_per_block(int shape);
int scan_mfma_rows_per_tile(int shape);
int scan_mfma_query_tiles(int shape);
int scan_mfma_max_slots(int shape);  // LDS staging capacity per (query, block)
hipError_t launch_scan_mfma(const MfmaScanArgs& args, int shape, int grid, hipStream_t stream, int* occupancy);
hipError_t launch_max_row_norm(const void* slab, uint32_t nrows, uint32_t dim, uint32_t row_stride_bytes, unsigned int* out_bits,
                               hipStream_t stream);
hipError_t launch_prepare_queries(const float* q, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t q_stride,
                                  const unsigned int* max_norm_bits, void* qh, float* delta, hipStream_t stream);

// int8_kernels.hip
// max_ready: *max_bits_dev already holds the (corpus-wide) max-abs — a sharded index reduces the shards' values first
hipError_t launch_slab_maxabs(const void* slab_f16, size_t n_values, unsigned int* max_bits_dev, hipStream_t stream);
hipError_t launch_quantize_slab_i8(const void* slab_f16, size_t n_values, unsigned int* max_bits_dev, void* out_i8,
                                   hipStream_t stream, bool max_ready = false);
// 4-bit levels (-7..7, the reference's nibble quantiser) one per byte: the batched 4-bit pass 1 reuses the int8 matrix-core kernels
hipError_t launch_quantize_slab_4bit_levels(const void* slab_f16, size_t n_values, unsigned int* max_bits_dev, void* out_i8,
                                            hipStream_t stream, bool max_ready = false);
// bounds on what the int8 slab misses of the f16 slab (int8 filter of the exact batched search): stats_dev[0..4) =
// { f32 bits of max_row sum eps^2, max_row sum |r|, max_row sum r^2, non-finite flag }
hipError_t launch_i8_slab_stats(const void* slab_f16, const void* slab_i8, uint32_t nrows, uint32_t dim,
                                const unsigned int* max_bits_dev, unsigned int* stats_dev, hipStream_t stream);
// the int8 filter's rotated copy (int8_kernels.hip): rows / queries through a fixed orthogonal map in f64, rounded once to f32;
// rt = the map TRANSPOSED, [dim][dim] f64.  _f32: a row with an element that is not finite or above 65,504 comes out as NaN.
hipError_t launch_rotate_rows_f16(const void* rows_f16, uint32_t nrows, uint32_t dim, const double* rt, float* out, hipStream_t stream);
hipError_t launch_rotate_rows_f32(const float* rows, uint32_t nrows, uint32_t row_stride, uint32_t dim, const double* rt, float* out,
                                  hipStream_t stream);
// ... and the quantiser / statistics of launch_quantize_slab_i8 / launch_i8_slab_stats over f32 rows, chunk by chunk: max-abs and
// statistics ACCUMULATE into words the caller zeroed
hipError_t launch_maxabs_f32(const float* v, size_t n, unsigned int* max_bits_dev, hipStream_t stream);
hipError_t launch_quantize_f32_i8(const float* v, size_t n, const unsigned int* max_bits_dev, void* out_i8, hipStream_t stream);
hipError_t launch_i8_stats_f32(const float* rows, const void* rows_i8, uint32_t nrows, uint32_t dim, const unsigned int* max_bits_dev,
                               unsigned int* stats_dev, hipStream_t stream);
bool scan_i8_fused_supported(int dim, int kcap);
// 4-bit two-pass (int8_kernels.hip, BITS = 4)
hipError_t launch_pack_slab_4bit(const void* slab_f16, uint64_t count, uint32_t dim, unsigned int* max_bits_dev,
                                 void* out_4bit, hipStream_t stream, bool max_ready = false);
bool scan_4bit_fused_supported(int dim, int kcap);
hipError_t launch_scan_4bit(const ScanArgs& args, const void* slab_4bit, const void* query_4bit, int kcap, int grid,
                            hipStream_t stream, int* occupancy);
hipError_t launch_score_rows_4bit(const ScanArgs& args, const void* slab_4bit, const void* query_4bit, u64* out_packed,
                                  hipStream_t stream);
hipError_t launch_scan_i8(const ScanArgs& args, const void* slab_i8, const void* query_i8, int kcap, int grid,
                          hipStream_t stream, int* occupancy);
hipError_t launch_score_rows_i8(const ScanArgs& args, const void* slab_i8, const void* query_i8, u64* out_packed,
                                hipStream_t stream);
hipError_t launch_packed_rows(const u64* packed, uint32_t n, uint32_t* rows, hipStream_t stream);
hipError_t launch_pack_hits(const uint32_t* rows, const float* scores, uint32_t n, u64* packed, hipStream_t stream);

// bert_kernels.hip
hipError_t launch_bert_embed_ln(const int32_t* ids, const int32_t* positions, const float* word, const float* pos,
                                const float* type0, const float* lnw, const float* lnb, float* x_f32, void* x_h,
                                int tokens, int hidden, float eps, hipStream_t stream);
hipError_t launch_bert_add_ln(float* x_f32, const float* delta, const float* lnw, const float* lnb, void* x_h, int tokens,
                              int hidden, float eps, hipStream_t stream);
bool bert_gemm_ln_supported(int hidden);
hipError_t launch_bert_gemm_ln(const void* a_h, const void* w_h, const float* bias, float* x_f32, void* x_h,
                               const float* lnw, const float* lnb, int M, int hidden, int K, float eps, hipStream_t stream);
hipError_t launch_bert_gemm(const void* a_h, const void* w_h, const float* bias, float* out_f32, void* out_h, int M,
                            int N, int K, bool gelu_half_out, hipStream_t stream);
hipError_t launch_bert_attention(const float* qkv, const uint32_t* offsets, void* ctx_h, int n_docs, int heads,
                                 int hidden, int max_seq, float scale, hipStream_t stream);
hipError_t launch_bert_attention_h(const void* qkv_h, const uint32_t* offsets, void* ctx_h, int n_docs, int heads,
                                   int hidden, int max_seq, float scale, hipStream_t stream);   // Q, K, V already f16
hipError_t launch_bert_pool(const float* x, const uint32_t* offsets, float* out, int n_docs, int hidden,
                            hipStream_t stream);
hipError_t launch_bert_to_half(const float* src, void* dst, size_t n, hipStream_t stream);

// bert_gemm_w.hip: batch-path linears over weights pre-packed in matrix-core fragment order
hipError_t launch_bert_pack_w(const void* w_h, void* packed_h, int N, int K, hipStream_t stream);
bool bert_gemm_w_supported(int N, int K);
// epilogue 0: f32 output; 1: GELU, f16 output; 2: f16 output
hipError_t launch_bert_gemm_w(const void* a_h, const void* wp, const float* bias, float* out_f32, void* out_h, int M, int N,
                              int K, int epilogue, hipStream_t stream);
bool bert_gemm_ln_w_supported(int hidden, int K);
hipError_t launch_bert_gemm_ln_w(const void* a_h, const void* wp, const float* bias, float* x_f32, void* x_h,
                                 const float* lnw, const float* lnb, int M, int hidden, int K, float eps, hipStream_t stream);
// the whole feed-forward block (up-projection, GELU, down-projection, residual, LayerNorm) in one launch
bool bert_ffn_w_supported(int hidden, int inter);
hipError_t launch_bert_ffn_w(const void* w1p, const float* b1, const void* w2p, const float* b2, float* x_f32, void* x_h,
                             const float* lnw, const float* lnb, int M, int hidden, int inter, float eps, hipStream_t stream);
// everything of a layer after the attention in one launch: output projection + LayerNorm, then the feed-forward block
bool bert_post_attn_w_supported(int hidden, int inter);
hipError_t launch_bert_post_attn_w(const void* ctx_h, const void* w0p, const float* b0, const float* ln0w, const float* ln0b,
                                   const void* w1p, const float* b1, const void* w2p, const float* b2, float* x_f32, void* x_h,
                                   const float* lnw, const float* lnb, int M, int hidden, int inter, float eps,
                                   hipStream_t stream);

// bert_query_kernels.hip: the MiniLM-L6 forward for at most 32 tokens in 25 launches (4 per layer + the pooling), every
// add+LayerNorm riding in the prologue of the GEMM that consumes it.  One argument block serves all stages; a stage reads
// the fields it needs.
struct BertQueryArgs {
    int tokens, n_docs;              // tokens <= 32
    const uint32_t* offsets;         // [n_docs + 1]
    // first stage: embedding gather + LayerNorm instead of the pending add+LayerNorm (ids != null)
    const int32_t* ids;
    const int32_t* positions;
    const float *word, *pos, *type0;
    // pending add+LayerNorm: x = LN(x_in + sum of n_parts slabs of parts ([slab][32][384] f32) + prev_bias)
    const float* x_in;
    float* x_out;                    // the block that owns it stores the normalised rows here (null: nobody does)
    const float* parts;
    int n_parts;
    const float *prev_bias, *lnw, *lnb;
    float eps, attn_scale;
    // GEMM: A rows from a_h (f16, leading dimension lda) unless the stage has a LayerNorm prologue; W [n, ldw] f16; bias [n]
    const _Float16* a_h;
    int lda;
    const _Float16* w;
    int ldw, n;
    const float* bias;
    float* out_f32;                  // partial slabs [slab][32][n]
    _Float16* out_h;                 // f16 rows [tokens][n]
};
bool bert_query_path_supported(int hidden, int inter, int heads);
hipError_t launch_bert_q_qkv_attn(const BertQueryArgs& a, int heads, hipStream_t stream);
hipError_t launch_bert_q_gemm(const BertQueryArgs& a, int mode, hipStream_t stream);
hipError_t launch_bert_q_pool(const BertQueryArgs& a, float* out, hipStream_t stream);
// the same stages in ONE launch of bert_q_one_launch_blocks() resident blocks with grid-wide barriers between them
int bert_q_one_launch_blocks();
hipError_t launch_bert_q_one_launch(const BertQueryArgs* stages_dev, const unsigned char* kinds_dev, int n_stages, int tokens, int n_docs,
                                    const uint32_t* offsets, const int32_t* ids, const int32_t* positions, float* out,
                                    unsigned int* counter_dev, unsigned int base, unsigned int* status_mapped, hipStream_t stream);

// bert_docs_w.hip: the whole MiniLM-L6 forward of a batch of texts of at most 32 tokens each in ONE launch — a 32-row block
// owns whole texts, so the six layers chain inside the kernel (native.rs:1142-1236)
struct BertDocsLayer {           // device pointers of one encoder layer: fragment-order f16 weights (bert_pack_w_kernel), f32 vectors
    const void *qkv_wp, *ao_wp, *i_wp, *o_wp;
    const float *qkv_b, *ao_b, *ln1_w, *ln1_b, *i_b, *o_b, *ln2_w, *ln2_b;
};
struct BertDocsArgs {
    const uint32_t* offsets;     // [n_docs + 1] text i owns tokens [offsets[i], offsets[i + 1])
    const uint32_t* blk_tok;     // [nblocks + 1] first token of each row block: boundaries on text boundaries, at most 32 apart
    const uint32_t* blk_doc;     // [nblocks + 1] first text of each row block
    const int32_t* row_id;       // [nblocks * 32] token id of each row of each block, -1 = padding row
    const uint32_t* row_meta;    // [nblocks * 32] rows of the block << 16 | position inside its text << 8 | its text's index among
                                 // the block's non-empty texts
    const float *word, *pos, *type0, *emb_lnw, *emb_lnb;
    const BertDocsLayer* layers; // [nlayers], device memory
    int nlayers;
    float eps, attn_scale;
    float* out;                  // [n_docs][hidden] pooled, L2-normalised
    unsigned long long* stamps;  // lab builds (FSGPU_EXPERIMENTS): shader-clock stamps of block 0's phases; null otherwise
};
bool bert_docs_w_supported(int hidden, int inter, int heads);
hipError_t launch_bert_docs_w(const BertDocsArgs& a, uint32_t nblocks, hipStream_t stream);

}  // namespace fsgpu
