/* fsgpu_lab.h — entry points of libfsgpu.so that exist for bench.py, the profiling scripts and A/B runs; NOT part of the drop-in
 * boundary (include/fsgpu.h), no reference interface behind them.  A host that embeds the library never needs this header.
 */
#ifndef FSGPU_LAB_H
#define FSGPU_LAB_H

#include "fsgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- bench / test fixture ---- */
/* The reference's own bench generator (frankensearch/benches/fsvi_4bit_vs_incumbent.rs:56-101,344-365) run on the GPU so that
 * a 10M- or 50M-row corpus never crosses PCIe: xorshift64 raw_vector, `clusters` normalised centroids
 * (seed 0xc0000000 + c), vector i = normalize(centroid[i % clusters] + noise * raw_vector(seed_base + i)) for
 * i in [first, first + n).  as_f16 = 1: out_dev receives n x dim little-endian f16 rows (corpus: seed_base 1);
 * as_f16 = 0: n x dim f32 (queries: seed_base 0xdead0000).  The bytes equal the CPU generator's (same operation order).
 * hip_stream may be NULL (default stream); the call returns after the kernels have finished. */
fsgpu_status fsgpu_bench_fixture_device(int32_t device, uint64_t first, uint64_t n, uint32_t dim, uint32_t clusters, float noise,
                                        uint64_t seed_base, int32_t as_f16, void *out_dev, void *hip_stream);

/* ---- instrumentation ---- */
/* When enabled, HIP events bracket the scan kernel of every fsgpu_search_topk* call; enabled = n > 1: of the batched exact search's
 * merged main launch only every n-th is bracketed (an event pair idles the stream ~6 us on either side of the launch). */
fsgpu_status fsgpu_index_set_profiling(fsgpu_index *idx, int32_t enabled);
/* Sum of scan-kernel time (ms) and number of scan launches since the last reset; synchronises. */
fsgpu_status fsgpu_index_scan_time(fsgpu_index *idx, double *total_ms, uint64_t *launches, int32_t reset);
/* Same, plus the number of slab rows those launches streamed (the batched path's timed main pass skips the rows its
 * sampling stage already covered), so that bytes/launch can be stated exactly. */
fsgpu_status fsgpu_index_scan_stats(fsgpu_index *idx, double *total_ms, uint64_t *launches, uint64_t *rows,
                                    int32_t reset);
/* Filtered searches (allow bitmap given) answered by scoring only the allowed rows (try_gather_filtered,
 * crates/frankensearch-index/src/search.rs:1114-1180: taken when allowed * 50 < rows) and by the masked full scan. */
fsgpu_status fsgpu_index_filter_stats(fsgpu_index *idx, uint64_t *gathered, uint64_t *scanned);
/* Name of the template instantiation the last batched main pass launched in this process ran ("" before the first one), spelled as
 * rocprofv3 prints it: lets bench.py tie a committed PMC summary to the kernel that actually ran. */
const char *fsgpu_last_main_pass_kernel(void);
/* The library's own descending radix sort of 64-bit sortkeys (csrc/sort_general.hip: the collect-all / large-k path,
 * search.rs:449-473), host arrays in and out — so that tests can check it against a host sort at tile boundaries.  varying_bits: the
 * bits in which two keys of the input may differ (~0 when unknown): a digit without one gets no pass. */
fsgpu_status fsgpu_lab_sort_keys_desc(int32_t device, const uint64_t *keys, uint64_t n, uint64_t varying_bits, uint64_t *out_sorted);
/* Selects the scan kernel variant (0 = default) — used by bench A/B runs only. */
fsgpu_status fsgpu_index_set_variant(fsgpu_index *idx, int32_t variant);

#ifdef __cplusplus
}
#endif
#endif /* FSGPU_LAB_H */
