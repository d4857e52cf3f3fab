// sharded_index.hpp — one VectorIndex per GPU behind ONE handle.
//
// The reference partitions the scan into contiguous row chunks and merges the per-chunk heaps
// (scan_parallel + merge_partial_heaps, crates/frankensearch-index/src/search.rs:1013-1036,1704-1720).  The same
// shape across the GPUs of a node, inside the library so that a host makes one call: shard r owns the contiguous
// rows [r*ceil(N/W), ...) and reports GLOBAL row ids (the (score, row) tie-break is shard-invariant); the queries are
// replicated; every shard produces packed [nq, k] hits on its own stream from its own host thread; ONE ncclAllGather
// (RCCL over xGMI, nq*k*8 bytes per shard) puts the W lists on every device; the root merges them with
// merge_topk_kernel under the reference order.  No all-reduce, no row exchange.
#pragma once

#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "vector_index.hpp"

namespace fsgpu {

class ShardedIndex {
  public:
    ShardedIndex() = default;
    ~ShardedIndex();
    ShardedIndex(const ShardedIndex&) = delete;
    ShardedIndex& operator=(const ShardedIndex&) = delete;

    // exchange: 0 = RCCL when it can be used (distinct devices, librccl loads), else peer copies; 1 = RCCL or fail;
    // 2 = peer copies (hipMemcpyAsync device-to-device into the root's gather buffer)
    SearchError init_host(const int32_t* devices, uint32_t ndev, uint32_t dim, uint64_t nrows, const void* slab_f16,
                          const uint64_t* live, int32_t exchange);
    SearchError init_device(const int32_t* devices, uint32_t ndev, uint32_t dim, const uint64_t* shard_rows,
                            const void* const* slabs_dev, const uint64_t* const* live_dev, int32_t exchange);

    // search_top_k over nq host queries: exact kernels (batched = false) or the matrix-core batched path (results
    // identical).  k <= 256, dim % 8 == 0 (the fused tiers; the packed lists are what travels).
    SearchError search(const float* queries, uint32_t nq, uint32_t query_len, uint32_t k, bool batched, uint32_t* out_rows,
                       float* out_scores, uint32_t* out_counts, uint32_t* fallbacks);

    uint64_t record_count() const { return nrows_; }
    uint32_t dimension() const { return dim_; }
    uint32_t shard_count() const { return (uint32_t)shards_.size(); }
    int32_t exchange_mode() const { return use_rccl_ ? 1 : 2; }
    bool shard_range(uint32_t shard, uint64_t* lo, uint64_t* hi) const;
    void set_hreduce(int32_t mode);
    std::mutex& mutex() { return call_mu_; }

  private:
    struct Shard {
        int device = -1;
        uint64_t lo = 0, rows = 0;
        VectorIndex index;
        hipStream_t stream = nullptr;
        DeviceBuffer queries, packed, gathered, out_rows, out_scores, out_counts;
        void* comm = nullptr;  // ncclComm_t
        std::thread worker;
        SearchError error;
        uint32_t fallbacks = 0;
    };
    struct Job {
        const float* queries = nullptr;  // pinned staging
        uint32_t nq = 0, k = 0;
        bool batched = false;
    };

    SearchError finish_init(int32_t exchange);
    void worker_main(uint32_t r);
    void run_phase(int phase);  // wakes the workers for one phase of the current job and waits for all of them
    SearchError shard_search(Shard& s);
    SearchError shard_exchange(uint32_t r);

    uint32_t dim_ = 0;
    uint64_t nrows_ = 0;
    std::vector<std::unique_ptr<Shard>> shards_;
    bool use_rccl_ = false;
    // one search at a time per handle (the workers and staging buffers are per handle)
    std::mutex call_mu_;
    void* stage_host_ = nullptr;  // pinned: queries in, hits out
    size_t stage_bytes_ = 0;
    Job job_;
    // phase hand-off between the calling thread and the shard workers
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    uint64_t generation_ = 0;
    int phase_ = 0;
    uint32_t pending_ = 0;
    bool stop_ = false;
};

}  // namespace fsgpu
