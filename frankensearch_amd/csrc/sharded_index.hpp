// sharded_index.hpp — one VectorIndex per GPU behind ONE handle.
//
// The reference partitions the scan into contiguous row chunks and merges the per-chunk heaps
// (scan_parallel + merge_partial_heaps, crates/frankensearch-index/src/search.rs:1013-1036,1704-1720).  The same
// shape across the GPUs of a node, inside the library so that a host makes one call: shard r owns the contiguous
// rows [r*ceil(N/W), ...) and reports GLOBAL row ids (the (score, row) tie-break is shard-invariant); the queries are
// replicated; every shard produces packed [nq, k] hits on its own stream; ONE ncclAllGather (RCCL over xGMI, nq*k*8 bytes
// per shard) puts the W lists on every device; the root merges them with merge_topk_kernel under the reference order.
// No row exchange.  The only other collective is the 4-byte ncclAllReduce(max) that gives the int8 / 4-bit quantisers their
// ONE corpus-wide scale (simd.rs:1865-1886), once per index.
//
// A search is begin() + end(), both on the CALLING thread: begin() enqueues every shard's scan on that shard's scan stream (the
// batched and two-pass searches through their own begin halves — nothing is waited for), records an event behind each, and
// enqueues the exchange — one ncclGroupStart .. ncclAllGather x W .. ncclGroupEnd on the shards' EXCHANGE streams, each behind its
// shard's event — then the merge and the copy of the hits to pinned host memory on the root's exchange stream.  Nothing between
// scan, all-gather and merge waits on the host; end() waits for ONE event, then reads every shard's verdicts (its end half); only
// when a shard had to answer an uncertified query after its list had travelled does the exchange run a second time.  Two tickets
// may be in flight, so the exchange + merge of search i run underneath the scan of search i + 1.  (Through round 4 every shard had
// a host thread that called the BLOCKING search: a mutex / condition-variable round trip per job and per shard, and none of the
// begin / end pipelining of the unsharded path.)
//
// Hybrid layout: W = G x S devices as G query groups x S row shards.  Rank r scans row shard r % S for the queries of group
// r / S (a contiguous 1/G of the batch); every row shard is resident G times (a 10M x 384 slab is 7.68 GB of a GPU's 288).  A
// shard's step has a fixed part (sample, selections, launches) that does not shrink with its rows: at 8 devices, 2 groups x 4 row
// shards pay it over 2.5M rows and half the queries each instead of 1.25M rows and all of them.  One all-gather over all W ranks
// (equal list sizes), one merge per group over its S lists.
//
// A LONE query (one host query, no filter; exact or two-pass) takes neither the device merge nor a collective: every shard of one
// group answers through its own latency lane (VectorIndex::lone_*_begin / _end: certified int8 pass, two-pass lane) into its pinned
// block, all begun before any is ended, and the calling thread merges the S short lists — merge_partial_heaps on the host, where the
// reference runs it (search.rs:1704-1720).  Groups take lone queries in turn.
#pragma once

#include <memory>
#include <mutex>
#include <string>
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
    // query_groups G (>= 1, divides ndev): ndev / G row shards, each resident on G devices (hybrid layout above)
    SearchError init_host(const int32_t* devices, uint32_t ndev, uint32_t dim, uint64_t nrows, const void* slab_f16,
                          const uint64_t* live, int32_t exchange, uint32_t query_groups = 1);
    // shard_rows / slabs_dev / live_dev per DEVICE (rank); with query groups, rank r holds row shard r % (ndev / G)
    SearchError init_device(const int32_t* devices, uint32_t ndev, uint32_t dim, const uint64_t* shard_rows,
                            const void* const* slabs_dev, const uint64_t* const* live_dev, int32_t exchange, uint32_t query_groups = 1);
    // VectorIndex::open for an FSVI v1 file (F16 slab), rows split over the devices; keeps the record table / doc ids / WAL
    SearchError open_fsvi(const char* path, const int32_t* devices, uint32_t ndev, int32_t exchange, uint32_t query_groups = 1);

    // what a search runs on every shard before the lists are exchanged
    enum Mode : int32_t { kExact = 0, kBatched = 1, kInt8TwoPass = 2, kFourBitTwoPass = 3 };
    struct Request {
        const float* queries = nullptr;   // [nq, dim] host
        const float* queries_dev = nullptr;   // ... or already resident on the ROOT shard's device (an encoder's device output): no
                                              // staging and no H2D copy; the other shards fetch them from the root over xGMI (peer
                                              // copies on their scan streams).  Must stay unchanged until end().
        uint32_t nq = 0, k = 0;
        Mode mode = kExact;
        uint32_t multiplier = 0;          // candidate_multiplier of the two-pass modes
        const uint64_t* allow = nullptr;  // [ceil(N/64)] index-wide allow bitmap (a precomputed SearchFilter), may be null
        // ... or resident in PARTS on several devices (data-parallel encoders: part p holds part_counts[p] consecutive queries on
        // device part_devices[p]); every rank fetches the slice of its query group peer to peer.  Must stay unchanged until end().
        const float* const* parts_dev = nullptr;
        const uint32_t* part_counts = nullptr;
        const int32_t* part_devices = nullptr;
        uint32_t n_parts = 0;
    };
    // search_top_k{,_batched,_int8_two_pass,...} over nq host queries.  k <= 256, dim % 8 == 0 (the fused tiers; the packed
    // lists are what travels).  begin() returns a ticket; at most two may be pending; end() in ticket order.
    SearchError begin(const Request& rq, uint32_t query_len, uint64_t* ticket);
    SearchError end(uint64_t ticket, uint32_t* out_rows, float* out_scores, uint32_t* out_counts, uint32_t* fallbacks);
    SearchError search(const Request& rq, uint32_t query_len, uint32_t* out_rows, float* out_scores, uint32_t* out_counts,
                       uint32_t* fallbacks);

    // index-wide tombstone bitmap (1 bit per row, set = live): split per shard
    SearchError set_live_bitmap(const uint64_t* live);
    // VectorIndex::soft_delete / append / doc ids / search_top_k with WAL + dedup: need open_fsvi's tables
    SearchError soft_delete(const char* doc_id, uint32_t len, int32_t* deleted);
    SearchError wal_append(const char* doc_id, uint32_t len, const float* vector, uint32_t vector_len);
    SearchError doc_id_at(uint32_t row, const char** ptr, uint32_t* len) const;
    SearchError search_hits(const float* query, uint32_t query_len, uint32_t k, uint32_t* out_rows, float* out_scores,
                            uint32_t* out_count);
    uint64_t wal_record_count() const { return catalog_ ? catalog_->wal_record_count() : 0; }
    // dot_query_at over global row ids, routed to the owning shards
    SearchError gather_dot(const float* query, uint32_t query_len, const uint32_t* rows, uint32_t n, float* out);

    uint64_t record_count() const { return nrows_; }
    uint32_t dimension() const { return dim_; }
    // the tables of an FSVI-opened handle (record table, doc ids, tombstones, WAL); null for raw shards
    const VectorIndex* catalog() const { return catalog_.get(); }
    uint32_t shard_count() const { return (uint32_t)shards_.size(); }   // devices (ranks): query groups x row shards
    uint32_t query_groups() const { return groups_; }
    uint32_t row_shards() const { return row_shards_; }
    int32_t exchange_mode() const { return use_rccl_ ? 1 : 2; }
    int32_t shard_device(uint32_t shard) const { return shard < shards_.size() ? shards_[shard]->device : -1; }
    bool shard_range(uint32_t shard, uint64_t* lo, uint64_t* hi) const;
    void set_hreduce(int32_t mode);
    // fsgpu_index_set_int8_latency on every shard: lone exact queries go through the certified pass over the shard's int8 copy
    // (built here, with the corpus-wide scale) instead of the exact kernel's pass over the f16 rows — same rows and score bits
    SearchError set_int8_latency(bool on);
    // the corpus-wide max-abs the quantised copies of every shard are built from (0 until a two-pass search asked for it)
    float quant_scale_max() const { return quant_max_; }
    std::mutex& mutex() { return call_mu_; }

  private:
    static constexpr int kSlots = 2;   // tickets in flight
    struct Slot {
        DeviceBuffer queries, packed, gathered, allow;
        hipEvent_t scan_done = nullptr;   // recorded on the scan stream behind the shard's enqueued search
        hipEvent_t sent = nullptr;        // peer copies: this shard's list has reached the root's gather buffer
        int32_t ticket = -1;              // the shard index's begun batched / two-pass search (-1: nothing to end)
    };
    struct Shard {
        int device = -1;
        uint64_t lo = 0, rows = 0;
        VectorIndex index;
        hipStream_t stream = nullptr;    // scan: the shard index's own stream (lone lanes and batch scans in ONE order)
        bool owns_stream = false;
        hipStream_t xstream = nullptr;   // exchange (+ merge and D2H on the root)
        Slot slot[kSlots];
        void* comm = nullptr;            // ncclComm_t
    };
    struct RootSlot {
        DeviceBuffer out_rows, out_scores, out_counts;
        void* stage = nullptr;           // pinned: [queries | rows | scores | counts]
        size_t stage_bytes = 0;
        hipEvent_t done = nullptr;
        bool pending = false;
        uint64_t ticket = 0;
        uint32_t nq = 0, k = 0, fallbacks = 0;
        Mode mode = kExact;
        uint32_t multiplier = 0;
        uint32_t per = 0;                // queries per group (the last groups may hold fewer)
        std::vector<uint64_t> allow_slices;   // per-row-shard slices of the request's allow bitmap, back to back (host)
        // a lone query: answered by the shards of ONE group into their pinned blocks, merged on the host in end()
        bool lone = false;
        uint32_t lone_group = 0;
        std::vector<float> lone_query;
    };

    SearchError finish_init(int32_t exchange);
    SearchError check_layout(uint32_t ndev, uint32_t query_groups);
    SearchError enqueue_scan(const Request& rq, uint32_t r, int slot);   // rank r's share of the request, on its scan stream
    SearchError enqueue_exchange(int slot);
    SearchError end_scans(int slot, uint32_t* late, uint32_t* fallbacks = nullptr);   // every rank's end half; *late: queries answered in it (their lists travel again), *fallbacks: by the exact kernels
    SearchError begin_lone(const Request& rq, int slot, uint32_t group);
    SearchError end_lone(RootSlot& rs, uint32_t* out_rows, float* out_scores, uint32_t* out_counts);
    SearchError ensure_quant_scale();    // corpus-wide max-abs: ncclAllReduce(max) / host max, once
    SearchError push_live_slices(const std::vector<uint64_t>& live);
    uint32_t owner_of(uint64_t row) const;

    uint32_t dim_ = 0;
    uint64_t nrows_ = 0;
    uint32_t groups_ = 1, row_shards_ = 1;
    std::vector<std::unique_ptr<Shard>> shards_;
    RootSlot root_[kSlots];
    bool use_rccl_ = false;
    bool quant_ready_ = false;
    float quant_max_ = 0.f;
    std::unique_ptr<VectorIndex> catalog_;   // open_fsvi: record table, doc ids, tombstones, WAL (no device state)
    // one begin/end at a time per handle (the staging buffers and the shards' streams are per handle)
    std::mutex call_mu_;
    uint64_t next_ticket_ = 1;
    uint32_t lone_rr_ = 0;               // the group that takes the next lone query
};

}  // namespace fsgpu
