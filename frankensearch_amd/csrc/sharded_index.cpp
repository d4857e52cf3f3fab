// sharded_index.cpp — row-sharded search over the GPUs of one node behind one handle (sharded_index.hpp).
//
// Reference shape: scan_parallel's contiguous chunks + merge_partial_heaps
// (crates/frankensearch-index/src/search.rs:1013-1036,1704-1720); SURVEY §8e for the multi-GPU form.
// RCCL is bound at run time (dlopen of librccl.so.1): libfsgpu.so keeps linking only libamdhip64, a single-GPU host
// never needs RCCL on its library path, and inside a PyTorch process the already loaded copy (same SONAME) is reused.
#include "sharded_index.hpp"

#include <chrono>
#include <cstdio>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>

#include "../../include/fsgpu.h"

namespace fsgpu {

namespace {

SearchError make_err(int32_t code, std::string detail) {
    SearchError e;
    e.code = code;
    e.detail = std::move(detail);
    return e;
}

SearchError hip_err(hipError_t e, const char* what) {
    return make_err(FSGPU_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

#define SH_HIP(expr)                                     \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return hip_err(_e, #expr); \
    } while (0)

#define SH_TRY(expr)               \
    do {                           \
        SearchError _s = (expr);   \
        if (!_s.ok()) return _s;   \
    } while (0)

// The handful of RCCL entry points the exchange needs, resolved once.
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*comm_init_all)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*group_start)() = nullptr;
    ncclResult_t (*group_end)() = nullptr;
    ncclResult_t (*comm_abort)(ncclComm_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    std::string why;  // why it is unavailable

    static Rccl& get() {
        static Rccl r = load();
        return r;
    }
    bool ok() const { return lib != nullptr; }

  private:
    static Rccl load() {
        Rccl r;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            const char* e = dlerror();
            r.why = std::string("librccl.so.1 could not be loaded: ") + (e ? e : "unknown error");
            return r;
        }
        r.comm_init_all = reinterpret_cast<decltype(r.comm_init_all)>(dlsym(r.lib, "ncclCommInitAll"));
        r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(r.lib, "ncclAllGather"));
        r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(r.lib, "ncclAllReduce"));
        r.group_start = reinterpret_cast<decltype(r.group_start)>(dlsym(r.lib, "ncclGroupStart"));
        r.group_end = reinterpret_cast<decltype(r.group_end)>(dlsym(r.lib, "ncclGroupEnd"));
        r.comm_abort = reinterpret_cast<decltype(r.comm_abort)>(dlsym(r.lib, "ncclCommAbort"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.comm_init_all || !r.all_gather || !r.all_reduce || !r.group_start || !r.group_end || !r.comm_destroy || !r.error_string) {
            r.why = "librccl.so.1 lacks ncclCommInitAll / ncclAllGather / ncclAllReduce / ncclGroupStart / ncclGroupEnd / ncclCommDestroy";
            r.lib = nullptr;
        }
        return r;
    }
};

// bits [lo, lo+rows) of a row bitmap, re-based to bit 0 (a shard's live bitmap out of the index-wide one)
std::vector<uint64_t> slice_bitmap(const uint64_t* bits, uint64_t lo, uint64_t rows) {
    std::vector<uint64_t> out((size_t)((rows + 63) / 64), 0);
    const unsigned sh = (unsigned)(lo & 63);
    const uint64_t w0 = lo >> 6, total_words = (lo + rows + 63) >> 6;
    for (size_t i = 0; i < out.size(); ++i) {
        uint64_t v = bits[w0 + i] >> sh;
        if (sh && w0 + i + 1 < total_words) v |= bits[w0 + i + 1] << (64 - sh);
        out[i] = v;
    }
    if (rows & 63) out.back() &= (1ull << (rows & 63)) - 1;
    return out;
}

// ---- host-side order (device_util.hpp's sortkey, restated for the lone query's host merge) ----
uint64_t host_sortkey(uint64_t packed) {
    uint32_t bits = (uint32_t)(packed >> 32);
    if ((bits & 0x7fffffffu) > 0x7f800000u) bits = 0xff800000u;            // NaN ranks as -inf (score_key, search.rs:1655-1661)
    const uint32_t ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);   // f32::total_cmp
    return ((uint64_t)ord << 32) | (uint32_t)(~(uint32_t)packed);           // ties: the lower row first
}

}  // namespace

ShardedIndex::~ShardedIndex() {
    for (auto& s : shards_) {
        if (s->device >= 0) (void)hipSetDevice(s->device);
        if (s->stream) (void)hipStreamSynchronize(s->stream);
        if (s->xstream) (void)hipStreamSynchronize(s->xstream);
        if (s->comm && Rccl::get().ok()) (void)Rccl::get().comm_destroy(static_cast<ncclComm_t>(s->comm));
        for (Slot& sl : s->slot) {
            for (DeviceBuffer* b : {&sl.queries, &sl.packed, &sl.gathered, &sl.allow}) b->release();
            if (sl.scan_done) (void)hipEventDestroy(sl.scan_done);
            if (sl.sent) (void)hipEventDestroy(sl.sent);
        }
        if (s->xstream) (void)hipStreamDestroy(s->xstream);
        if (s->stream && s->owns_stream) (void)hipStreamDestroy(s->stream);
    }
    if (!shards_.empty()) (void)hipSetDevice(shards_[0]->device);
    for (RootSlot& r : root_) {
        for (DeviceBuffer* b : {&r.out_rows, &r.out_scores, &r.out_counts}) b->release();
        if (r.stage) (void)hipHostFree(r.stage);
        if (r.done) (void)hipEventDestroy(r.done);
    }
}

SearchError ShardedIndex::check_layout(uint32_t ndev, uint32_t query_groups) {
    if (ndev == 0) return make_err(FSGPU_ERR_INVALID_CONFIG, "at least one device is required");
    if (query_groups == 0 || ndev % query_groups != 0)
        return make_err(FSGPU_ERR_INVALID_CONFIG, "query_groups must divide the number of devices (query groups x row shards)");
    groups_ = query_groups;
    row_shards_ = ndev / query_groups;
    return SearchError{};
}

SearchError ShardedIndex::init_host(const int32_t* devices, uint32_t ndev, uint32_t dim, uint64_t nrows, const void* slab_f16,
                                    const uint64_t* live, int32_t exchange, uint32_t query_groups) {
    if (!devices) return make_err(FSGPU_ERR_INVALID_CONFIG, "at least one device is required");
    SH_TRY(check_layout(ndev, query_groups));
    if (dim == 0) return make_err(FSGPU_ERR_INVALID_CONFIG, "dimension must be greater than zero");
    if (nrows >= 0xffffffffull) return make_err(FSGPU_ERR_INVALID_CONFIG, "row ids must fit in u32 (VectorHit.index)");
    if (nrows > 0 && !slab_f16) return make_err(FSGPU_ERR_NULL_ARGUMENT, "slab is null");
    dim_ = dim;
    nrows_ = nrows;
    // contiguous ceil split, exactly the chunking of scan_parallel (search.rs:1020-1035) at shard granularity; with query groups
    // every row shard is uploaded once per group
    const uint64_t per = (nrows + row_shards_ - 1) / row_shards_;
    for (uint32_t r = 0; r < ndev; ++r) {
        const uint32_t rs = r % row_shards_;
        auto s = std::make_unique<Shard>();
        s->device = devices[r];
        s->lo = std::min<uint64_t>(nrows, (uint64_t)rs * per);
        s->rows = std::min<uint64_t>(nrows, s->lo + per) - s->lo;
        std::vector<uint64_t> bits;
        if (live && s->rows) bits = slice_bitmap(live, s->lo, s->rows);
        const unsigned char* base = static_cast<const unsigned char*>(slab_f16) + (size_t)s->lo * dim * 2;
        SH_TRY(s->index.init_host(s->device, dim, s->rows, s->rows ? base : nullptr, bits.empty() ? nullptr : bits.data(), s->lo));
        shards_.push_back(std::move(s));
    }
    return finish_init(exchange);
}

SearchError ShardedIndex::init_device(const int32_t* devices, uint32_t ndev, uint32_t dim, const uint64_t* shard_rows,
                                      const void* const* slabs_dev, const uint64_t* const* live_dev, int32_t exchange, uint32_t query_groups) {
    if (!devices || !shard_rows || !slabs_dev) return make_err(FSGPU_ERR_INVALID_CONFIG, "devices, shard_rows and slabs are required");
    SH_TRY(check_layout(ndev, query_groups));
    if (dim == 0) return make_err(FSGPU_ERR_INVALID_CONFIG, "dimension must be greater than zero");
    dim_ = dim;
    std::vector<uint64_t> los(row_shards_ + 1, 0);
    for (uint32_t rs = 0; rs < row_shards_; ++rs) los[rs + 1] = los[rs] + shard_rows[rs];
    if (los[row_shards_] >= 0xffffffffull) return make_err(FSGPU_ERR_INVALID_CONFIG, "row ids must fit in u32 (VectorHit.index)");
    for (uint32_t r = 0; r < ndev; ++r) {
        const uint32_t rs = r % row_shards_;
        if (shard_rows[r] != shard_rows[rs])
            return make_err(FSGPU_ERR_INVALID_CONFIG, "with query groups, device r holds row shard r % (devices / groups): the same rows in every group");
        auto s = std::make_unique<Shard>();
        s->device = devices[r];
        s->lo = los[rs];
        s->rows = shard_rows[r];
        SH_TRY(s->index.init_device(s->device, dim, s->rows, slabs_dev[r], live_dev ? live_dev[r] : nullptr, s->lo));
        shards_.push_back(std::move(s));
    }
    nrows_ = los[row_shards_];
    return finish_init(exchange);
}

// VectorIndex::open (lib.rs:1747-1909) for a sharded index: the file is read and validated once; its record table, doc-id
// strings, tombstone flags (and later its WAL) stay in the catalog, the F16 slab is split over the devices.
SearchError ShardedIndex::open_fsvi(const char* path, const int32_t* devices, uint32_t ndev, int32_t exchange, uint32_t query_groups) {
    if (!devices || ndev == 0) return make_err(FSGPU_ERR_INVALID_CONFIG, "at least one device is required");
    catalog_ = std::make_unique<VectorIndex>();
    VectorIndex::FsviImage img;
    SH_TRY(catalog_->open_fsvi_catalog(path, &img));
    if (img.f32_rows) return make_err(FSGPU_ERR_INVALID_CONFIG, "a sharded index needs an F16 slab (Quantization::F16)");
    const std::vector<uint64_t>& live = catalog_->live_host();
    SH_TRY(init_host(devices, ndev, img.dim, img.nrows, img.bytes.data() + img.slab_offset, live.empty() ? nullptr : live.data(), exchange,
                     query_groups));
    // hits of the catalog's search_top_k (WAL merge, shadowing, dedup) come from the shards
    catalog_->topk_override = [this](const float* q, uint32_t k, uint32_t* rows, float* scores, uint32_t* count) -> SearchError {
        Request rq;
        rq.queries = q;
        rq.nq = 1;
        rq.k = k;
        rq.mode = kExact;
        return search(rq, dim_, rows, scores, count, nullptr);
    };
    return SearchError{};
}

SearchError ShardedIndex::finish_init(int32_t exchange) {
    if (exchange < 0 || exchange > 2) return make_err(FSGPU_ERR_INVALID_CONFIG, "exchange must be 0 (auto), 1 (RCCL) or 2 (peer copies)");
    const uint32_t w = (uint32_t)shards_.size();
    bool distinct = true;
    for (uint32_t a = 0; a < w; ++a)
        for (uint32_t b = a + 1; b < w; ++b) distinct &= shards_[a]->device != shards_[b]->device;
    for (auto& s : shards_) {
        SH_HIP(hipSetDevice(s->device));
        // ONE stream per shard for everything that touches the shard index's workspaces: the lone lanes run on the index's own stream
        // (VectorIndex::lone_*), so the batch scans are enqueued there too — a lone ticket and a batch ticket in flight together are
        // ordered by the stream instead of racing on ws_partial_ / rot_q_ / ws_queries_
        s->stream = s->index.stream();
        if (!s->stream) {
            SH_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
            s->owns_stream = true;
        }
        SH_HIP(hipStreamCreateWithFlags(&s->xstream, hipStreamNonBlocking));
        for (Slot& sl : s->slot) {
            SH_HIP(hipEventCreateWithFlags(&sl.scan_done, hipEventDisableTiming));
            SH_HIP(hipEventCreateWithFlags(&sl.sent, hipEventDisableTiming));
        }
    }
    SH_HIP(hipSetDevice(shards_[0]->device));
    for (RootSlot& r : root_) SH_HIP(hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
    // RCCL wants one rank per device; several shards on one device (a rehearsal of the N-way path on fewer GPUs)
    // exchange their lists with plain device-to-device copies instead
    if (exchange == 1 && !distinct) return make_err(FSGPU_ERR_INVALID_CONFIG, "RCCL exchange needs distinct devices");
    if (exchange != 2 && distinct) {
        Rccl& rc = Rccl::get();
        if (!rc.ok()) {
            if (exchange == 1) return make_err(FSGPU_ERR_DEVICE, rc.why);
        } else {
            std::vector<int> devs(w);
            for (uint32_t r = 0; r < w; ++r) devs[r] = shards_[r]->device;
            std::vector<ncclComm_t> comms(w, nullptr);
            const ncclResult_t st = rc.comm_init_all(comms.data(), (int)w, devs.data());
            if (st != ncclSuccess) {
                if (exchange == 1) return make_err(FSGPU_ERR_DEVICE, std::string("ncclCommInitAll: ") + rc.error_string(st));
            } else {
                for (uint32_t r = 0; r < w; ++r) shards_[r]->comm = comms[r];
                use_rccl_ = true;
            }
        }
    }
    if (w > 1) {
        // peer access among the devices: the peer-copy exchange writes into the root's gather buffer, and device-resident queries
        // (the root's, or the parts data-parallel encoders left on their devices) are fetched peer to peer — without it HIP stages
        // such copies through the host: slower, still correct
        for (uint32_t a = 0; a < w; ++a)
            for (uint32_t b = 0; b < w; ++b) {
                if (shards_[a]->device == shards_[b]->device) continue;
                int can = 0;
                SH_HIP(hipDeviceCanAccessPeer(&can, shards_[a]->device, shards_[b]->device));
                if (!can) continue;
                SH_HIP(hipSetDevice(shards_[a]->device));
                const hipError_t e = hipDeviceEnablePeerAccess(shards_[b]->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled && !use_rccl_ && b == 0) return hip_err(e, "hipDeviceEnablePeerAccess");
                (void)hipGetLastError();
            }
    }
    return SearchError{};
}

bool ShardedIndex::shard_range(uint32_t shard, uint64_t* lo, uint64_t* hi) const {
    if (shard >= shards_.size()) return false;
    *lo = shards_[shard]->lo;
    *hi = shards_[shard]->lo + shards_[shard]->rows;
    return true;
}

void ShardedIndex::set_hreduce(int32_t mode) {
    for (auto& s : shards_) s->index.hreduce = mode;
    if (catalog_) catalog_->hreduce = mode;
}

SearchError ShardedIndex::set_int8_latency(bool on) {
    for (const RootSlot& r : root_)
        if (r.pending) return make_err(FSGPU_ERR_INVALID_CONFIG, "a search is in flight on this handle: end it first");
    if (on) {
        SH_TRY(ensure_quant_scale());   // the copies are built from ONE corpus-wide scale (simd.rs:1865-1886)
        for (auto& s : shards_) {
            if (!s->rows) continue;
            SH_HIP(hipSetDevice(s->device));
            SH_TRY(s->index.prepare_int8_latency());
        }
    }
    for (auto& s : shards_) s->index.int8_latency = on;
    return SearchError{};
}

uint32_t ShardedIndex::owner_of(uint64_t row) const {   // (the first group's copy of the row shard)
    for (uint32_t r = 0; r < row_shards_ && r < shards_.size(); ++r)
        if (row >= shards_[r]->lo && row < shards_[r]->lo + shards_[r]->rows) return r;
    return (uint32_t)shards_.size();
}

// Rank r's share of a request, enqueued on its scan stream by the calling thread: its query group's slice of the batch against its
// row shard, the result left packed — best-first [per, k] (global row ids, ~0 padding), or for the two-pass modes the candidate
// pairs [2][per, cc] — and `scan_done` behind it.  The batched and two-pass searches go through their begin halves (their verdicts
// are read in end()); the exact kernels are enqueue-only anyway.
SearchError ShardedIndex::enqueue_scan(const Request& rq, uint32_t r, int slot) {
    Shard& s = *shards_[r];
    Slot& sl = s.slot[slot];
    const RootSlot& rs = root_[slot];
    const bool two_pass = rq.mode == kInt8TwoPass || rq.mode == kFourBitTwoPass;
    const uint64_t cc = two_pass ? std::max<uint64_t>((uint64_t)rq.k * (rq.multiplier ? rq.multiplier : 1), rq.k) : rq.k;
    const uint32_t per = rs.per, g = r / row_shards_, rsh = r % row_shards_;
    const uint32_t q_lo = g * per, nqg = q_lo < rq.nq ? std::min(per, rq.nq - q_lo) : 0;
    const size_t lbytes = (size_t)per * cc * 8 * (two_pass ? 2 : 1);
    sl.ticket = -1;
    SH_HIP(hipSetDevice(s.device));
    uint64_t* packed = static_cast<uint64_t*>(sl.packed.ptr);
    if (nqg < per || s.rows == 0) SH_HIP(hipMemsetAsync(packed, 0xff, lbytes, s.stream));   // (queries this rank does not hold: empty lists)
    if (nqg && s.rows) {
        const size_t qbytes = (size_t)nqg * dim_ * 4;
        // host queries: every rank copies its slice over ITS OWN PCIe link from the pinned block (the copies run side by side);
        // device queries: in the root's HBM (or in parts on the encoders' devices): fetched peer to peer
        const float* qd = static_cast<const float*>(sl.queries.ptr);
        if (rq.n_parts) {
            uint32_t at = 0;   // first query of the current part
            for (uint32_t p = 0; p < rq.n_parts; ++p) {
                const uint32_t lo = std::max(at, q_lo), hi = std::min(at + rq.part_counts[p], q_lo + nqg);
                if (lo < hi) {
                    const float* src = rq.parts_dev[p] + (size_t)(lo - at) * dim_;
                    if (lo == q_lo && hi == q_lo + nqg && rq.part_devices[p] == s.device) {
                        qd = src;   // the whole slice lies in one part on this very device: scanned in place
                    } else {
                        SH_HIP(hipMemcpyAsync(static_cast<float*>(sl.queries.ptr) + (size_t)(lo - q_lo) * dim_, src, (size_t)(hi - lo) * dim_ * 4,
                                              hipMemcpyDeviceToDevice, s.stream));
                    }
                }
                at += rq.part_counts[p];
            }
        } else if (rq.queries_dev) {
            if (s.device == shards_[0]->device) qd = rq.queries_dev + (size_t)q_lo * dim_;
            else SH_HIP(hipMemcpyAsync(sl.queries.ptr, rq.queries_dev + (size_t)q_lo * dim_, qbytes, hipMemcpyDeviceToDevice, s.stream));
        } else {
            SH_HIP(hipMemcpyAsync(sl.queries.ptr, static_cast<const float*>(rs.stage) + (size_t)q_lo * dim_, qbytes, hipMemcpyHostToDevice, s.stream));
        }
        const uint64_t* allow_dev = nullptr;
        if (rq.allow) {
            size_t off = 0;
            for (uint32_t x = 0; x < rsh; ++x) off += (size_t)((shards_[x]->rows + 63) / 64);
            SH_HIP(hipMemcpyAsync(sl.allow.ptr, rs.allow_slices.data() + off, (size_t)((s.rows + 63) / 64) * 8, hipMemcpyHostToDevice, s.stream));
            allow_dev = static_cast<const uint64_t*>(sl.allow.ptr);
        }
        if (two_pass) {
            SH_TRY(s.index.two_pass_candidates_device_begin(qd, nqg, dim_, rq.k, rq.multiplier, rq.mode == kFourBitTwoPass ? 4 : 8, packed,
                                                            packed + (size_t)per * cc, s.stream, &sl.ticket));
        } else if (rq.mode == kBatched) {
            SH_TRY(s.index.search_top_k_batched_device_begin(qd, nqg, dim_, rq.k, allow_dev, nullptr, nullptr, nullptr, s.stream, packed, &sl.ticket));
        } else {
            SH_TRY(s.index.search_top_k_packed_device(qd, nqg, dim_, rq.k, allow_dev, packed, s.stream));
        }
    }
    SH_HIP(hipEventRecord(sl.scan_done, s.stream));
    return SearchError{};
}

// Every rank's end half (the verdicts of its batched / two-pass search; the rare uncertified query answered by the exact kernels on
// the rank's scan stream).  *late = queries answered that way: their lists were corrected after the exchange had been enqueued.
SearchError ShardedIndex::end_scans(int slot, uint32_t* late, uint32_t* fallbacks) {
    uint32_t fb_sink = 0;
    if (!fallbacks) fallbacks = &fb_sink;
    *fallbacks = 0;
    *late = 0;
    const RootSlot& rs = root_[slot];
    const bool two_pass = rs.mode == kInt8TwoPass || rs.mode == kFourBitTwoPass;
    SearchError first;
    for (auto& sp : shards_) {
        Slot& sl = sp->slot[slot];
        if (sl.ticket < 0) continue;
        uint32_t fb = 0, late_here = 0;
        const SearchError e = two_pass ? sp->index.two_pass_candidates_device_end(sl.ticket, &fb, &late_here)
                                       : sp->index.search_top_k_batched_device_end(sl.ticket, &fb, &late_here);
        sl.ticket = -1;
        if (!e.ok() && first.ok()) first = e;   // (every ticket is ended whatever the others reported)
        // (late = every query answered in the end half: exact fallbacks AND queries re-filtered on the f16 slab — through round 5 only
        // the fallbacks counted, and a re-filtered query's corrected list never travelled: scripts/fuzz_sharded.py, round 6)
        *late += late_here;
        *fallbacks += fb;
    }
    return first;
}

// The corpus-wide max-abs of the quantisers (simd.rs:1865-1886 computes ONE scale over the whole slab): every shard's own
// max-abs, reduced with ncclAllReduce(max) over the shards' communicators (4 bytes) — or on the host when the
// shards exchange by peer copies —, then adopted by every shard as THE scale of its int8 / 4-bit copies.
SearchError ShardedIndex::ensure_quant_scale() {
    if (quant_ready_) return SearchError{};
    const uint32_t w = (uint32_t)shards_.size();
    std::vector<unsigned int*> bits(w, nullptr);
    for (uint32_t r = 0; r < w; ++r) {
        SH_HIP(hipSetDevice(shards_[r]->device));
        SH_TRY(shards_[r]->index.compute_local_quant_max(&bits[r], shards_[r]->stream));
    }
    unsigned int global_bits = 0;
    if (use_rccl_ && w > 1) {
        Rccl& rc = Rccl::get();
        // (max over non-negative f32 values: reduced as floats)
        ncclResult_t st = rc.group_start();
        for (uint32_t r = 0; r < w && st == ncclSuccess; ++r) {
            (void)hipSetDevice(shards_[r]->device);
            st = rc.all_reduce(bits[r], bits[r], 1, ncclFloat32, ncclMax, static_cast<ncclComm_t>(shards_[r]->comm), shards_[r]->stream);
        }
        const ncclResult_t st2 = rc.group_end();
        if (st != ncclSuccess || st2 != ncclSuccess)
            return make_err(FSGPU_ERR_DEVICE, std::string("ncclAllReduce(max): ") + rc.error_string(st != ncclSuccess ? st : st2));
        SH_HIP(hipSetDevice(shards_[0]->device));
        SH_HIP(hipMemcpyAsync(&global_bits, bits[0], 4, hipMemcpyDeviceToHost, shards_[0]->stream));
        for (uint32_t r = 0; r < w; ++r) {
            SH_HIP(hipSetDevice(shards_[r]->device));
            SH_HIP(hipStreamSynchronize(shards_[r]->stream));
        }
    } else {
        std::vector<unsigned int> local(w, 0);
        for (uint32_t r = 0; r < w; ++r) {
            SH_HIP(hipSetDevice(shards_[r]->device));
            SH_HIP(hipMemcpyAsync(&local[r], bits[r], 4, hipMemcpyDeviceToHost, shards_[r]->stream));
            SH_HIP(hipStreamSynchronize(shards_[r]->stream));
        }
        for (unsigned int b : local) global_bits = std::max(global_bits, b);   // non-negative floats order like their bits
        for (uint32_t r = 0; r < w; ++r) {
            SH_HIP(hipSetDevice(shards_[r]->device));
            SH_HIP(hipMemcpyAsync(bits[r], &global_bits, 4, hipMemcpyHostToDevice, shards_[r]->stream));
            SH_HIP(hipStreamSynchronize(shards_[r]->stream));
        }
    }
    for (auto& s : shards_) s->index.adopt_global_quant_max();
    std::memcpy(&quant_max_, &global_bits, 4);
    quant_ready_ = true;
    return SearchError{};
}

// The exchange + merge of one slot, enqueued by the calling thread with no host wait in between: every rank's exchange stream
// waits for that rank's scan_done event; the lists travel (ncclAllGather inside ONE group call / peer copies into the root's
// gather buffer); the root's exchange stream merges every query group's S lists and copies the hits into the slot's pinned block;
// `done` marks the end.
SearchError ShardedIndex::enqueue_exchange(int slot) {
    const uint32_t w = (uint32_t)shards_.size();
    RootSlot& rs = root_[slot];
    const uint32_t nq = rs.nq, k = rs.k, per = rs.per;
    const bool two_pass = rs.mode == kInt8TwoPass || rs.mode == kFourBitTwoPass;
    const uint64_t cc = two_pass ? std::max<uint64_t>((uint64_t)k * (rs.multiplier ? rs.multiplier : 1), k) : k;
    const size_t count = (size_t)per * cc * (two_pass ? 2 : 1), lbytes = count * 8;   // one rank's list(s)
    Shard& root = *shards_[0];
    for (uint32_t r = 0; r < w; ++r) {
        SH_HIP(hipSetDevice(shards_[r]->device));
        SH_HIP(hipStreamWaitEvent(shards_[r]->xstream, shards_[r]->slot[slot].scan_done, 0));
    }
    if (w > 1 && use_rccl_) {
        Rccl& rc = Rccl::get();
        ncclResult_t st = rc.group_start();
        for (uint32_t r = 0; r < w && st == ncclSuccess; ++r) {
            Slot& sl = shards_[r]->slot[slot];
            (void)hipSetDevice(shards_[r]->device);   // (one thread, W communicators: the device of each call's communicator is current when it is issued)
            st = rc.all_gather(sl.packed.ptr, sl.gathered.ptr, count, ncclUint64, static_cast<ncclComm_t>(shards_[r]->comm), shards_[r]->xstream);
        }
        const ncclResult_t st2 = rc.group_end();
        if (st != ncclSuccess || st2 != ncclSuccess) {
            // a communicator that saw a failed call is unusable: abort them all so that nothing stays parked in a collective
            if (rc.comm_abort)
                for (auto& s : shards_)
                    if (s->comm) {
                        (void)rc.comm_abort(static_cast<ncclComm_t>(s->comm));
                        s->comm = nullptr;
                    }
            return make_err(FSGPU_ERR_DEVICE, std::string("ncclAllGather: ") + rc.error_string(st != ncclSuccess ? st : st2));
        }
    } else if (w > 1) {
        for (uint32_t r = 0; r < w; ++r) {
            Shard& s = *shards_[r];
            SH_HIP(hipSetDevice(s.device));
            SH_HIP(hipMemcpyAsync(static_cast<unsigned char*>(root.slot[slot].gathered.ptr) + (size_t)r * lbytes, s.slot[slot].packed.ptr,
                                  lbytes, hipMemcpyDeviceToDevice, s.xstream));
            SH_HIP(hipEventRecord(s.slot[slot].sent, s.xstream));
        }
        SH_HIP(hipSetDevice(root.device));
        for (uint32_t r = 1; r < w; ++r) SH_HIP(hipStreamWaitEvent(root.xstream, shards_[r]->slot[slot].sent, 0));
    }
    // merge_partial_heaps across a group's row shards (search.rs:1704-1720) on the root; one row shard: its own list is the
    // answer, the merge only unpacks it.  Two-pass modes: the corpus-wide candidate selection, then the exact top-k.
    SH_HIP(hipSetDevice(root.device));
    const uint64_t* lists = static_cast<const uint64_t*>(w > 1 ? root.slot[slot].gathered.ptr : root.slot[slot].packed.ptr);
    uint32_t* d_rows = static_cast<uint32_t*>(rs.out_rows.ptr);
    float* d_scores = static_cast<float*>(rs.out_scores.ptr);
    uint32_t* d_counts = static_cast<uint32_t*>(rs.out_counts.ptr);
    for (uint32_t g = 0; g < groups_; ++g) {
        const uint32_t q_lo = g * per;
        if (q_lo >= nq) break;
        const uint32_t nqg = std::min(per, nq - q_lo);
        const uint64_t* base = lists + (size_t)g * row_shards_ * count;   // rank g * S's list(s); the group's next S - 1 follow, `count` apart
        if (two_pass) {
            // a rank's block: [approx per * cc | exact per * cc]
            const u64* pairs = reinterpret_cast<const u64*>(base);
            SH_HIP(launch_two_pass_merge(pairs, pairs + (size_t)per * cc, row_shards_, (uint64_t)count, nqg, (uint32_t)cc, k, k,
                                         d_rows + (size_t)q_lo * k, d_scores + (size_t)q_lo * k, d_counts + q_lo, root.xstream));
        } else {
            SH_TRY(merge_packed_lists_device(root.device, base, nqg, row_shards_, k, k, (uint64_t)count, k, d_rows + (size_t)q_lo * k,
                                             d_scores + (size_t)q_lo * k, d_counts + q_lo, root.xstream));
        }
    }
    const size_t qbytes = (size_t)nq * dim_ * 4, hbytes = (size_t)nq * k * 4, cbytes = (size_t)nq * 4;
    unsigned char* stage = static_cast<unsigned char*>(rs.stage);
    SH_HIP(hipMemcpyAsync(stage + qbytes, d_rows, hbytes, hipMemcpyDeviceToHost, root.xstream));
    SH_HIP(hipMemcpyAsync(stage + qbytes + hbytes, d_scores, hbytes, hipMemcpyDeviceToHost, root.xstream));
    SH_HIP(hipMemcpyAsync(stage + qbytes + 2 * hbytes, d_counts, cbytes, hipMemcpyDeviceToHost, root.xstream));
    SH_HIP(hipEventRecord(rs.done, root.xstream));
    return SearchError{};
}

// A lone query: one group's shards answer through their own latency lanes (VectorIndex::lone_*), all begun here.
SearchError ShardedIndex::begin_lone(const Request& rq, int slot, uint32_t group) {
    RootSlot& rs = root_[slot];
    rs.lone = true;
    rs.lone_group = group;
    rs.lone_query.assign(rq.queries, rq.queries + dim_);
    const bool two_pass = rq.mode == kInt8TwoPass || rq.mode == kFourBitTwoPass;
    SearchError first;
    uint32_t begun = 0;
    for (uint32_t x = 0; x < row_shards_ && first.ok(); ++x) {
        Shard& s = *shards_[rs.lone_group * row_shards_ + x];
        first = two_pass ? s.index.lone_two_pass_begin(rs.lone_query.data(), rq.k, rq.multiplier, rq.mode == kFourBitTwoPass ? 4 : 8)
                         : s.index.lone_exact_begin(rs.lone_query.data(), rq.k);
        if (first.ok()) ++begun;
    }
    if (!first.ok()) {   // drain what was begun: a shard index holds one lone query at a time
        std::vector<uint64_t> scratch(2 * 256);
        std::vector<uint32_t> r32(rq.k + 1);
        std::vector<float> f32(rq.k + 1);
        uint32_t c = 0;
        for (uint32_t x = 0; x < begun; ++x) {
            Shard& s = *shards_[rs.lone_group * row_shards_ + x];
            if (two_pass) (void)s.index.lone_two_pass_end(scratch.data(), scratch.data() + 256);
            else (void)s.index.lone_exact_end(r32.data(), f32.data(), &c);
        }
        rs.lone = false;
    }
    return first;
}

// ... ended here, and merged on the host: exact = the k best of the shards' k-lists (merge_partial_heaps, search.rs:1704-1720);
// two-pass = the cc best candidates by pass-1 key across the shards, re-keyed by their exact entries, the k best of those
// (two_pass_merge_kernel's selection, on S x cc pairs).
SearchError ShardedIndex::end_lone(RootSlot& rs, uint32_t* out_rows, float* out_scores, uint32_t* out_counts) {
    const uint32_t k = rs.k;
    const bool two_pass = rs.mode == kInt8TwoPass || rs.mode == kFourBitTwoPass;
    const uint64_t cc = two_pass ? std::max<uint64_t>((uint64_t)k * (rs.multiplier ? rs.multiplier : 1), k) : k;
    SearchError first;
    std::vector<std::pair<uint64_t, uint64_t>> cand;   // (sort key to select by, the entry that is emitted)
    std::vector<uint64_t> approx(cc), exact(cc);
    std::vector<uint32_t> rows(k);
    std::vector<float> scores(k);
    for (uint32_t x = 0; x < row_shards_; ++x) {   // (every shard is ended, whatever the others reported)
        Shard& s = *shards_[rs.lone_group * row_shards_ + x];
        if (two_pass) {
            const SearchError e = s.index.lone_two_pass_end(approx.data(), exact.data());
            if (!e.ok()) {
                if (first.ok()) first = e;
                continue;
            }
            for (uint64_t i = 0; i < cc; ++i)
                if (approx[i] != ~0ull && exact[i] != ~0ull) cand.emplace_back(host_sortkey(approx[i]), exact[i]);
        } else {
            uint32_t c = 0;
            const SearchError e = s.index.lone_exact_end(rows.data(), scores.data(), &c);
            if (!e.ok()) {
                if (first.ok()) first = e;
                continue;
            }
            for (uint32_t i = 0; i < c && i < k; ++i) {
                uint32_t bits;
                std::memcpy(&bits, &scores[i], 4);
                const uint64_t entry = ((uint64_t)bits << 32) | rows[i];
                cand.emplace_back(host_sortkey(entry), entry);
            }
        }
    }
    rs.lone = false;
    if (!first.ok()) return first;
    auto by_key = [](const std::pair<uint64_t, uint64_t>& a, const std::pair<uint64_t, uint64_t>& b) { return a.first > b.first; };
    std::sort(cand.begin(), cand.end(), by_key);
    if (two_pass) {
        if (cand.size() > cc) cand.resize((size_t)cc);   // the corpus-wide pass-1 candidates ...
        for (auto& c : cand) c.first = host_sortkey(c.second);   // ... re-keyed by their exact entries
        std::sort(cand.begin(), cand.end(), by_key);
    }
    uint32_t n = 0;
    for (; n < k && n < cand.size(); ++n) {
        out_rows[n] = (uint32_t)cand[n].second;
        const uint32_t bits = (uint32_t)(cand[n].second >> 32);
        std::memcpy(&out_scores[n], &bits, 4);
    }
    for (uint32_t i = n; i < k; ++i) {   // the padding the device merge writes (kEmpty unpacked)
        out_rows[i] = 0xffffffffu;
        const uint32_t bits = 0xffffffffu;
        std::memcpy(&out_scores[i], &bits, 4);
    }
    out_counts[0] = n;
    return SearchError{};
}

#ifdef FSGPU_SHARDED_TIMING   // lab: where a search through the handle spends its host time (printed every 256 searches)
namespace {
struct ShTiming {
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t n = 0;
    void add(int i, std::chrono::steady_clock::time_point a) { t[i] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); }
    void done() {
        if (++n % 256 == 0) {
            std::fprintf(stderr, "[sharded timing] per search, us: stage copy %.1f | reservations %.1f | enqueue scans %.1f | enqueue exchange %.1f | wait %.1f | end scans %.1f | copy out %.1f\n",
                         t[0] / n, t[1] / n, t[2] / n, t[3] / n, t[4] / n, t[5] / n, t[6] / n);
        }
    }
};
ShTiming g_sh_timing;
}  // namespace
#define SH_T0() auto sh_t0 = std::chrono::steady_clock::now()
#define SH_T(i) do { g_sh_timing.add((i), sh_t0); sh_t0 = std::chrono::steady_clock::now(); } while (0)
#define SH_TDONE() g_sh_timing.done()
#else
#define SH_T0() do { } while (0)
#define SH_T(i) do { } while (0)
#define SH_TDONE() do { } while (0)
#endif

SearchError ShardedIndex::begin(const Request& rq, uint32_t query_len, uint64_t* ticket) {
    if (!ticket) return make_err(FSGPU_ERR_NULL_ARGUMENT, "ticket is null");
    *ticket = 0;
    if (query_len != dim_)
        return make_err(FSGPU_ERR_DIMENSION_MISMATCH, "expected " + std::to_string(dim_) + ", found " + std::to_string(query_len));
    const bool two_pass = rq.mode == kInt8TwoPass || rq.mode == kFourBitTwoPass;
    const uint64_t cc = two_pass ? std::max<uint64_t>((uint64_t)rq.k * (rq.multiplier ? rq.multiplier : 1), rq.k) : rq.k;
    if (rq.nq && rq.k && nrows_) {
        if (dim_ % 8 != 0 || rq.k > 256)
            return make_err(FSGPU_ERR_INVALID_CONFIG, "the sharded search exchanges the fused tiers' packed lists: k <= 256 and dim % 8 == 0");
        if (two_pass && (cc > 256 || cc * row_shards_ > 1024))
            return make_err(FSGPU_ERR_INVALID_CONFIG, "sharded two-pass: k * multiplier <= 256 and row shards * k * multiplier <= 1024");
        if (two_pass && rq.allow) return make_err(FSGPU_ERR_INVALID_CONFIG, "the two-pass searches take no filter (search.rs:514-661)");
        if (rq.n_parts) {
            if (!rq.parts_dev || !rq.part_counts || !rq.part_devices) return make_err(FSGPU_ERR_NULL_ARGUMENT, "query parts: pointers, counts and devices are required");
            uint64_t total = 0;
            for (uint32_t p = 0; p < rq.n_parts; ++p) total += rq.part_counts[p];
            if (total != rq.nq) return make_err(FSGPU_ERR_INVALID_CONFIG, "query parts must add up to nq");
        } else if (!rq.queries && !rq.queries_dev) {
            return make_err(FSGPU_ERR_NULL_ARGUMENT, "queries is null");
        }
    }
    const int slot = (int)(next_ticket_ % kSlots);
    RootSlot& rs = root_[slot];
    if (rs.pending) return make_err(FSGPU_ERR_INVALID_CONFIG, "two searches are already in flight on this handle: end one first");
    const uint32_t w = (uint32_t)shards_.size();
    Shard& root = *shards_[0];
    rs.nq = rq.nq;
    rs.k = rq.k;
    rs.fallbacks = 0;
    rs.ticket = next_ticket_;
    rs.mode = rq.mode;
    rs.multiplier = rq.multiplier;
    rs.lone = false;
    for (auto& s : shards_) s->slot[slot].ticket = -1;
    if (rq.nq == 0 || rq.k == 0 || nrows_ == 0) {   // nothing to enqueue: end() reports empty results
        rs.pending = true;
        rs.nq = rq.k == 0 || nrows_ == 0 ? rq.nq : 0;
        rs.k = 0;
        *ticket = next_ticket_++;
        return SearchError{};
    }
    if (two_pass) SH_TRY(ensure_quant_scale());
    if (rq.nq == 1 && rq.queries && !rq.queries_dev && !rq.n_parts && !rq.allow && rq.mode != kBatched) {
        // A shard index holds ONE lone query at a time (one LoneState, one pinned block).  With another lone ticket still in flight
        // on the group whose turn it is, this query goes to the next group — or, with a single group, down the collective path
        // below (the same kernels' rows and score bits, a device merge instead of the host's): ADVICE r05, high.
        uint32_t group = lone_rr_ % groups_;
        bool lane_free = true;
        for (int o = 0; o < kSlots; ++o) {
            const RootSlot& other = root_[o];
            if (o == slot || !other.pending || !other.lone || other.k == 0 || other.lone_group != group) continue;
            if (groups_ > 1) group = (group + 1) % groups_;   // (kSlots = 2: one other ticket at most, the next group is free)
            else lane_free = false;
        }
        if (lane_free) {
            ++lone_rr_;
            SH_TRY(begin_lone(rq, slot, group));
            rs.pending = true;
            *ticket = next_ticket_++;
            return SearchError{};
        }
    }
    // query groups: group g takes queries [g * per, (g + 1) * per) — the last groups may hold fewer, or none
    const uint32_t per = (rq.nq + groups_ - 1) / groups_;
    rs.per = per;
    // pinned staging: [queries | rows | scores | counts]
    const size_t qbytes = (size_t)rq.nq * dim_ * 4, hbytes = (size_t)rq.nq * rq.k * 4, cbytes = (size_t)rq.nq * 4;
    const size_t need = qbytes + 2 * hbytes + cbytes;
    SH_HIP(hipSetDevice(root.device));
    if (need > rs.stage_bytes) {
        if (rs.stage) (void)hipHostFree(rs.stage);
        rs.stage = nullptr;
        rs.stage_bytes = 0;
        SH_HIP(hipHostMalloc(&rs.stage, need, hipHostMallocPortable));
        rs.stage_bytes = need;
    }
    SH_T0();
    if (!rq.queries_dev && !rq.n_parts) std::memcpy(rs.stage, rq.queries, qbytes);
    SH_T(0);
    // EVERY reservation a rank or the exchange needs happens here, before any work is enqueued: a failed allocation must not
    // leave some ranks inside a collective that others never enter
    const size_t lbytes = (size_t)per * cc * 8 * (two_pass ? 2 : 1);
    for (uint32_t r = 0; r < w; ++r) {
        Slot& sl = shards_[r]->slot[slot];
        SH_HIP(hipSetDevice(shards_[r]->device));
        SH_TRY(sl.queries.reserve((size_t)per * dim_ * 4));
        SH_TRY(sl.packed.reserve(lbytes));
        if (w > 1 && (use_rccl_ || r == 0)) SH_TRY(sl.gathered.reserve(lbytes * w));
        if (rq.allow && shards_[r]->rows) SH_TRY(sl.allow.reserve((size_t)((shards_[r]->rows + 63) / 64) * 8));
    }
    SH_HIP(hipSetDevice(root.device));
    SH_TRY(rs.out_rows.reserve(hbytes));
    SH_TRY(rs.out_scores.reserve(hbytes));
    SH_TRY(rs.out_counts.reserve(cbytes));
    rs.allow_slices.clear();
    if (rq.allow)
        for (uint32_t x = 0; x < row_shards_; ++x) {
            const std::vector<uint64_t> sl = shards_[x]->rows ? slice_bitmap(rq.allow, shards_[x]->lo, shards_[x]->rows) : std::vector<uint64_t>();
            rs.allow_slices.insert(rs.allow_slices.end(), sl.begin(), sl.end());
        }
    SH_T(1);
    SearchError first;
    for (uint32_t r = 0; r < w && first.ok(); ++r) first = enqueue_scan(rq, r, slot);
    SH_T(2);
    if (!first.ok()) {   // nothing has entered a collective yet; the searches that were begun are ended so that their tickets are free again
        uint32_t late = 0;
        (void)end_scans(slot, &late);
        return first;
    }
    {
        const SearchError xe = enqueue_exchange(slot);
        if (!xe.ok()) {   // the shards' begun searches are ended: each index has two tickets, a leaked one is gone for good
            uint32_t late = 0;
            (void)end_scans(slot, &late);
            return xe;
        }
    }
    SH_T(3);
    rs.pending = true;
    *ticket = next_ticket_++;
    return SearchError{};
}

SearchError ShardedIndex::end(uint64_t ticket, uint32_t* out_rows, float* out_scores, uint32_t* out_counts, uint32_t* fallbacks) {
    if (fallbacks) *fallbacks = 0;
    const int slot = (int)(ticket % kSlots);
    RootSlot& rs = root_[slot];
    if (!rs.pending || rs.ticket != ticket) return make_err(FSGPU_ERR_INVALID_CONFIG, "no search with this ticket is in flight");
    rs.pending = false;
    if (rs.k == 0) {   // k == 0, an empty index or an empty batch
        for (uint32_t q = 0; q < rs.nq; ++q) out_counts[q] = 0;
        return SearchError{};
    }
    if (rs.lone) return end_lone(rs, out_rows, out_scores, out_counts);
    SH_T0();
    hipError_t waited = hipSetDevice(shards_[0]->device);
    if (waited == hipSuccess) waited = hipEventSynchronize(rs.done);
    SH_T(4);
    // the ranks' verdicts; a rank that had to answer an uncertified query did so on its scan stream AFTER its list had travelled:
    // the corrected lists travel again (rare: the bench corpora never take this path).  The end halves run even when the wait
    // failed: a ticket that is never ended is lost to its index.
    uint32_t late = 0, exact_fallbacks = 0;
    const SearchError ended = end_scans(slot, &late, &exact_fallbacks);
    if (waited != hipSuccess) return hip_err(waited, "hipEventSynchronize(done)");
    SH_TRY(ended);
    SH_T(5);
    rs.fallbacks = exact_fallbacks;
    if (late) {
        for (auto& s : shards_) {
            SH_HIP(hipSetDevice(s->device));
            SH_HIP(hipEventRecord(s->slot[slot].scan_done, s->stream));
        }
        SH_TRY(enqueue_exchange(slot));
        SH_HIP(hipSetDevice(shards_[0]->device));
        SH_HIP(hipEventSynchronize(rs.done));
    }
    const size_t qbytes = (size_t)rs.nq * dim_ * 4, hbytes = (size_t)rs.nq * rs.k * 4, cbytes = (size_t)rs.nq * 4;
    const unsigned char* stage = static_cast<const unsigned char*>(rs.stage);
    std::memcpy(out_rows, stage + qbytes, hbytes);
    std::memcpy(out_scores, stage + qbytes + hbytes, hbytes);
    std::memcpy(out_counts, stage + qbytes + 2 * hbytes, cbytes);
    SH_T(6);
    SH_TDONE();
    if (fallbacks) *fallbacks = rs.fallbacks;
    return SearchError{};
}

SearchError ShardedIndex::search(const Request& rq, uint32_t query_len, uint32_t* out_rows, float* out_scores, uint32_t* out_counts,
                                 uint32_t* fallbacks) {
    uint64_t t = 0;
    SH_TRY(begin(rq, query_len, &t));
    return end(t, out_rows, out_scores, out_counts, fallbacks);
}

SearchError ShardedIndex::push_live_slices(const std::vector<uint64_t>& live) {
    for (auto& s : shards_) {
        if (!s->rows) continue;
        const std::vector<uint64_t> bits = slice_bitmap(live.data(), s->lo, s->rows);
        SH_TRY(s->index.set_live_bitmap(bits.data()));
    }
    return SearchError{};
}

SearchError ShardedIndex::set_live_bitmap(const uint64_t* live) {
    for (const RootSlot& r : root_)
        if (r.pending) return make_err(FSGPU_ERR_INVALID_CONFIG, "a search is in flight on this handle: end it first");
    if (!live) {
        for (auto& s : shards_) SH_TRY(s->index.set_live_bitmap(nullptr));
        if (catalog_) SH_TRY(catalog_->set_live_bitmap(nullptr));
        return SearchError{};
    }
    const std::vector<uint64_t> copy(live, live + (size_t)((nrows_ + 63) / 64));
    if (catalog_) SH_TRY(catalog_->set_live_bitmap(copy.data()));
    return push_live_slices(copy);
}

SearchError ShardedIndex::soft_delete(const char* doc_id, uint32_t len, int32_t* deleted) {
    if (!catalog_) return make_err(FSGPU_ERR_INVALID_CONFIG, "index has no doc-id table");
    // (the shards' scan streams do not wait for the blocking copy that rewrites a live bitmap: a begun search must be ended first)
    for (const RootSlot& r : root_)
        if (r.pending) return make_err(FSGPU_ERR_INVALID_CONFIG, "a search is in flight on this handle: end it first");
    SH_TRY(catalog_->soft_delete(doc_id, len, deleted));
    if (*deleted && !catalog_->live_host().empty()) return push_live_slices(catalog_->live_host());
    return SearchError{};
}

SearchError ShardedIndex::wal_append(const char* doc_id, uint32_t len, const float* vector, uint32_t vector_len) {
    if (!catalog_) return make_err(FSGPU_ERR_INVALID_CONFIG, "index has no doc-id table");
    for (const RootSlot& r : root_)
        if (r.pending) return make_err(FSGPU_ERR_INVALID_CONFIG, "a search is in flight on this handle: end it first");
    SH_TRY(catalog_->wal_append(doc_id, len, vector, vector_len));   // tombstones the main row it supersedes (lib.rs:2665-2710)
    if (!catalog_->live_host().empty()) return push_live_slices(catalog_->live_host());
    return SearchError{};
}

SearchError ShardedIndex::doc_id_at(uint32_t row, const char** ptr, uint32_t* len) const {
    if (!catalog_) return make_err(FSGPU_ERR_INVALID_CONFIG, "index has no doc-id table");
    return catalog_->doc_id_at(row, ptr, len);
}

SearchError ShardedIndex::search_hits(const float* query, uint32_t query_len, uint32_t k, uint32_t* out_rows, float* out_scores,
                                      uint32_t* out_count) {
    if (!catalog_) return make_err(FSGPU_ERR_INVALID_CONFIG, "index has no doc-id table");
    return catalog_->search_hits(query, query_len, k, out_rows, out_scores, out_count);   // its top-k comes from the shards
}

// dot_query_at over global rows: every row goes to the shard that owns it, one gather launch per shard touched.
SearchError ShardedIndex::gather_dot(const float* query, uint32_t query_len, const uint32_t* rows, uint32_t n, float* out) {
    if (query_len != dim_)
        return make_err(FSGPU_ERR_DIMENSION_MISMATCH, "expected " + std::to_string(dim_) + ", found " + std::to_string(query_len));
    std::vector<std::vector<uint32_t>> by_shard(shards_.size()), slot(shards_.size());
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t r = owner_of(rows[i]);
        if (r >= shards_.size()) return make_err(FSGPU_ERR_INVALID_CONFIG, "row index out of range for dot_query_at");
        by_shard[r].push_back(rows[i]);
        slot[r].push_back(i);
    }
    std::vector<float> tmp;
    for (uint32_t r = 0; r < shards_.size(); ++r) {
        if (by_shard[r].empty()) continue;
        tmp.resize(by_shard[r].size());
        std::lock_guard<std::mutex> lock(shards_[r]->index.mutex());
        SH_TRY(shards_[r]->index.gather_dot(query, query_len, by_shard[r].data(), (uint32_t)by_shard[r].size(), tmp.data()));
        for (size_t x = 0; x < tmp.size(); ++x) out[slot[r][x]] = tmp[x];
    }
    return SearchError{};
}

}  // namespace fsgpu
