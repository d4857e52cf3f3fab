// sharded_index.cpp — row-sharded search over the GPUs of one node behind one handle (sharded_index.hpp).
//
// Reference shape: scan_parallel's contiguous chunks + merge_partial_heaps
// (crates/frankensearch-index/src/search.rs:1013-1036,1704-1720); SURVEY §8e for the multi-GPU form.
// RCCL is bound at run time (dlopen of librccl.so.1): libfsgpu.so keeps linking only libamdhip64, a single-GPU host
// never needs RCCL on its library path, and inside a PyTorch process the already loaded copy (same SONAME) is reused.
#include "sharded_index.hpp"

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
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.comm_init_all || !r.all_gather || !r.comm_destroy || !r.error_string) {
            r.why = "librccl.so.1 lacks ncclCommInitAll / ncclAllGather / ncclCommDestroy / ncclGetErrorString";
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

}  // namespace

ShardedIndex::~ShardedIndex() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
        ++generation_;
    }
    cv_work_.notify_all();
    for (auto& s : shards_)
        if (s->worker.joinable()) s->worker.join();
    for (auto& s : shards_) {
        if (s->device >= 0) (void)hipSetDevice(s->device);
        if (s->comm && Rccl::get().ok()) (void)Rccl::get().comm_destroy(static_cast<ncclComm_t>(s->comm));
        if (s->stream) (void)hipStreamDestroy(s->stream);
        for (DeviceBuffer* b : {&s->queries, &s->packed, &s->gathered, &s->out_rows, &s->out_scores, &s->out_counts}) b->release();
    }
    if (stage_host_) (void)hipHostFree(stage_host_);
}

SearchError ShardedIndex::init_host(const int32_t* devices, uint32_t ndev, uint32_t dim, uint64_t nrows, const void* slab_f16,
                                    const uint64_t* live, int32_t exchange) {
    if (!devices || ndev == 0) return make_err(FSGPU_ERR_INVALID_CONFIG, "at least one device is required");
    if (dim == 0) return make_err(FSGPU_ERR_INVALID_CONFIG, "dimension must be greater than zero");
    if (nrows >= 0xffffffffull) return make_err(FSGPU_ERR_INVALID_CONFIG, "row ids must fit in u32 (VectorHit.index)");
    if (nrows > 0 && !slab_f16) return make_err(FSGPU_ERR_NULL_ARGUMENT, "slab is null");
    dim_ = dim;
    nrows_ = nrows;
    // contiguous ceil split, exactly the chunking of scan_parallel (search.rs:1020-1035) at shard granularity
    const uint64_t per = (nrows + ndev - 1) / ndev;
    for (uint32_t r = 0; r < ndev; ++r) {
        auto s = std::make_unique<Shard>();
        s->device = devices[r];
        s->lo = std::min<uint64_t>(nrows, (uint64_t)r * per);
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
                                      const void* const* slabs_dev, const uint64_t* const* live_dev, int32_t exchange) {
    if (!devices || ndev == 0 || !shard_rows || !slabs_dev)
        return make_err(FSGPU_ERR_INVALID_CONFIG, "devices, shard_rows and slabs are required");
    if (dim == 0) return make_err(FSGPU_ERR_INVALID_CONFIG, "dimension must be greater than zero");
    dim_ = dim;
    uint64_t lo = 0;
    for (uint32_t r = 0; r < ndev; ++r) {
        auto s = std::make_unique<Shard>();
        s->device = devices[r];
        s->lo = lo;
        s->rows = shard_rows[r];
        lo += s->rows;
        if (lo >= 0xffffffffull) return make_err(FSGPU_ERR_INVALID_CONFIG, "row ids must fit in u32 (VectorHit.index)");
        SH_TRY(s->index.init_device(s->device, dim, s->rows, slabs_dev[r], live_dev ? live_dev[r] : nullptr, s->lo));
        shards_.push_back(std::move(s));
    }
    nrows_ = lo;
    return finish_init(exchange);
}

SearchError ShardedIndex::finish_init(int32_t exchange) {
    if (exchange < 0 || exchange > 2) return make_err(FSGPU_ERR_INVALID_CONFIG, "exchange must be 0 (auto), 1 (RCCL) or 2 (peer copies)");
    const uint32_t w = (uint32_t)shards_.size();
    bool distinct = true;
    for (uint32_t a = 0; a < w; ++a)
        for (uint32_t b = a + 1; b < w; ++b) distinct &= shards_[a]->device != shards_[b]->device;
    for (auto& s : shards_) {
        SH_HIP(hipSetDevice(s->device));
        SH_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    }
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
    if (!use_rccl_ && w > 1) {
        // peer copies: let every shard's device write into the root's gather buffer
        for (uint32_t r = 1; r < w; ++r) {
            if (shards_[r]->device == shards_[0]->device) continue;
            int can = 0;
            SH_HIP(hipDeviceCanAccessPeer(&can, shards_[r]->device, shards_[0]->device));
            if (can) {
                SH_HIP(hipSetDevice(shards_[r]->device));
                const hipError_t e = hipDeviceEnablePeerAccess(shards_[0]->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return hip_err(e, "hipDeviceEnablePeerAccess");
                (void)hipGetLastError();
            }
        }
    }
    for (uint32_t r = 0; r < w; ++r) shards_[r]->worker = std::thread([this, r] { worker_main(r); });
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
}

// One host thread per shard: HIP's current device is per thread, the batched search synchronises its stream, and
// RCCL's single-process mode wants one caller per rank.
void ShardedIndex::worker_main(uint32_t r) {
    Shard& s = *shards_[r];
    (void)hipSetDevice(s.device);
    uint64_t seen = 0;
    for (;;) {
        int phase;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_work_.wait(lk, [&] { return generation_ != seen; });
            seen = generation_;
            if (stop_) return;
            phase = phase_;
        }
        SearchError e;
        try {
            e = phase == 1 ? shard_search(s) : shard_exchange(r);
        } catch (const std::exception& ex) {
            e = make_err(FSGPU_ERR_DEVICE, ex.what());
        } catch (...) {
            e = make_err(FSGPU_ERR_DEVICE, "unknown exception in a shard worker");
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!e.ok() && s.error.ok()) s.error = e;
            if (--pending_ == 0) cv_done_.notify_one();
        }
    }
}

void ShardedIndex::run_phase(int phase) {
    std::unique_lock<std::mutex> lk(mu_);
    phase_ = phase;
    pending_ = (uint32_t)shards_.size();
    ++generation_;
    cv_work_.notify_all();
    cv_done_.wait(lk, [&] { return pending_ == 0; });
}

// Phase 1: this shard's packed best-first lists [nq, k] (global row ids, ~0 padding), stream drained on return.
SearchError ShardedIndex::shard_search(Shard& s) {
    const Job& j = job_;
    const size_t qbytes = (size_t)j.nq * dim_ * 4, lbytes = (size_t)j.nq * j.k * 8;
    SH_HIP(hipSetDevice(s.device));
    SH_TRY(s.queries.reserve(qbytes));
    SH_TRY(s.packed.reserve(lbytes));
    SH_HIP(hipMemcpyAsync(s.queries.ptr, j.queries, qbytes, hipMemcpyHostToDevice, s.stream));
    s.fallbacks = 0;
    if (s.rows == 0) {
        SH_HIP(hipMemsetAsync(s.packed.ptr, 0xff, lbytes, s.stream));
    } else if (j.batched) {
        SH_TRY(s.index.search_top_k_batched_device(static_cast<const float*>(s.queries.ptr), j.nq, dim_, j.k, nullptr, nullptr,
                                                   nullptr, nullptr, s.stream, &s.fallbacks,
                                                   static_cast<uint64_t*>(s.packed.ptr)));
    } else {
        SH_TRY(s.index.search_top_k_packed_device(static_cast<const float*>(s.queries.ptr), j.nq, dim_, j.k, nullptr,
                                                  static_cast<uint64_t*>(s.packed.ptr), s.stream));
    }
    SH_HIP(hipStreamSynchronize(s.stream));
    return SearchError{};
}

// Phase 2 (entered only when every shard's phase 1 succeeded — a rank missing from the collective would hang the
// others): the W lists land in gather layout [W][nq][k] — on every device through ncclAllGather, or on the root only
// through device-to-device copies.
SearchError ShardedIndex::shard_exchange(uint32_t r) {
    Shard& s = *shards_[r];
    const Job& j = job_;
    const uint32_t w = (uint32_t)shards_.size();
    const size_t count = (size_t)j.nq * j.k, lbytes = count * 8;
    SH_HIP(hipSetDevice(s.device));
    if (use_rccl_) {
        SH_TRY(s.gathered.reserve(lbytes * w));
        Rccl& rc = Rccl::get();
        const ncclResult_t st = rc.all_gather(s.packed.ptr, s.gathered.ptr, count, ncclUint64, static_cast<ncclComm_t>(s.comm), s.stream);
        if (st != ncclSuccess) return make_err(FSGPU_ERR_DEVICE, std::string("ncclAllGather: ") + rc.error_string(st));
    } else {
        Shard& root = *shards_[0];
        SH_HIP(hipMemcpyAsync(static_cast<unsigned char*>(root.gathered.ptr) + (size_t)r * lbytes, s.packed.ptr, lbytes,
                              hipMemcpyDeviceToDevice, s.stream));
    }
    SH_HIP(hipStreamSynchronize(s.stream));
    return SearchError{};
}

SearchError ShardedIndex::search(const float* queries, uint32_t nq, uint32_t query_len, uint32_t k, bool batched,
                                 uint32_t* out_rows, float* out_scores, uint32_t* out_counts, uint32_t* fallbacks) {
    if (fallbacks) *fallbacks = 0;
    if (query_len != dim_)
        return make_err(FSGPU_ERR_DIMENSION_MISMATCH, "expected " + std::to_string(dim_) + ", found " + std::to_string(query_len));
    if (nq == 0) return SearchError{};
    if (k == 0 || nrows_ == 0) {
        for (uint32_t q = 0; q < nq; ++q) out_counts[q] = 0;
        return SearchError{};
    }
    if (dim_ % 8 != 0 || k > 256)
        return make_err(FSGPU_ERR_INVALID_CONFIG, "the sharded search exchanges the fused tiers' packed lists: k <= 256 and dim % 8 == 0");
    const uint32_t w = (uint32_t)shards_.size();
    Shard& root = *shards_[0];
    // pinned staging: [queries | rows | scores | counts]
    const size_t qbytes = (size_t)nq * dim_ * 4, hbytes = (size_t)nq * k * 4, cbytes = (size_t)nq * 4;
    const size_t need = qbytes + 2 * hbytes + cbytes;
    SH_HIP(hipSetDevice(root.device));
    if (need > stage_bytes_) {
        if (stage_host_) (void)hipHostFree(stage_host_);
        stage_host_ = nullptr;
        stage_bytes_ = 0;
        SH_HIP(hipHostMalloc(&stage_host_, need, hipHostMallocPortable));
        stage_bytes_ = need;
    }
    unsigned char* stage = static_cast<unsigned char*>(stage_host_);
    std::memcpy(stage, queries, qbytes);
    job_.queries = reinterpret_cast<const float*>(stage);
    job_.nq = nq;
    job_.k = k;
    job_.batched = batched;
    for (auto& s : shards_) s->error = SearchError{};
    if (!use_rccl_) SH_TRY(root.gathered.reserve((size_t)nq * k * 8 * w));  // before any shard copies into it
    run_phase(1);
    for (auto& s : shards_)
        if (!s->error.ok()) return s->error;
    if (w > 1) {
        run_phase(2);
        for (auto& s : shards_)
            if (!s->error.ok()) return s->error;
    }
    // merge_partial_heaps across shards (search.rs:1704-1720) on the root; one shard: its own list is the answer,
    // the merge only unpacks it
    SH_TRY(root.out_rows.reserve(hbytes));
    SH_TRY(root.out_scores.reserve(hbytes));
    SH_TRY(root.out_counts.reserve(cbytes));
    const uint64_t* lists = static_cast<const uint64_t*>(w > 1 ? root.gathered.ptr : root.packed.ptr);
    SH_TRY(merge_packed_lists_device(root.device, lists, nq, w, k, k, (uint64_t)nq * k, k, static_cast<uint32_t*>(root.out_rows.ptr),
                                     static_cast<float*>(root.out_scores.ptr), static_cast<uint32_t*>(root.out_counts.ptr),
                                     root.stream));
    unsigned char* h_rows = stage + qbytes;
    unsigned char* h_scores = h_rows + hbytes;
    unsigned char* h_counts = h_scores + hbytes;
    SH_HIP(hipMemcpyAsync(h_rows, root.out_rows.ptr, hbytes, hipMemcpyDeviceToHost, root.stream));
    SH_HIP(hipMemcpyAsync(h_scores, root.out_scores.ptr, hbytes, hipMemcpyDeviceToHost, root.stream));
    SH_HIP(hipMemcpyAsync(h_counts, root.out_counts.ptr, cbytes, hipMemcpyDeviceToHost, root.stream));
    SH_HIP(hipStreamSynchronize(root.stream));
    std::memcpy(out_rows, h_rows, hbytes);
    std::memcpy(out_scores, h_scores, hbytes);
    std::memcpy(out_counts, h_counts, cbytes);
    if (fallbacks)
        for (auto& s : shards_) *fallbacks += s->fallbacks;
    return SearchError{};
}

}  // namespace fsgpu
