// fsgpu_api.cpp — the extern "C" boundary of libfsgpu.so (declared in include/fsgpu.h).
// Plain pointers and sizes only; every entry point catches C++ exceptions and reports a status.
#include "../../include/fsgpu.h"
#include "../../include/fsgpu_lab.h"

#include <cmath>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include <atomic>
#include <mutex>
#include <shared_mutex>

#include "bert_embedder.hpp"
#include "coalescer.hpp"
#include "sharded_index.hpp"
#include "vector_index.hpp"
#include "two_tier_index.hpp"

// One in-flight single-item call, parked in a Coalescer until a leader serves it (coalescer.hpp).
struct CoalescedCall : fsgpu::CoalescedRequest {
    fsgpu_status status = FSGPU_OK;
    std::string detail;
};
struct SearchCall : CoalescedCall {
    const float* query = nullptr;
    uint32_t k = 0;
    uint32_t int8_mult = 0;  // 0 = exact search, else search_top_k_int8_two_pass with this multiplier
    const uint64_t* allow = nullptr;  // the caller's allow bitmap (host words): calls that pass the SAME bitmap share a batch
    const uint64_t* allow_dev = nullptr;  // its resident device copy (fsgpu_allow_bitmap), if the caller keeps one
    uint32_t* out_rows = nullptr;
    float* out_scores = nullptr;
    uint32_t* out_count = nullptr;
};
template <class Id>
struct EmbedCall : CoalescedCall {
    const Id* ids = nullptr;
    uint32_t len = 0;
    float* out = nullptr;
};

struct fsgpu_index {
    fsgpu::VectorIndex impl;
    // Lanes (vector_index.hpp): row-level searches from different host threads run on replicas of the index, each behind
    // its own mutex; state_mu is held shared by a search on a replica and exclusively by whatever changes the index
    // (tombstones, WAL, live bitmap, hreduce / variant), which then re-syncs the replicas.
    std::shared_mutex state_mu;
    std::mutex lanes_mu;
    std::atomic<bool> lanes_ready{false};
    std::atomic<uint32_t> lane_rr{0};
    fsgpu::Coalescer<SearchCall> coalescer;
    // leader-only staging (guarded by impl.mutex())
    std::vector<float> co_queries, co_scores;
    std::vector<uint32_t> co_rows, co_counts;
};
struct fsgpu_alignment {
    fsgpu::QualityAlignment impl;
};
struct fsgpu_allow_bitmap {   // a precomputed SearchFilter resident on an index's device
    int device = -1;
    uint64_t nrows = 0, allowed = 0;
    std::vector<uint64_t> words;   // host copy: the selectivity rule (1/50) and the coalescer's batch key
    fsgpu::DeviceBuffer dev;
};
struct fsgpu_sharded {
    fsgpu::ShardedIndex impl;
    // single-query fsgpu_sharded_search calls in flight together ride one batched search of the shards (fsgpu_sharded_set_coalescing)
    fsgpu::Coalescer<SearchCall> coalescer;
    std::vector<float> co_queries, co_scores;   // leader-only staging (guarded by impl.mutex())
    std::vector<uint32_t> co_rows, co_counts;
};
struct fsgpu_m2v {
    fsgpu::Model2VecEmbedder impl;
    fsgpu::Coalescer<EmbedCall<uint32_t>> coalescer;
    std::mutex co_mu;
    std::vector<uint32_t> co_ids, co_offsets;
    std::vector<float> co_out;
};
struct fsgpu_bert {
    fsgpu::NativeEmbedder impl;
    fsgpu::Coalescer<EmbedCall<int32_t>> coalescer;
    std::mutex co_mu;
    std::vector<int32_t> co_ids;
    std::vector<uint32_t> co_offsets;
    std::vector<float> co_out;
};

namespace {

thread_local std::string g_last_error;

fsgpu_status finish(const fsgpu::SearchError& e) {
    if (!e.ok()) g_last_error = e.detail;
    return e.code;
}

fsgpu_status fail(fsgpu_status code, const char* detail) {
    g_last_error = detail;
    return code;
}

template <typename F>
fsgpu_status guarded(F&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(FSGPU_ERR_DEVICE, "host allocation failed");
    } catch (const std::exception& ex) {
        g_last_error = ex.what();
        return FSGPU_ERR_DEVICE;
    } catch (...) {
        return fail(FSGPU_ERR_DEVICE, "unknown exception");
    }
}

// Runs one coalesced batch of single-text embed calls through embed_batch (ids concatenated, offsets rebuilt).
template <class Handle, class Id>
void run_embed_batch(Handle* h, uint32_t dim, std::vector<EmbedCall<Id>*>& batch) {
    std::lock_guard<std::mutex> lock(h->co_mu);
    fsgpu_status st = FSGPU_ERR_DEVICE;
    std::string detail;
    try {
        h->co_ids.clear();
        h->co_offsets.assign(1, 0u);
        for (auto* c : batch) {
            h->co_ids.insert(h->co_ids.end(), c->ids, c->ids + c->len);
            h->co_offsets.push_back((uint32_t)h->co_ids.size());
        }
        h->co_out.resize((size_t)batch.size() * dim);
        fsgpu::SearchError e = h->impl.embed_batch(h->co_ids.data(), h->co_offsets.data(), (uint32_t)batch.size(), h->co_out.data());
        st = e.code;
        detail = e.detail;
    } catch (const std::exception& ex) {
        detail = ex.what();
    } catch (...) {
        detail = "unknown exception";
    }
    for (size_t i = 0; i < batch.size(); ++i) {
        batch[i]->status = st;
        batch[i]->detail = detail;
        if (st == FSGPU_OK) std::memcpy(batch[i]->out, h->co_out.data() + i * dim, (size_t)dim * 4);
    }
}

// One single-query call parked in the index's coalescer: concurrent callers ride one batched pass (results are
// bit-identical to the direct path).  int8_mult 0 = exact search, else the int8 two-pass with that multiplier.
fsgpu_status coalesced_search(fsgpu_index* idx, const float* query, uint32_t k, uint32_t int8_mult, uint32_t* out_rows,
                              float* out_scores, uint32_t* out_count, const uint64_t* allow = nullptr, const uint64_t* allow_dev = nullptr) {
    return guarded([&]() -> fsgpu_status {
        SearchCall call;
        call.query = query;
        call.k = k;
        call.int8_mult = int8_mult;
        call.allow = allow;
        call.allow_dev = allow_dev;
        call.out_rows = out_rows;
        call.out_scores = out_scores;
        call.out_count = out_count;
        idx->coalescer.submit(
            &call,
            [idx](std::vector<SearchCall*>& batch) {
                std::lock_guard<std::mutex> lock(idx->impl.mutex());
                const uint32_t n = (uint32_t)batch.size(), dim = idx->impl.dimension(), kk = batch[0]->k;
                const uint32_t mult = batch[0]->int8_mult;
                const uint64_t* allow_bm = batch[0]->allow;   // one filter for the whole batch (compatible() below)
                const uint64_t* allow_res = batch[0]->allow_dev;
                fsgpu_status st = FSGPU_ERR_DEVICE;
                std::string detail;
                try {
                    idx->co_queries.resize((size_t)n * dim);
                    idx->co_rows.resize((size_t)n * kk);
                    idx->co_scores.resize((size_t)n * kk);
                    idx->co_counts.resize(n);
                    for (uint32_t i = 0; i < n; ++i)
                        std::memcpy(idx->co_queries.data() + (size_t)i * dim, batch[i]->query, (size_t)dim * 4);
                    uint32_t fb = 0;
                    fsgpu::SearchError e;
                    if (mult) {
                        e = idx->impl.search_top_k_int8_batched(idx->co_queries.data(), n, dim, kk, mult, idx->co_rows.data(),
                                                                idx->co_scores.data(), idx->co_counts.data(), &fb);
                    } else if (n <= 4) {
                        // up to four callers: one pass of the exact multi-query kernel is quicker than the staged
                        // matrix-core pipeline; beyond that the batched path serves 128 and more per pass
                        e = idx->impl.search_top_k(idx->co_queries.data(), n, dim, kk, allow_bm, idx->co_rows.data(),
                                                   idx->co_scores.data(), idx->co_counts.data(), allow_res);
                    } else {
                        e = idx->impl.search_top_k_batched(idx->co_queries.data(), n, dim, kk, allow_bm, idx->co_rows.data(),
                                                           idx->co_scores.data(), idx->co_counts.data(), &fb, allow_res);
                    }
                    st = e.code;
                    detail = e.detail;
                } catch (const std::exception& ex) {
                    detail = ex.what();
                } catch (...) {
                    detail = "unknown exception";
                }
                for (uint32_t i = 0; i < n; ++i) {
                    batch[i]->status = st;
                    batch[i]->detail = detail;
                    if (st != FSGPU_OK) continue;
                    std::memcpy(batch[i]->out_rows, idx->co_rows.data() + (size_t)i * kk, (size_t)kk * 4);
                    std::memcpy(batch[i]->out_scores, idx->co_scores.data() + (size_t)i * kk, (size_t)kk * 4);
                    *batch[i]->out_count = idx->co_counts[i];
                }
            },
            [](const SearchCall& a, const SearchCall& b) { return a.k == b.k && a.int8_mult == b.int8_mult && a.allow == b.allow; });
        if (call.exec_threw) return fail(FSGPU_ERR_DEVICE, "coalesced batch failed before this request was served");
        if (call.status != FSGPU_OK) g_last_error = call.detail;
        return call.status;
    });
}

// The sharded twin of coalesced_search: one single-query request (exact, or the int8 two-pass with call.int8_mult) parked in the
// handle's coalescer; the leader runs ONE search of the shards for the whole batch — more than four exact callers take the
// matrix-core batched mode, whose rows and score bits are the exact kernels' — and hands every caller its own hits.
fsgpu_status coalesced_sharded_search(fsgpu_sharded* idx, const float* query, uint32_t query_len, uint32_t k, uint32_t int8_mult,
                                      uint32_t* out_rows, float* out_scores, uint32_t* out_count) {
    return guarded([&]() -> fsgpu_status {
        SearchCall call;
        call.query = query;
        call.k = k;
        call.int8_mult = int8_mult;
        call.out_rows = out_rows;
        call.out_scores = out_scores;
        call.out_count = out_count;
        idx->coalescer.submit(
            &call,
            [idx, query_len](std::vector<SearchCall*>& batch) {
                std::lock_guard<std::mutex> lock(idx->impl.mutex());
                const uint32_t n = (uint32_t)batch.size(), dim = idx->impl.dimension(), kk = batch[0]->k;
                fsgpu_status st = FSGPU_ERR_DEVICE;
                std::string detail;
                try {
                    idx->co_queries.resize((size_t)n * dim);
                    idx->co_rows.resize((size_t)n * kk);
                    idx->co_scores.resize((size_t)n * kk);
                    idx->co_counts.resize(n);
                    for (uint32_t i = 0; i < n; ++i)
                        std::memcpy(idx->co_queries.data() + (size_t)i * dim, batch[i]->query, (size_t)dim * 4);
                    fsgpu::ShardedIndex::Request rq;
                    rq.queries = idx->co_queries.data();
                    rq.nq = n;
                    rq.k = kk;
                    rq.multiplier = batch[0]->int8_mult;
                    rq.mode = batch[0]->int8_mult ? fsgpu::ShardedIndex::kInt8TwoPass
                              : n <= 4            ? fsgpu::ShardedIndex::kExact
                                                  : fsgpu::ShardedIndex::kBatched;
                    const fsgpu::SearchError e =
                        idx->impl.search(rq, query_len, idx->co_rows.data(), idx->co_scores.data(), idx->co_counts.data(), nullptr);
                    st = e.code;
                    detail = e.detail;
                } catch (const std::exception& ex) {
                    detail = ex.what();
                } catch (...) {
                    detail = "unknown exception";
                }
                for (uint32_t i = 0; i < n; ++i) {
                    batch[i]->status = st;
                    batch[i]->detail = detail;
                    if (st != FSGPU_OK) continue;
                    std::memcpy(batch[i]->out_rows, idx->co_rows.data() + (size_t)i * kk, (size_t)kk * 4);
                    std::memcpy(batch[i]->out_scores, idx->co_scores.data() + (size_t)i * kk, (size_t)kk * 4);
                    *batch[i]->out_count = idx->co_counts[i];
                }
            },
            [](const SearchCall& a, const SearchCall& b) { return a.k == b.k && a.int8_mult == b.int8_mult; });
        if (call.exec_threw) return fail(FSGPU_ERR_DEVICE, "coalesced batch failed before this request was served");
        if (call.status != FSGPU_OK) g_last_error = call.detail;
        return call.status;
    });
}

}  // namespace

extern "C" {

const char* fsgpu_version(void) { return "fsgpu 0.1.0 (gfx950)"; }

int32_t fsgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* fsgpu_last_error(void) { return g_last_error.c_str(); }

const char* fsgpu_last_main_pass_kernel(void) { return fsgpu::last_main_pass_kernel(); }

fsgpu_status fsgpu_index_create(int32_t device, uint32_t dim, uint64_t nrows, const void* slab_f16_le,
                                const uint64_t* live_bitmap, uint64_t row_base, fsgpu_index** out) {
    if (!out) return fail(FSGPU_ERR_NULL_ARGUMENT, "out is null");
    *out = nullptr;
    return guarded([&]() -> fsgpu_status {
        auto* h = new fsgpu_index();
        fsgpu::SearchError e = h->impl.init_host(device, dim, nrows, slab_f16_le, live_bitmap, row_base);
        if (!e.ok()) {
            delete h;
            return finish(e);
        }
        *out = h;
        return FSGPU_OK;
    });
}

fsgpu_status fsgpu_index_create_device(int32_t device, uint32_t dim, uint64_t nrows, const void* slab_f16_dev,
                                       const uint64_t* live_bitmap_dev, uint64_t row_base, fsgpu_index** out) {
    if (!out) return fail(FSGPU_ERR_NULL_ARGUMENT, "out is null");
    *out = nullptr;
    return guarded([&]() -> fsgpu_status {
        auto* h = new fsgpu_index();
        fsgpu::SearchError e = h->impl.init_device(device, dim, nrows, slab_f16_dev, live_bitmap_dev, row_base);
        if (!e.ok()) {
            delete h;
            return finish(e);
        }
        *out = h;
        return FSGPU_OK;
    });
}

fsgpu_status fsgpu_index_open_fsvi(const char* path, int32_t device, fsgpu_index** out) {
    if (!out) return fail(FSGPU_ERR_NULL_ARGUMENT, "out is null");
    *out = nullptr;
    return guarded([&]() -> fsgpu_status {
        auto* h = new fsgpu_index();
        fsgpu::SearchError e = h->impl.open_fsvi(path, device);
        if (!e.ok()) {
            delete h;
            return finish(e);
        }
        *out = h;
        return FSGPU_OK;
    });
}

void fsgpu_index_destroy(fsgpu_index* idx) { delete idx; }

uint64_t fsgpu_index_record_count(const fsgpu_index* idx) { return idx ? idx->impl.record_count() : 0; }
uint32_t fsgpu_index_dimension(const fsgpu_index* idx) { return idx ? idx->impl.dimension() : 0; }

fsgpu_status fsgpu_index_set_hreduce(fsgpu_index* idx, int32_t mode) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (mode < FSGPU_HREDUCE_SSE2 || mode > FSGPU_HREDUCE_SEQ) return fail(FSGPU_ERR_INVALID_CONFIG, "unknown hreduce mode");
    std::unique_lock<std::shared_mutex> state(idx->state_mu);
    std::lock_guard<std::mutex> lock(idx->impl.mutex());
    idx->impl.hreduce = mode;
    idx->impl.sync_replicas();
    return FSGPU_OK;
}

fsgpu_status fsgpu_index_set_batched_filter(fsgpu_index* idx, int32_t filter) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (filter < FSGPU_FILTER_AUTO || filter > FSGPU_FILTER_INT8) return fail(FSGPU_ERR_INVALID_CONFIG, "unknown batched filter");
    std::unique_lock<std::shared_mutex> state(idx->state_mu);
    std::lock_guard<std::mutex> lock(idx->impl.mutex());
    idx->impl.batched_filter = filter;
    return FSGPU_OK;
}

fsgpu_status fsgpu_index_set_filter_rotation(fsgpu_index* idx, int32_t mode) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (mode < FSGPU_ROTATION_AUTO || mode > FSGPU_ROTATION_ON) return fail(FSGPU_ERR_INVALID_CONFIG, "unknown rotation mode");
    std::unique_lock<std::shared_mutex> state(idx->state_mu);
    std::lock_guard<std::mutex> lock(idx->impl.mutex());
    idx->impl.filter_rotation = mode;
    return FSGPU_OK;
}

int32_t fsgpu_index_filter_rotated(fsgpu_index* idx) {
    if (!idx) return 0;
    std::lock_guard<std::mutex> lock(idx->impl.mutex());
    return idx->impl.filter_rotated() ? 1 : 0;
}

fsgpu_status fsgpu_index_set_int8_latency(fsgpu_index* idx, int32_t enabled) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    std::unique_lock<std::shared_mutex> state(idx->state_mu);
    std::lock_guard<std::mutex> lock(idx->impl.mutex());
    idx->impl.int8_latency = enabled != 0;
    if (enabled == FSGPU_INT8_LATENCY_BUILD_NOW) return finish(idx->impl.prepare_int8_latency());
    return FSGPU_OK;
}

fsgpu_status fsgpu_index_batched_filter_stats(fsgpu_index* idx, uint64_t* int8_queries, uint64_t* refiltered_f16,
                                              int32_t* int8_active) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    std::unique_lock<std::shared_mutex> state(idx->state_mu);
    std::lock_guard<std::mutex> lock(idx->impl.mutex());
    if (int8_queries) *int8_queries = idx->impl.i8f_queries;
    if (refiltered_f16) *refiltered_f16 = idx->impl.i8f_refiltered;
    if (int8_active) *int8_active = idx->impl.int8_filter_active() ? 1 : 0;
    return FSGPU_OK;
}

fsgpu_status fsgpu_index_int8_filter_bound(fsgpu_index* idx, const float* queries, uint32_t nq, uint32_t query_len, float* out_delta,
                                           float* out_query_scale, float* out_slab_scale, int8_t* out_queries_i8, int8_t* out_slab_i8) {
    if (!idx || (nq && !queries)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::unique_lock<std::shared_mutex> state(idx->state_mu);
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.int8_filter_bound(queries, nq, query_len, out_delta, out_query_scale, out_slab_scale, out_queries_i8,
                                                  out_slab_i8));
    });
}

fsgpu_status fsgpu_index_doc_id(const fsgpu_index* idx, uint32_t row, const char** ptr, uint32_t* len) {
    if (!idx || !ptr || !len) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return finish(idx->impl.doc_id_at(row, ptr, len));
}

fsgpu_status fsgpu_index_soft_delete(fsgpu_index* idx, const char* doc_id, uint32_t doc_id_len, int32_t* deleted) {
    if (!idx || !doc_id || !deleted) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::unique_lock<std::shared_mutex> state(idx->state_mu);
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        const fsgpu::SearchError e = idx->impl.soft_delete(doc_id, doc_id_len, deleted);
        idx->impl.sync_replicas();   // the live bitmap may have been re-uploaded
        return finish(e);
    });
}

fsgpu_status fsgpu_index_allow_bitmap_for_hashes(const fsgpu_index* idx, const uint64_t* hashes, uint32_t n,
                                                 uint64_t* allow_bitmap_out, uint64_t* rows_matched) {
    if (!idx || (!hashes && n) || !allow_bitmap_out) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(const_cast<fsgpu_index*>(idx)->impl.mutex());
        return finish(idx->impl.allow_bitmap_for_hashes(hashes, n, allow_bitmap_out, rows_matched));
    });
}

fsgpu_status fsgpu_index_set_live_bitmap(fsgpu_index* idx, const uint64_t* live_bitmap) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    return guarded([&]() -> fsgpu_status {
        std::unique_lock<std::shared_mutex> state(idx->state_mu);
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        const fsgpu::SearchError e = idx->impl.set_live_bitmap(live_bitmap);
        idx->impl.sync_replicas();
        return finish(e);
    });
}

static fsgpu_status search_topk_common(fsgpu_index* idx, const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                       const uint64_t* allow_bitmap, const uint64_t* allow_resident_dev, uint32_t* out_rows,
                                       float* out_scores, uint32_t* out_counts, bool exact_only = false) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (nq && (!queries || !out_counts || (k && (!out_rows || !out_scores))))
        return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (!exact_only && idx->coalescer.enabled() && nq == 1 && k >= 1 && k <= 64 && query_len == idx->impl.dimension())
        return coalesced_search(idx, queries, k, 0, out_rows, out_scores, out_counts, allow_bitmap, allow_resident_dev);
    return guarded([&]() -> fsgpu_status {
        // this index if it is free, else a free replica, else queue on one of the lanes in turn
        std::shared_lock<std::shared_mutex> state(idx->state_mu);
        std::unique_lock<std::mutex> lock(idx->impl.mutex(), std::try_to_lock);
        fsgpu::VectorIndex* lane = &idx->impl;
        if (!lock.owns_lock()) {
            if (!idx->lanes_ready.load(std::memory_order_acquire)) {
                std::lock_guard<std::mutex> init(idx->lanes_mu);
                if (!idx->lanes_ready.load(std::memory_order_relaxed)) {
                    std::lock_guard<std::mutex> primary(idx->impl.mutex());   // the replicas copy the primary's state
                    fsgpu::SearchError e = idx->impl.ensure_replicas();
                    if (!e.ok()) return finish(e);
                    idx->lanes_ready.store(true, std::memory_order_release);
                }
            }
            const size_t n = idx->impl.replica_count();
            for (size_t i = 0; i < n && !lock.owns_lock(); ++i) {
                lock = std::unique_lock<std::mutex>(idx->impl.replica(i)->mutex(), std::try_to_lock);
                if (lock.owns_lock()) lane = idx->impl.replica(i);
            }
            if (!lock.owns_lock()) {
                const uint32_t pick = idx->lane_rr.fetch_add(1, std::memory_order_relaxed) % (uint32_t)(n + 1);
                lane = pick == 0 ? &idx->impl : idx->impl.replica(pick - 1);
                lock = std::unique_lock<std::mutex>(lane->mutex());
            }
        }
        lane->exact_only_ = exact_only;   // (under the lane's mutex)
        const fsgpu::SearchError e = lane->search_top_k(queries, nq, query_len, k, allow_bitmap, out_rows, out_scores, out_counts, allow_resident_dev);
        lane->exact_only_ = false;
        return finish(e);
    });
}

fsgpu_status fsgpu_search_topk_exact(fsgpu_index* idx, const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                     const uint64_t* allow_bitmap, uint32_t* out_rows, float* out_scores, uint32_t* out_counts) {
    return search_topk_common(idx, queries, nq, query_len, k, allow_bitmap, nullptr, out_rows, out_scores, out_counts, true);
}

fsgpu_status fsgpu_search_topk(fsgpu_index* idx, const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                               const uint64_t* allow_bitmap, uint32_t* out_rows, float* out_scores,
                               uint32_t* out_counts) {
    return search_topk_common(idx, queries, nq, query_len, k, allow_bitmap, nullptr, out_rows, out_scores, out_counts);
}

// ---- resident filters: a precomputed SearchFilter uploaded once and reused (filter.rs:19-56; search.rs:1114-1255) ----
fsgpu_status fsgpu_allow_bitmap_create(fsgpu_index* idx, const uint64_t* allow_bitmap, fsgpu_allow_bitmap** out) {
    if (!idx || !allow_bitmap || !out) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    *out = nullptr;
    return guarded([&]() -> fsgpu_status {
        auto f = std::make_unique<fsgpu_allow_bitmap>();
        f->device = idx->impl.device();
        f->nrows = idx->impl.record_count();
        const size_t words = (size_t)((f->nrows + 63) / 64);
        f->words.assign(allow_bitmap, allow_bitmap + words);
        if (words && (f->nrows & 63)) f->words.back() &= (1ull << (f->nrows & 63)) - 1ull;
        for (uint64_t w : f->words) f->allowed += (uint64_t)__builtin_popcountll(w);
        if (f->device < 0) return fail(FSGPU_ERR_NO_DEVICE, "index has no device");
        if (hipSetDevice(f->device) != hipSuccess) return fail(FSGPU_ERR_DEVICE, "hipSetDevice failed");
        fsgpu::SearchError e = f->dev.reserve(std::max<size_t>(words, 1) * 8);
        if (!e.ok()) return finish(e);
        if (words && hipMemcpy(f->dev.ptr, f->words.data(), words * 8, hipMemcpyHostToDevice) != hipSuccess) {
            f->dev.release();
            return fail(FSGPU_ERR_DEVICE, "upload of the allow bitmap failed");
        }
        *out = f.release();
        return FSGPU_OK;
    });
}

void fsgpu_allow_bitmap_destroy(fsgpu_allow_bitmap* f) {
    if (!f) return;
    if (f->device >= 0) (void)hipSetDevice(f->device);
    f->dev.release();
    delete f;
}

uint64_t fsgpu_allow_bitmap_allowed_rows(const fsgpu_allow_bitmap* f) { return f ? f->allowed : 0; }

static fsgpu_status check_filter(const fsgpu_index* idx, const fsgpu_allow_bitmap* f) {
    if (!idx || !f) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (f->device != idx->impl.device() || f->nrows != idx->impl.record_count())
        return fail(FSGPU_ERR_INVALID_CONFIG, "the allow bitmap was made for another index (device or record count differ)");
    return FSGPU_OK;
}

fsgpu_status fsgpu_search_topk_filtered(fsgpu_index* idx, const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                        const fsgpu_allow_bitmap* filter, uint32_t* out_rows, float* out_scores, uint32_t* out_counts) {
    const fsgpu_status c = check_filter(idx, filter);
    if (c != FSGPU_OK) return c;
    return search_topk_common(idx, queries, nq, query_len, k, filter->words.data(), static_cast<const uint64_t*>(filter->dev.ptr), out_rows,
                              out_scores, out_counts);
}

fsgpu_status fsgpu_search_topk_batched_filtered(fsgpu_index* idx, const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                                const fsgpu_allow_bitmap* filter, uint32_t* out_rows, float* out_scores,
                                                uint32_t* out_counts, uint32_t* out_fallbacks) {
    const fsgpu_status c = check_filter(idx, filter);
    if (c != FSGPU_OK) return c;
    if (nq && (!queries || !out_counts || (k && (!out_rows || !out_scores)))) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_top_k_batched(queries, nq, query_len, k, filter->words.data(), out_rows, out_scores, out_counts,
                                                     out_fallbacks, static_cast<const uint64_t*>(filter->dev.ptr)));
    });
}

fsgpu_status fsgpu_index_set_coalescing(fsgpu_index* idx, uint32_t max_batch, uint32_t max_wait_us) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    idx->coalescer.configure(max_batch, max_wait_us);
    return FSGPU_OK;
}

fsgpu_status fsgpu_index_coalescing_stats(fsgpu_index* idx, uint64_t* batches, uint64_t* requests) {
    if (!idx || !batches || !requests) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    idx->coalescer.stats(batches, requests);
    return FSGPU_OK;
}

fsgpu_status fsgpu_search_topk_device(fsgpu_index* idx, const float* queries_dev, uint32_t nq, uint32_t query_len,
                                      uint32_t k, const uint64_t* allow_bitmap_dev, uint32_t* out_rows_dev,
                                      float* out_scores_dev, uint32_t* out_counts_dev, void* hip_stream) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (nq && (!queries_dev || !out_counts_dev || (k && (!out_rows_dev || !out_scores_dev))))
        return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_top_k_device(queries_dev, nq, query_len, k, allow_bitmap_dev, out_rows_dev,
                                                    out_scores_dev, out_counts_dev,
                                                    static_cast<hipStream_t>(hip_stream)));
    });
}

fsgpu_status fsgpu_search_topk_batched(fsgpu_index* idx, const float* queries, uint32_t nq, uint32_t query_len,
                                       uint32_t k, const uint64_t* allow_bitmap, uint32_t* out_rows, float* out_scores,
                                       uint32_t* out_counts, uint32_t* out_fallbacks) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (nq && (!queries || !out_counts || (k && (!out_rows || !out_scores))))
        return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_top_k_batched(queries, nq, query_len, k, allow_bitmap, out_rows, out_scores,
                                                     out_counts, out_fallbacks));
    });
}

fsgpu_status fsgpu_search_topk_batched_device(fsgpu_index* idx, const float* queries_dev, uint32_t nq,
                                              uint32_t query_len, uint32_t k, const uint64_t* allow_bitmap_dev,
                                              uint32_t* out_rows_dev, float* out_scores_dev, uint32_t* out_counts_dev,
                                              void* hip_stream, uint32_t* out_fallbacks) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (nq && (!queries_dev || !out_counts_dev || (k && (!out_rows_dev || !out_scores_dev))))
        return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        if (k == 0 || idx->impl.record_count() == 0) {
            if (hipMemsetAsync(out_counts_dev, 0, (size_t)nq * 4, static_cast<hipStream_t>(hip_stream)) != hipSuccess)
                return fail(FSGPU_ERR_DEVICE, "hipMemsetAsync failed");
            return FSGPU_OK;
        }
        return finish(idx->impl.search_top_k_batched_device(queries_dev, nq, query_len, k, allow_bitmap_dev, out_rows_dev,
                                                            out_scores_dev, out_counts_dev,
                                                            static_cast<hipStream_t>(hip_stream), out_fallbacks));
    });
}

fsgpu_status fsgpu_search_topk_batched_packed_device(fsgpu_index* idx, const float* queries_dev, uint32_t nq,
                                                     uint32_t query_len, uint32_t k, const uint64_t* allow_bitmap_dev,
                                                     uint64_t* out_packed_dev, void* hip_stream, uint32_t* out_fallbacks) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (nq && k && (!queries_dev || !out_packed_dev)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (k > 256) return fail(FSGPU_ERR_INVALID_CONFIG, "packed shard search supports k <= 256");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        if (nq == 0 || k == 0) return FSGPU_OK;
        if (idx->impl.record_count() == 0) {
            if (hipMemsetAsync(out_packed_dev, 0xff, (size_t)nq * k * 8, static_cast<hipStream_t>(hip_stream)) != hipSuccess)
                return fail(FSGPU_ERR_DEVICE, "hipMemsetAsync failed");
            return FSGPU_OK;
        }
        return finish(idx->impl.search_top_k_batched_device(queries_dev, nq, query_len, k, allow_bitmap_dev, nullptr,
                                                            nullptr, nullptr, static_cast<hipStream_t>(hip_stream),
                                                            out_fallbacks, out_packed_dev));
    });
}

fsgpu_status fsgpu_search_topk_batched_device_begin(fsgpu_index* idx, const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                                    const uint64_t* allow_bitmap_dev, uint32_t* out_rows_dev, float* out_scores_dev,
                                                    uint32_t* out_counts_dev, uint64_t* out_packed_dev, void* hip_stream, int32_t* out_ticket) {
    if (!idx || !out_ticket) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (nq == 0 || k == 0 || !queries_dev) return fail(FSGPU_ERR_INVALID_CONFIG, "a begun search needs queries and k >= 1");
    if (out_packed_dev && k > 256) return fail(FSGPU_ERR_INVALID_CONFIG, "packed shard search supports k <= 256");
    if (!out_packed_dev && !(out_rows_dev && out_scores_dev && out_counts_dev)) return fail(FSGPU_ERR_NULL_ARGUMENT, "no output");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        if (idx->impl.record_count() == 0) return fail(FSGPU_ERR_INVALID_CONFIG, "the index is empty: use the blocking form");
        return finish(idx->impl.search_top_k_batched_device_begin(queries_dev, nq, query_len, k, allow_bitmap_dev, out_rows_dev, out_scores_dev,
                                                                  out_counts_dev, static_cast<hipStream_t>(hip_stream), out_packed_dev, out_ticket));
    });
}

fsgpu_status fsgpu_search_topk_batched_device_end(fsgpu_index* idx, int32_t ticket, uint32_t* out_fallbacks) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_top_k_batched_device_end(ticket, out_fallbacks));
    });
}

fsgpu_status fsgpu_search_topk_batched_device_end_late(fsgpu_index* idx, int32_t ticket, uint32_t* out_fallbacks, uint32_t* out_late_answers) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_top_k_batched_device_end(ticket, out_fallbacks, out_late_answers));
    });
}

fsgpu_status fsgpu_search_topk_packed_device(fsgpu_index* idx, const float* queries_dev, uint32_t nq,
                                             uint32_t query_len, uint32_t k, const uint64_t* allow_bitmap_dev,
                                             uint64_t* out_packed_dev, void* hip_stream) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (nq && k && (!queries_dev || !out_packed_dev)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_top_k_packed_device(queries_dev, nq, query_len, k, allow_bitmap_dev,
                                                           out_packed_dev, static_cast<hipStream_t>(hip_stream)));
    });
}

fsgpu_status fsgpu_merge_topk_device(int32_t device, const uint64_t* lists_dev, uint32_t nq, uint32_t nlists,
                                     uint32_t list_len, uint64_t q_stride, uint64_t l_stride, uint32_t k,
                                     uint32_t* out_rows_dev, float* out_scores_dev, uint32_t* out_counts_dev,
                                     void* hip_stream) {
    if (nq && (!lists_dev || !out_rows_dev || !out_scores_dev || !out_counts_dev))
        return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (k == 0 || k > 1024) return fail(FSGPU_ERR_INVALID_CONFIG, "merge supports 1 <= k <= 1024");
    return guarded([&]() -> fsgpu_status {
        return finish(fsgpu::merge_packed_lists_device(device, lists_dev, nq, nlists, list_len, q_stride, l_stride, k,
                                                       out_rows_dev, out_scores_dev, out_counts_dev,
                                                       static_cast<hipStream_t>(hip_stream)));
    });
}

// search_top_k_classified (crates/frankensearch-index/src/search.rs:227-261)
// ---- row-sharded index: one handle, one call per search (sharded_index.cpp) ----
fsgpu_status fsgpu_sharded_create_grouped(const int32_t* devices, uint32_t ndev, uint32_t query_groups, uint32_t dim, uint64_t nrows,
                                          const void* slab_f16_le, const uint64_t* live_bitmap, int32_t exchange, fsgpu_sharded** out) {
    if (!out) return fail(FSGPU_ERR_NULL_ARGUMENT, "out is null");
    *out = nullptr;
    return guarded([&]() -> fsgpu_status {
        auto* h = new fsgpu_sharded();
        fsgpu::SearchError e = h->impl.init_host(devices, ndev, dim, nrows, slab_f16_le, live_bitmap, exchange, query_groups);
        if (!e.ok()) {
            delete h;
            return finish(e);
        }
        *out = h;
        return FSGPU_OK;
    });
}

fsgpu_status fsgpu_sharded_create(const int32_t* devices, uint32_t ndev, uint32_t dim, uint64_t nrows, const void* slab_f16_le,
                                  const uint64_t* live_bitmap, int32_t exchange, fsgpu_sharded** out) {
    return fsgpu_sharded_create_grouped(devices, ndev, 1, dim, nrows, slab_f16_le, live_bitmap, exchange, out);
}

fsgpu_status fsgpu_sharded_create_device_grouped(const int32_t* devices, uint32_t ndev, uint32_t query_groups, uint32_t dim,
                                                 const uint64_t* shard_rows, const void* const* shard_slabs_dev,
                                                 const uint64_t* const* shard_live_dev, int32_t exchange, fsgpu_sharded** out) {
    if (!out) return fail(FSGPU_ERR_NULL_ARGUMENT, "out is null");
    *out = nullptr;
    return guarded([&]() -> fsgpu_status {
        auto* h = new fsgpu_sharded();
        fsgpu::SearchError e = h->impl.init_device(devices, ndev, dim, shard_rows, shard_slabs_dev, shard_live_dev, exchange, query_groups);
        if (!e.ok()) {
            delete h;
            return finish(e);
        }
        *out = h;
        return FSGPU_OK;
    });
}

fsgpu_status fsgpu_sharded_create_device(const int32_t* devices, uint32_t ndev, uint32_t dim, const uint64_t* shard_rows,
                                         const void* const* shard_slabs_dev, const uint64_t* const* shard_live_dev,
                                         int32_t exchange, fsgpu_sharded** out) {
    return fsgpu_sharded_create_device_grouped(devices, ndev, 1, dim, shard_rows, shard_slabs_dev, shard_live_dev, exchange, out);
}

uint32_t fsgpu_sharded_query_groups(const fsgpu_sharded* idx) { return idx ? idx->impl.query_groups() : 0; }
uint32_t fsgpu_sharded_row_shards(const fsgpu_sharded* idx) { return idx ? idx->impl.row_shards() : 0; }

fsgpu_status fsgpu_sharded_set_int8_latency(fsgpu_sharded* idx, int32_t enabled) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.set_int8_latency(enabled != 0));
    });
}

void fsgpu_sharded_destroy(fsgpu_sharded* idx) { delete idx; }
uint64_t fsgpu_sharded_record_count(const fsgpu_sharded* idx) { return idx ? idx->impl.record_count() : 0; }
uint32_t fsgpu_sharded_dimension(const fsgpu_sharded* idx) { return idx ? idx->impl.dimension() : 0; }
uint32_t fsgpu_sharded_shard_count(const fsgpu_sharded* idx) { return idx ? idx->impl.shard_count() : 0; }
int32_t fsgpu_sharded_exchange_mode(const fsgpu_sharded* idx) { return idx ? idx->impl.exchange_mode() : 0; }
int32_t fsgpu_sharded_device(const fsgpu_sharded* idx, uint32_t shard) { return idx ? idx->impl.shard_device(shard) : -1; }

fsgpu_status fsgpu_sharded_shard_range(const fsgpu_sharded* idx, uint32_t shard, uint64_t* row_lo, uint64_t* row_hi) {
    if (!idx || !row_lo || !row_hi) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (!idx->impl.shard_range(shard, row_lo, row_hi)) return fail(FSGPU_ERR_INVALID_CONFIG, "shard ordinal out of range");
    return FSGPU_OK;
}

fsgpu_status fsgpu_sharded_set_hreduce(fsgpu_sharded* idx, int32_t mode) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (mode < FSGPU_HREDUCE_SSE2 || mode > FSGPU_HREDUCE_SEQ) return fail(FSGPU_ERR_INVALID_CONFIG, "unknown hreduce mode");
    std::lock_guard<std::mutex> lock(idx->impl.mutex());
    idx->impl.set_hreduce(mode);
    return FSGPU_OK;
}

static fsgpu::ShardedIndex::Request sharded_request(const fsgpu_sharded_request* rq) {
    fsgpu::ShardedIndex::Request r;
    r.queries = rq->queries;
    r.queries_dev = rq->queries_dev;
    r.nq = rq->nq;
    r.k = rq->k;
    r.mode = static_cast<fsgpu::ShardedIndex::Mode>(rq->mode);
    r.multiplier = rq->candidate_multiplier;
    r.allow = rq->allow_bitmap;
    return r;
}

static fsgpu_status check_sharded_request(const fsgpu_sharded* idx, const fsgpu_sharded_request* rq) {
    if (!idx || !rq) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (rq->nq && !rq->queries && !rq->queries_dev) return fail(FSGPU_ERR_NULL_ARGUMENT, "queries is null");
    if (rq->mode < FSGPU_SHARDED_EXACT || rq->mode > FSGPU_SHARDED_4BIT_TWO_PASS) return fail(FSGPU_ERR_INVALID_CONFIG, "unknown search mode");
    return FSGPU_OK;
}

fsgpu_status fsgpu_sharded_search(fsgpu_sharded* idx, const fsgpu_sharded_request* request, uint32_t* out_rows, float* out_scores,
                                  uint32_t* out_counts, uint32_t* out_fallbacks) {
    const fsgpu_status c = check_sharded_request(idx, request);
    if (c != FSGPU_OK) return c;
    if (request->nq && (!out_counts || (request->k && (!out_rows || !out_scores)))) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    // concurrent single-query callers share one search of the shards (same hits: the batched mode is exact, the two-pass
    // candidates are per query)
    if (idx->coalescer.enabled() && request->nq == 1 && request->k >= 1 && request->k <= 64 && !request->allow_bitmap && request->queries &&
        request->query_len == idx->impl.dimension() && idx->impl.record_count() > 0 &&
        (request->mode == FSGPU_SHARDED_EXACT || request->mode == FSGPU_SHARDED_INT8_TWO_PASS)) {
        if (out_fallbacks) *out_fallbacks = 0;
        const uint32_t mult = request->mode == FSGPU_SHARDED_INT8_TWO_PASS ? (request->candidate_multiplier ? request->candidate_multiplier : 1) : 0;
        return coalesced_sharded_search(idx, request->queries, request->query_len, request->k, mult, out_rows, out_scores, out_counts);
    }
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search(sharded_request(request), request->query_len, out_rows, out_scores, out_counts, out_fallbacks));
    });
}

fsgpu_status fsgpu_sharded_set_coalescing(fsgpu_sharded* idx, uint32_t max_batch, uint32_t max_wait_us) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    idx->coalescer.configure(max_batch, max_wait_us);
    return FSGPU_OK;
}

fsgpu_status fsgpu_sharded_coalescing_stats(fsgpu_sharded* idx, uint64_t* batches, uint64_t* requests) {
    if (!idx || !batches || !requests) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    idx->coalescer.stats(batches, requests);
    return FSGPU_OK;
}

// queries resident in parts on several devices (data-parallel encoders): request->queries / queries_dev are ignored
fsgpu_status fsgpu_sharded_search_parts(fsgpu_sharded* idx, const fsgpu_sharded_request* request, const float* const* parts_dev,
                                        const uint32_t* part_counts, const int32_t* part_devices, uint32_t n_parts, uint32_t* out_rows,
                                        float* out_scores, uint32_t* out_counts, uint32_t* out_fallbacks) {
    if (!idx || !request) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (request->mode < FSGPU_SHARDED_EXACT || request->mode > FSGPU_SHARDED_4BIT_TWO_PASS) return fail(FSGPU_ERR_INVALID_CONFIG, "unknown search mode");
    if (request->nq && (!n_parts || !parts_dev || !part_counts || !part_devices)) return fail(FSGPU_ERR_NULL_ARGUMENT, "query parts are required");
    if (request->nq && (!out_counts || (request->k && (!out_rows || !out_scores)))) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        fsgpu::ShardedIndex::Request r = sharded_request(request);
        r.queries = nullptr;
        r.queries_dev = nullptr;
        r.parts_dev = parts_dev;
        r.part_counts = part_counts;
        r.part_devices = part_devices;
        r.n_parts = n_parts;
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search(r, request->query_len, out_rows, out_scores, out_counts, out_fallbacks));
    });
}

fsgpu_status fsgpu_sharded_search_begin(fsgpu_sharded* idx, const fsgpu_sharded_request* request, uint64_t* out_ticket) {
    const fsgpu_status c = check_sharded_request(idx, request);
    if (c != FSGPU_OK) return c;
    if (!out_ticket) return fail(FSGPU_ERR_NULL_ARGUMENT, "ticket is null");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.begin(sharded_request(request), request->query_len, out_ticket));
    });
}

fsgpu_status fsgpu_sharded_search_end(fsgpu_sharded* idx, uint64_t ticket, uint32_t* out_rows, float* out_scores, uint32_t* out_counts,
                                      uint32_t* out_fallbacks) {
    if (!idx || !out_counts) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.end(ticket, out_rows, out_scores, out_counts, out_fallbacks));
    });
}

static fsgpu_status sharded_search(fsgpu_sharded* idx, const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                   bool batched, uint32_t* out_rows, float* out_scores, uint32_t* out_counts,
                                   uint32_t* out_fallbacks) {
    fsgpu_sharded_request rq{queries, nq, query_len, k, batched ? FSGPU_SHARDED_BATCHED : FSGPU_SHARDED_EXACT, 0, nullptr, nullptr};
    return fsgpu_sharded_search(idx, &rq, out_rows, out_scores, out_counts, out_fallbacks);
}

fsgpu_status fsgpu_sharded_search_topk(fsgpu_sharded* idx, const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                       uint32_t* out_rows, float* out_scores, uint32_t* out_counts) {
    return sharded_search(idx, queries, nq, query_len, k, false, out_rows, out_scores, out_counts, nullptr);
}

fsgpu_status fsgpu_sharded_search_topk_batched(fsgpu_sharded* idx, const float* queries, uint32_t nq, uint32_t query_len,
                                               uint32_t k, uint32_t* out_rows, float* out_scores, uint32_t* out_counts,
                                               uint32_t* out_fallbacks) {
    return sharded_search(idx, queries, nq, query_len, k, true, out_rows, out_scores, out_counts, out_fallbacks);
}

float fsgpu_sharded_quant_scale_max(const fsgpu_sharded* idx) { return idx ? idx->impl.quant_scale_max() : 0.0f; }

fsgpu_status fsgpu_sharded_open_fsvi(const char* path, const int32_t* devices, uint32_t ndev, int32_t exchange, fsgpu_sharded** out) {
    return fsgpu_sharded_open_fsvi_grouped(path, devices, ndev, 1, exchange, out);
}

fsgpu_status fsgpu_sharded_open_fsvi_grouped(const char* path, const int32_t* devices, uint32_t ndev, uint32_t query_groups, int32_t exchange,
                                             fsgpu_sharded** out) {
    if (!out || !path || !devices) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(FSGPU_ERR_NO_DEVICE, "no HIP device visible (libfsgpu has no CPU fallback)");
    return guarded([&]() -> fsgpu_status {
        auto* h = new fsgpu_sharded();
        fsgpu::SearchError e = h->impl.open_fsvi(path, devices, ndev, exchange, query_groups);
        if (!e.ok()) {
            delete h;
            return finish(e);
        }
        *out = h;
        return FSGPU_OK;
    });
}

fsgpu_status fsgpu_sharded_set_live_bitmap(fsgpu_sharded* idx, const uint64_t* live_bitmap) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.set_live_bitmap(live_bitmap));
    });
}

fsgpu_status fsgpu_sharded_soft_delete(fsgpu_sharded* idx, const char* doc_id, uint32_t doc_id_len, int32_t* out_deleted) {
    if (!idx || !out_deleted || (doc_id_len && !doc_id)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.soft_delete(doc_id, doc_id_len, out_deleted));
    });
}

fsgpu_status fsgpu_sharded_wal_append(fsgpu_sharded* idx, const char* doc_id, uint32_t doc_id_len, const float* vector,
                                      uint32_t vector_len) {
    if (!idx || !vector || (doc_id_len && !doc_id)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.wal_append(doc_id, doc_id_len, vector, vector_len));
    });
}

uint64_t fsgpu_sharded_wal_record_count(const fsgpu_sharded* idx) { return idx ? idx->impl.wal_record_count() : 0; }

fsgpu_status fsgpu_sharded_doc_id(const fsgpu_sharded* idx, uint32_t row, const char** out_ptr, uint32_t* out_len) {
    if (!idx || !out_ptr || !out_len) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return finish(idx->impl.doc_id_at(row, out_ptr, out_len));
}

fsgpu_status fsgpu_sharded_search_hits(fsgpu_sharded* idx, const float* query, uint32_t query_len, uint32_t k, uint32_t* out_rows,
                                       float* out_scores, uint32_t* out_count) {
    if (!idx || !query || !out_count || (k && (!out_rows || !out_scores))) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_hits(query, query_len, k, out_rows, out_scores, out_count));
    });
}

fsgpu_status fsgpu_sharded_gather_dot(fsgpu_sharded* idx, const float* query, uint32_t query_len, const uint32_t* rows, uint32_t n,
                                      float* out_scores) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (n && (!query || !rows || !out_scores)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.gather_dot(query, query_len, rows, n, out_scores));
    });
}

fsgpu_status fsgpu_search_topk_classified(fsgpu_index* idx, const float* query, uint32_t query_len, uint32_t k,
                                          uint32_t* out_rows, float* out_scores, uint32_t* out_count,
                                          int32_t* zero_signal) {
    if (!idx || !query || !out_count || !zero_signal) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    *zero_signal = FSGPU_ZERO_SIGNAL_NONE;
    *out_count = 0;
    if (query_len != idx->impl.dimension()) {
        g_last_error = "expected " + std::to_string(idx->impl.dimension()) + ", found " + std::to_string(query_len);
        return FSGPU_ERR_DIMENSION_MISMATCH;
    }
    if (k == 0) {
        *zero_signal = FSGPU_ZERO_SIGNAL_CALLER_REQUESTED_ZERO_K;
        return FSGPU_OK;
    }
    bool all_zero = true;
    for (uint32_t i = 0; i < query_len; ++i) {
        if (!std::isfinite(query[i])) return fail(FSGPU_ERR_INVALID_CONFIG, "query vector must be finite");
        if (query[i] != 0.0f) all_zero = false;
    }
    if (all_zero) {
        *zero_signal = FSGPU_ZERO_SIGNAL_ZERO_NORM_QUERY;
        return FSGPU_OK;
    }
    const bool table = idx->impl.has_doc_ids();
    fsgpu_status st = table ? fsgpu_search_hits(idx, query, query_len, k, out_rows, out_scores, out_count)
                            : fsgpu_search_topk(idx, query, 1, query_len, k, nullptr, out_rows, out_scores, out_count);
    if (st == FSGPU_OK && *out_count == 0) {
        // ZeroSignalState::empty_result_reason without a filter (config.rs:696-740).  The exact scan returns every live row
        // (a NaN score still ranks, search.rs:1655-1661), so an empty result means no live main record: the census the
        // reference computes lazily reduces to the record and WAL counts.
        // (the counts are read under the state lock a concurrent fsgpu_index_wal_append / soft_delete takes exclusively)
        uint64_t records, wal;
        {
            std::shared_lock<std::shared_mutex> state(idx->state_mu);
            records = idx->impl.record_count();
            wal = idx->impl.wal_record_count();
        }
        *zero_signal = records == 0 && wal == 0 ? FSGPU_ZERO_SIGNAL_NEWLY_CREATED_EMPTY
                       : wal == 0               ? FSGPU_ZERO_SIGNAL_ALL_TOMBSTONED
                                                : FSGPU_ZERO_SIGNAL_WAL_ONLY_NO_LIVE_RECORDS;
    }
    return st;
}

// search_top_k -> scan_wal -> resolve_sorted_entries (search.rs:426-494, 1449-1475, 1503-1558).
fsgpu_status fsgpu_search_hits(fsgpu_index* idx, const float* query, uint32_t query_len, uint32_t k,
                               uint32_t* out_rows, float* out_scores, uint32_t* out_count) {
    if (!idx || !query || !out_count) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    *out_count = 0;
    if (k && (!out_rows || !out_scores)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_hits(query, query_len, k, out_rows, out_scores, out_count));
    });
}

fsgpu_status fsgpu_search_topk_int8_two_pass_batched(fsgpu_index* idx, const float* queries, uint32_t nq, uint32_t query_len,
                                                     uint32_t k, uint32_t candidate_multiplier, uint32_t* out_rows,
                                                     float* out_scores, uint32_t* out_counts, uint32_t* out_fallbacks) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (nq && (!queries || !out_counts || (k && (!out_rows || !out_scores))))
        return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_top_k_int8_batched(queries, nq, query_len, k, candidate_multiplier, out_rows,
                                                          out_scores, out_counts, out_fallbacks));
    });
}

fsgpu_status fsgpu_search_topk_4bit_two_pass_batched(fsgpu_index* idx, const float* queries, uint32_t nq, uint32_t query_len,
                                                     uint32_t k, uint32_t candidate_multiplier, uint32_t* out_rows,
                                                     float* out_scores, uint32_t* out_counts, uint32_t* out_fallbacks) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (nq && (!queries || !out_counts || (k && (!out_rows || !out_scores))))
        return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_top_k_int8_batched(queries, nq, query_len, k, candidate_multiplier, out_rows,
                                                          out_scores, out_counts, out_fallbacks, 4));
    });
}

// VectorIndex::search_top_k_4bit_two_pass (search.rs:876-946)
fsgpu_status fsgpu_search_topk_4bit_two_pass(fsgpu_index* idx, const float* query, uint32_t query_len, uint32_t k,
                                             uint32_t candidate_multiplier, uint32_t* out_rows, float* out_scores,
                                             uint32_t* out_count) {
    if (!idx || !query || !out_count) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    *out_count = 0;
    if (k && (!out_rows || !out_scores)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_top_k_4bit_two_pass(query, query_len, k, candidate_multiplier, out_rows,
                                                           out_scores, out_count));
    });
}

// VectorIndex::mrl_search_with_stats (mrl.rs:241-395)
fsgpu_status fsgpu_search_mrl(fsgpu_index* idx, const float* query, uint32_t query_len, uint32_t k, uint32_t search_dims,
                              uint32_t rescore_dims, uint32_t rescore_top_k, uint32_t* out_rows, float* out_scores,
                              uint32_t* out_count, fsgpu_mrl_stats* stats) {
    if (!idx || !query || !out_count) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    *out_count = 0;
    if (k && (!out_rows || !out_scores)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        fsgpu::VectorIndex::MrlStats st;
        const fsgpu_status rc = finish(idx->impl.mrl_search(query, query_len, k, search_dims, rescore_dims, rescore_top_k,
                                                            out_rows, out_scores, out_count, &st));
        if (stats) {
            stats->scan_dims = st.scan_dims;
            stats->rescore_dims = st.rescore_dims;
            stats->candidates_rescored = st.candidates_rescored;
            stats->records_scanned = st.records_scanned;
            stats->fell_back_to_full = st.fell_back_to_full ? 1 : 0;
        }
        return rc;
    });
}

fsgpu_status fsgpu_search_mrl_batched(fsgpu_index* idx, const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                      uint32_t search_dims, uint32_t rescore_dims, uint32_t rescore_top_k, uint32_t* out_rows,
                                      float* out_scores, uint32_t* out_counts, uint32_t* out_fallbacks) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (nq && (!queries || !out_counts || (k && (!out_rows || !out_scores)))) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.mrl_search_batched(queries, nq, query_len, k, search_dims, rescore_dims, rescore_top_k, out_rows,
                                                   out_scores, out_counts, out_fallbacks));
    });
}

// VectorIndex::search_top_k_int8_two_pass (search.rs:514-661)
fsgpu_status fsgpu_search_topk_int8_two_pass(fsgpu_index* idx, const float* query, uint32_t query_len, uint32_t k,
                                             uint32_t candidate_multiplier, uint32_t* out_rows, float* out_scores,
                                             uint32_t* out_count) {
    if (!idx || !query || !out_count) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    *out_count = 0;
    if (k && (!out_rows || !out_scores)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (idx->coalescer.enabled() && k >= 1 && k <= 64 && query_len == idx->impl.dimension() && !idx->impl.has_doc_ids() &&
        idx->impl.wal_record_count() == 0)
        return coalesced_search(idx, query, k, candidate_multiplier ? candidate_multiplier : 1, out_rows, out_scores, out_count);
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_top_k_int8_two_pass(query, query_len, k, candidate_multiplier, out_rows,
                                                           out_scores, out_count));
    });
}

// VectorIndex::append (lib.rs:2532-2720)
fsgpu_status fsgpu_index_wal_append(fsgpu_index* idx, const char* doc_id, uint32_t doc_id_len, const float* vector,
                                    uint32_t vector_len) {
    if (!idx || !doc_id || !vector) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::unique_lock<std::shared_mutex> state(idx->state_mu);
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        const fsgpu::SearchError e = idx->impl.wal_append(doc_id, doc_id_len, vector, vector_len);
        idx->impl.sync_replicas();   // the shadowed main row was tombstoned
        return finish(e);
    });
}

uint64_t fsgpu_index_wal_record_count(const fsgpu_index* idx) { return idx ? idx->impl.wal_record_count() : 0; }

fsgpu_status fsgpu_gather_dot(fsgpu_index* idx, const float* query, uint32_t query_len, const uint32_t* rows,
                              uint32_t n, float* out_scores) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (n && (!query || !rows || !out_scores)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.gather_dot(query, query_len, rows, n, out_scores));
    });
}

fsgpu_status fsgpu_alignment_create(fsgpu_index* fast, fsgpu_index* quality, fsgpu_alignment** out) {
    if (!fast || !quality || !out) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    *out = nullptr;
    return guarded([&]() -> fsgpu_status {
        auto a = std::make_unique<fsgpu_alignment>();
        std::shared_lock<std::shared_mutex> lf(fast->state_mu);
        std::shared_lock<std::shared_mutex> lq(quality->state_mu, std::defer_lock);
        if (quality != fast) lq.lock();
        const fsgpu_status st = finish(a->impl.build(fast->impl, quality->impl));
        if (st == FSGPU_OK) *out = a.release();
        return st;
    });
}

void fsgpu_alignment_destroy(fsgpu_alignment* a) { delete a; }
int32_t fsgpu_alignment_kind(const fsgpu_alignment* a) { return a ? (int32_t)a->impl.kind() : FSGPU_ALIGNMENT_NONE; }
int64_t fsgpu_alignment_quality_row(const fsgpu_alignment* a, uint64_t fast_row) { return a ? a->impl.quality_row(fast_row) : -1; }
uint64_t fsgpu_alignment_unmatched_quality_docs(const fsgpu_alignment* a) { return a ? a->impl.unmatched_quality_docs() : 0; }

fsgpu_status fsgpu_quality_scores_for_hits(fsgpu_index* fast, fsgpu_index* quality, const fsgpu_alignment* alignment,
                                           const float* query, uint32_t query_len, const fsgpu_scored_doc* hits, uint32_t n,
                                           float* out_scores, uint8_t* out_present) {
    if (!fast || !quality || !alignment) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (n && (!query || !hits || !out_scores || !out_present)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::vector<fsgpu::HitRef> refs(n);
        for (uint32_t i = 0; i < n; ++i) refs[i] = fsgpu::HitRef{hits[i].doc_id, hits[i].doc_id_len, hits[i].index};
        std::shared_lock<std::shared_mutex> lf(fast->state_mu);
        std::shared_lock<std::shared_mutex> lq(quality->state_mu, std::defer_lock);
        if (quality != fast) lq.lock();
        return finish(fsgpu::quality_scores_for_hits(fast->impl, quality->impl, alignment->impl, query, query_len, refs.data(), n,
                                                     out_scores, out_present));
    });
}

fsgpu_status fsgpu_quality_scores_for_hits_batched(fsgpu_index* fast, fsgpu_index* quality, const fsgpu_alignment* alignment,
                                                   const float* queries, uint32_t nq, uint32_t query_len, const fsgpu_scored_doc* hits,
                                                   const uint32_t* hit_offsets, float* out_scores, uint8_t* out_present) {
    if (!fast || !quality || !alignment) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (nq && (!queries || !hit_offsets)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    const uint32_t n = nq ? hit_offsets[nq] : 0;
    if (n && (!hits || !out_scores || !out_present)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::vector<fsgpu::HitRef> refs(n);
        for (uint32_t i = 0; i < n; ++i) refs[i] = fsgpu::HitRef{hits[i].doc_id, hits[i].doc_id_len, hits[i].index};
        std::shared_lock<std::shared_mutex> lf(fast->state_mu);
        std::shared_lock<std::shared_mutex> lq(quality->state_mu, std::defer_lock);
        if (quality != fast) lq.lock();
        return finish(fsgpu::quality_scores_for_hits_batched(fast->impl, quality->impl, alignment->impl, queries, nq, query_len, refs.data(),
                                                             hit_offsets, out_scores, out_present));
    });
}

// The same pairing over two row-sharded handles: the walk runs over their catalogs (fsgpu_sharded_open_fsvi) — raw shards pair by
// row —, the re-scoring gathers dot_query_at on the shards that own the quality rows (fsgpu_sharded_gather_dot).
fsgpu_status fsgpu_sharded_alignment_create(fsgpu_sharded* fast, fsgpu_sharded* quality, fsgpu_alignment** out) {
    if (!fast || !quality || !out) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    *out = nullptr;
    return guarded([&]() -> fsgpu_status {
        auto a = std::make_unique<fsgpu_alignment>();
        std::unique_lock<std::mutex> lf(fast->impl.mutex());
        std::unique_lock<std::mutex> lq(quality->impl.mutex(), std::defer_lock);
        if (quality != fast) lq.lock();
        const fsgpu_status st = finish(a->impl.build(fast->impl.catalog(), fast->impl.record_count(), quality->impl.catalog(),
                                                     quality->impl.record_count()));
        if (st == FSGPU_OK) *out = a.release();
        return st;
    });
}

fsgpu_status fsgpu_sharded_quality_scores_for_hits(fsgpu_sharded* fast, fsgpu_sharded* quality, const fsgpu_alignment* alignment,
                                                   const float* query, uint32_t query_len, const fsgpu_scored_doc* hits, uint32_t n,
                                                   float* out_scores, uint8_t* out_present) {
    if (!fast || !quality || !alignment) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (n && (!query || !hits || !out_scores || !out_present)) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::vector<fsgpu::HitRef> refs(n);
        for (uint32_t i = 0; i < n; ++i) refs[i] = fsgpu::HitRef{hits[i].doc_id, hits[i].doc_id_len, hits[i].index};
        // The fast side is read through its record table AND its tombstones (find_index_by_doc_id), which fsgpu_sharded_soft_delete /
        // _wal_append change under fast's mutex; the quality handle is held for the gather.  Fast first, then quality: the order of
        // fsgpu_sharded_alignment_create.
        std::unique_lock<std::mutex> lf(fast->impl.mutex());
        std::unique_lock<std::mutex> lq(quality->impl.mutex(), std::defer_lock);
        if (quality != fast) lq.lock();
        fsgpu::QualityTierView view;
        view.table = quality->impl.catalog();
        view.rows = quality->impl.record_count();
        view.dim = quality->impl.dimension();
        view.gather = [quality](const float* q, uint32_t len, const uint32_t* rows, uint32_t cnt, float* out) {
            return quality->impl.gather_dot(q, len, rows, cnt, out);
        };
        return finish(fsgpu::quality_scores_for_hits(fast->impl.catalog(), fast->impl.record_count(), view, alignment->impl, query,
                                                     query_len, refs.data(), n, out_scores, out_present));
    });
}

static fsgpu_status convert_on_device(int32_t device, const void* src, size_t src_elem, uint64_t n, void* dst,
                                      size_t dst_elem, bool encode) {
    if (n == 0) return FSGPU_OK;
    if (!src || !dst) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(FSGPU_ERR_NO_DEVICE, "no HIP device visible (libfsgpu has no CPU fallback)");
    if (device < 0 || device >= count) return fail(FSGPU_ERR_INVALID_CONFIG, "device ordinal out of range");
    return guarded([&]() -> fsgpu_status {
        if (hipSetDevice(device) != hipSuccess) return fail(FSGPU_ERR_DEVICE, "hipSetDevice failed");
        fsgpu::DeviceBuffer a, b;
        fsgpu::SearchError e = a.reserve((size_t)n * src_elem);
        if (e.ok()) e = b.reserve((size_t)n * dst_elem);
        if (!e.ok()) {
            a.release();
            b.release();
            return finish(e);
        }
        hipError_t he = hipMemcpy(a.ptr, src, (size_t)n * src_elem, hipMemcpyHostToDevice);
        if (he == hipSuccess)
            he = encode ? fsgpu::launch_encode_f16(static_cast<const float*>(a.ptr), (size_t)n,
                                                   static_cast<unsigned short*>(b.ptr), nullptr)
                        : fsgpu::launch_widen_f16(static_cast<const unsigned short*>(a.ptr), (size_t)n,
                                                  static_cast<float*>(b.ptr), nullptr);
        if (he == hipSuccess) he = hipMemcpy(dst, b.ptr, (size_t)n * dst_elem, hipMemcpyDeviceToHost);
        a.release();
        b.release();
        if (he != hipSuccess) {
            g_last_error = hipGetErrorString(he);
            return FSGPU_ERR_DEVICE;
        }
        return FSGPU_OK;
    });
}

fsgpu_status fsgpu_fsvi_write(const char* path, const char* embedder_id, const char* embedder_revision, uint32_t dim,
                              uint64_t n, const char* const* doc_ids, const uint32_t* doc_id_lens, const float* vectors,
                              uint8_t compaction_gen, int32_t device) {
    if (n >= 0x7fffffffull) return fail(FSGPU_ERR_INVALID_CONFIG, "record count must fit the launch grid");
    return guarded([&]() -> fsgpu_status {
        return finish(fsgpu::write_fsvi_v1(path, embedder_id, embedder_revision, dim, n, doc_ids, doc_id_lens, vectors,
                                           compaction_gen, device));
    });
}

fsgpu_status fsgpu_fsvi_write_quant(const char* path, const char* embedder_id, const char* embedder_revision, uint32_t dim,
                                    uint64_t n, const char* const* doc_ids, const uint32_t* doc_id_lens,
                                    const float* vectors, uint8_t compaction_gen, int32_t device, uint8_t quantization) {
    if (n >= 0x7fffffffull) return fail(FSGPU_ERR_INVALID_CONFIG, "record count must fit the launch grid");
    return guarded([&]() -> fsgpu_status {
        return finish(fsgpu::write_fsvi_v1(path, embedder_id, embedder_revision, dim, n, doc_ids, doc_id_lens, vectors,
                                           compaction_gen, device, quantization));
    });
}

fsgpu_status fsgpu_bench_fixture_device(int32_t device, uint64_t first, uint64_t n, uint32_t dim, uint32_t clusters, float noise,
                                        uint64_t seed_base, int32_t as_f16, void* out_dev, void* hip_stream) {
    if (n && !out_dev) return fail(FSGPU_ERR_NULL_ARGUMENT, "out_dev is null");
    if (dim == 0 || clusters == 0) return fail(FSGPU_ERR_INVALID_CONFIG, "dim and clusters must be positive");
    return guarded([&]() -> fsgpu_status {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(FSGPU_ERR_NO_DEVICE, "no HIP device visible");
        if (device < 0 || device >= count) return fail(FSGPU_ERR_INVALID_CONFIG, "device ordinal out of range");
        if (hipSetDevice(device) != hipSuccess) return fail(FSGPU_ERR_DEVICE, "hipSetDevice failed");
        fsgpu::DeviceBuffer cent;
        fsgpu::SearchError e = cent.reserve((size_t)clusters * dim * 4);
        if (!e.ok()) return finish(e);
        hipStream_t st = static_cast<hipStream_t>(hip_stream);
        hipError_t he = fsgpu::launch_bench_fixture(first, n, dim, clusters, noise, seed_base, static_cast<float*>(cent.ptr),
                                                    as_f16 ? static_cast<unsigned short*>(out_dev) : nullptr,
                                                    as_f16 ? nullptr : static_cast<float*>(out_dev), st);
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        cent.release();
        if (he != hipSuccess) return fail(FSGPU_ERR_DEVICE, hipGetErrorString(he));
        return FSGPU_OK;
    });
}

fsgpu_status fsgpu_lab_sort_keys_desc(int32_t device, const uint64_t* keys, uint64_t n, uint64_t varying_bits, uint64_t* out_sorted) {
    if (n && (!keys || !out_sorted)) return fail(FSGPU_ERR_NULL_ARGUMENT, "keys / out_sorted is null");
    return guarded([&]() -> fsgpu_status {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(FSGPU_ERR_NO_DEVICE, "no HIP device visible");
        if (device < 0 || device >= count) return fail(FSGPU_ERR_INVALID_CONFIG, "device ordinal out of range");
        if (hipSetDevice(device) != hipSuccess) return fail(FSGPU_ERR_DEVICE, "hipSetDevice failed");
        if (n == 0) return FSGPU_OK;
        fsgpu::DeviceBuffer in, out, tmp;
        size_t tmp_bytes = 0;
        (void)fsgpu::sort_keys_desc_temp_bytes(n, &tmp_bytes);
        fsgpu::SearchError e = in.reserve(n * 8);
        if (e.ok()) e = out.reserve(n * 8);
        if (e.ok()) e = tmp.reserve(tmp_bytes);
        hipError_t he = hipSuccess;
        if (e.ok()) {
            he = hipMemcpy(in.ptr, keys, n * 8, hipMemcpyHostToDevice);
            if (he == hipSuccess)
                he = fsgpu::sort_keys_desc(tmp.ptr, tmp.bytes, static_cast<const fsgpu::u64*>(in.ptr), static_cast<fsgpu::u64*>(out.ptr), n, nullptr, varying_bits);
            if (he == hipSuccess) he = hipStreamSynchronize(nullptr);
            if (he == hipSuccess) he = hipMemcpy(out_sorted, out.ptr, n * 8, hipMemcpyDeviceToHost);
        }
        in.release();
        out.release();
        tmp.release();
        if (!e.ok()) return finish(e);
        if (he != hipSuccess) return fail(FSGPU_ERR_DEVICE, hipGetErrorString(he));
        return FSGPU_OK;
    });
}

fsgpu_status fsgpu_encode_f32_to_f16(int32_t device, const float* src, uint64_t n, uint16_t* dst) {
    return convert_on_device(device, src, 4, n, dst, 2, true);
}

fsgpu_status fsgpu_widen_f16_to_f32(int32_t device, const uint16_t* src, uint64_t n, float* dst) {
    return convert_on_device(device, src, 2, n, dst, 4, false);
}

fsgpu_status fsgpu_m2v_create(int32_t device, const float* table, uint32_t vocab, uint32_t dim, fsgpu_m2v** out) {
    if (!out) return fail(FSGPU_ERR_NULL_ARGUMENT, "out is null");
    *out = nullptr;
    return guarded([&]() -> fsgpu_status {
        auto* h = new fsgpu_m2v();
        fsgpu::SearchError e = h->impl.init(device, table, vocab, dim);
        if (!e.ok()) {
            delete h;
            return finish(e);
        }
        *out = h;
        return FSGPU_OK;
    });
}

void fsgpu_m2v_destroy(fsgpu_m2v* m) { delete m; }

fsgpu_status fsgpu_m2v_embed_device(fsgpu_m2v* m, const uint32_t* ids, const uint32_t* offsets, uint32_t n, float* out_dev) {
    if (!m) return fail(FSGPU_ERR_NULL_ARGUMENT, "embedder is null");
    if (n && !out_dev) return fail(FSGPU_ERR_NULL_ARGUMENT, "out_dev is null");
    return guarded([&]() -> fsgpu_status { return finish(m->impl.embed_batch(ids, offsets, n, nullptr, out_dev)); });
}
uint32_t fsgpu_m2v_dimension(const fsgpu_m2v* m) { return m ? m->impl.dimension() : 0; }

fsgpu_status fsgpu_m2v_embed(fsgpu_m2v* m, const uint32_t* ids, const uint32_t* offsets, uint32_t n, float* out) {
    if (!m) return fail(FSGPU_ERR_NULL_ARGUMENT, "embedder is null");
    if (m->coalescer.enabled() && n == 1 && offsets && out && (ids || offsets[1] == offsets[0])) {
        return guarded([&]() -> fsgpu_status {
            EmbedCall<uint32_t> call;
            call.ids = ids ? ids + offsets[0] : nullptr;
            call.len = offsets[1] - offsets[0];
            call.out = out;
            m->coalescer.submit(
                &call, [m](std::vector<EmbedCall<uint32_t>*>& batch) { run_embed_batch(m, m->impl.dimension(), batch); },
                [](const EmbedCall<uint32_t>&, const EmbedCall<uint32_t>&) { return true; });
            if (call.exec_threw) return fail(FSGPU_ERR_DEVICE, "coalesced batch failed before this request was served");
            if (call.status != FSGPU_OK) g_last_error = call.detail;
            return call.status;
        });
    }
    return guarded([&]() -> fsgpu_status { return finish(m->impl.embed_batch(ids, offsets, n, out)); });
}

fsgpu_status fsgpu_m2v_set_coalescing(fsgpu_m2v* m, uint32_t max_batch, uint32_t max_wait_us) {
    if (!m) return fail(FSGPU_ERR_NULL_ARGUMENT, "embedder is null");
    m->coalescer.configure(max_batch, max_wait_us);
    return FSGPU_OK;
}

fsgpu_status fsgpu_bert_create(int32_t device, const fsgpu_bert_config* config, const fsgpu_bert_weights* weights,
                               fsgpu_bert** out) {
    if (!out || !config || !weights) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    *out = nullptr;
    return guarded([&]() -> fsgpu_status {
        auto* h = new fsgpu_bert();
        fsgpu::SearchError e = h->impl.init(device, *config, *weights);
        if (!e.ok()) {
            delete h;
            return finish(e);
        }
        *out = h;
        return FSGPU_OK;
    });
}

fsgpu_status fsgpu_bert_create_safetensors(int32_t device, const void* blob, uint64_t blob_len, float ln_eps, fsgpu_bert** out) {
    if (!out || !blob) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    *out = nullptr;
    return guarded([&]() -> fsgpu_status {
        auto h = std::make_unique<fsgpu_bert>();
        // the blob first (a malformed model file is reported as such on any host), then the device
        fsgpu::SearchError e = h->impl.init_safetensors(-1, blob, blob_len, ln_eps);
        if (!e.ok()) return finish(e);
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            return fail(FSGPU_ERR_NO_DEVICE, "no HIP device visible (libfsgpu has no CPU fallback)");
        if (device < 0 || device >= count) return fail(FSGPU_ERR_INVALID_CONFIG, "device ordinal out of range");
        e = h->impl.init_safetensors(device, blob, blob_len, ln_eps);
        if (!e.ok()) return finish(e);
        *out = h.release();
        return FSGPU_OK;
    });
}

void fsgpu_bert_destroy(fsgpu_bert* m) { delete m; }

fsgpu_status fsgpu_bert_embed_device(fsgpu_bert* m, const int32_t* ids, const uint32_t* offsets, uint32_t n, float* out_dev) {
    if (!m) return fail(FSGPU_ERR_NULL_ARGUMENT, "embedder is null");
    if (n && !out_dev) return fail(FSGPU_ERR_NULL_ARGUMENT, "out_dev is null");
    return guarded([&]() -> fsgpu_status { return finish(m->impl.embed_batch(ids, offsets, n, nullptr, out_dev)); });
}

fsgpu_status fsgpu_device_malloc(int32_t device, uint64_t bytes, void** out) {
    if (!out) return fail(FSGPU_ERR_NULL_ARGUMENT, "out is null");
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return fail(FSGPU_ERR_NO_DEVICE, "no such HIP device");
    if (hipMalloc(out, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        return fail(FSGPU_ERR_DEVICE, "device allocation failed");
    }
    return FSGPU_OK;
}

fsgpu_status fsgpu_device_free(int32_t device, void* ptr) {
    if (!ptr) return FSGPU_OK;
    if (hipSetDevice(device) != hipSuccess) return fail(FSGPU_ERR_NO_DEVICE, "no such HIP device");
    return hipFree(ptr) == hipSuccess ? FSGPU_OK : fail(FSGPU_ERR_DEVICE, "hipFree failed");
}

int32_t fsgpu_bert_device(const fsgpu_bert* m) { return m ? m->impl.device() : -1; }
int32_t fsgpu_m2v_device(const fsgpu_m2v* m) { return m ? m->impl.device() : -1; }
int32_t fsgpu_index_device(const fsgpu_index* idx) { return idx ? idx->impl.device() : -1; }

fsgpu_status fsgpu_search_topk_batched_device_queries(fsgpu_index* idx, const float* queries_dev, uint32_t nq, uint32_t query_len,
                                                      uint32_t k, uint32_t* out_rows, float* out_scores, uint32_t* out_counts,
                                                      uint32_t* out_fallbacks) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (nq && (!queries_dev || !out_counts || (k && (!out_rows || !out_scores)))) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_top_k_batched(queries_dev, nq, query_len, k, nullptr, out_rows, out_scores, out_counts, out_fallbacks,
                                                     nullptr, true));
    });
}
fsgpu_status fsgpu_search_topk_int8_two_pass_batched_device_queries(fsgpu_index* idx, const float* queries_dev, uint32_t nq, uint32_t query_len,
                                                                    uint32_t k, uint32_t candidate_multiplier, uint32_t* out_rows,
                                                                    float* out_scores, uint32_t* out_counts, uint32_t* out_fallbacks) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    if (nq && (!queries_dev || !out_counts || (k && (!out_rows || !out_scores)))) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.search_top_k_int8_batched(queries_dev, nq, query_len, k, candidate_multiplier, out_rows, out_scores, out_counts,
                                                          out_fallbacks, 8, true));
    });
}
uint32_t fsgpu_bert_dimension(const fsgpu_bert* m) { return m ? m->impl.dimension() : 0; }

fsgpu_status fsgpu_bert_embed(fsgpu_bert* m, const int32_t* ids, const uint32_t* offsets, uint32_t n, float* out) {
    if (!m) return fail(FSGPU_ERR_NULL_ARGUMENT, "embedder is null");
    if (m->coalescer.enabled() && n == 1 && offsets && out && (ids || offsets[1] == offsets[0])) {
        return guarded([&]() -> fsgpu_status {
            EmbedCall<int32_t> call;
            call.ids = ids ? ids + offsets[0] : nullptr;
            call.len = offsets[1] - offsets[0];
            call.out = out;
            m->coalescer.submit(
                &call, [m](std::vector<EmbedCall<int32_t>*>& batch) { run_embed_batch(m, m->impl.dimension(), batch); },
                [](const EmbedCall<int32_t>&, const EmbedCall<int32_t>&) { return true; });
            if (call.exec_threw) return fail(FSGPU_ERR_DEVICE, "coalesced batch failed before this request was served");
            if (call.status != FSGPU_OK) g_last_error = call.detail;
            return call.status;
        });
    }
    return guarded([&]() -> fsgpu_status { return finish(m->impl.embed_batch(ids, offsets, n, out)); });
}

fsgpu_status fsgpu_bert_set_coalescing(fsgpu_bert* m, uint32_t max_batch, uint32_t max_wait_us) {
    if (!m) return fail(FSGPU_ERR_NULL_ARGUMENT, "embedder is null");
    m->coalescer.configure(max_batch, max_wait_us);
    return FSGPU_OK;
}

fsgpu_status fsgpu_index_set_after_enqueue_hook(fsgpu_index* idx, fsgpu_after_enqueue_fn fn, void* ctx) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    std::lock_guard<std::mutex> lock(idx->impl.mutex());
    idx->impl.after_enqueue_fn = fn;
    idx->impl.after_enqueue_ctx = ctx;
    return FSGPU_OK;
}

fsgpu_status fsgpu_index_set_profiling(fsgpu_index* idx, int32_t enabled) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    std::lock_guard<std::mutex> lock(idx->impl.mutex());
    idx->impl.profiling = enabled != 0;
    idx->impl.profile_period = enabled > 1 ? enabled : 1;
    idx->impl.profile_tick_ = 0;
    return FSGPU_OK;
}

fsgpu_status fsgpu_index_scan_time(fsgpu_index* idx, double* total_ms, uint64_t* launches, int32_t reset) {
    if (!idx || !total_ms || !launches) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.scan_time(total_ms, launches, nullptr, reset != 0));
    });
}

fsgpu_status fsgpu_index_scan_stats(fsgpu_index* idx, double* total_ms, uint64_t* launches, uint64_t* rows, int32_t reset) {
    if (!idx || !total_ms || !launches || !rows) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    return guarded([&]() -> fsgpu_status {
        std::lock_guard<std::mutex> lock(idx->impl.mutex());
        return finish(idx->impl.scan_time(total_ms, launches, rows, reset != 0));
    });
}

fsgpu_status fsgpu_index_filter_stats(fsgpu_index* idx, uint64_t* gathered, uint64_t* scanned) {
    if (!idx || !gathered || !scanned) return fail(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    std::unique_lock<std::shared_mutex> state(idx->state_mu);   // no search is running on any lane
    std::lock_guard<std::mutex> lock(idx->impl.mutex());
    *gathered = idx->impl.filter_gathered;
    *scanned = idx->impl.filter_scanned;
    for (size_t i = 0; i < idx->impl.replica_count(); ++i) {
        *gathered += idx->impl.replica(i)->filter_gathered;
        *scanned += idx->impl.replica(i)->filter_scanned;
    }
    return FSGPU_OK;
}

fsgpu_status fsgpu_index_set_variant(fsgpu_index* idx, int32_t variant) {
    if (!idx) return fail(FSGPU_ERR_NULL_ARGUMENT, "index is null");
    std::unique_lock<std::shared_mutex> state(idx->state_mu);
    std::lock_guard<std::mutex> lock(idx->impl.mutex());
    idx->impl.variant = variant;
    idx->impl.sync_replicas();
    return FSGPU_OK;
}

}  // extern "C"
