// two_tier_many.cpp — SyncTwoTierSearcher's many-queries engine (include/fshost.h: fshost_two_tier_search_many,
// fshost_two_tier_set_batching).
//
// The reference's two-phase flow is per query (crates/frankensearch-fusion/src/sync_searcher.rs:616-943: fast embed -> fast-tier
// top-(k x multiplier) -> RRF = Initial; quality embed -> quality-tier top-(k x multiplier) -> blend -> RRF = Refined); its
// many-queries shapes are the batched scan bench (crates/frankensearch-index/benches/batched_query_scan.rs) and the embedder's batch
// coalescer (crates/frankensearch-embed/src/batch_coalescer.rs:18-23).  A GPU wants the whole flow in that shape: a thousand
// blocking per-query calls from a thousand host threads, each woken four times per query, keep the device a fifth busy.
//
// The engine runs CHUNKS of queries (up to 1,024: two 512-query passes of the matrix-core scan) through a pipeline of four stage
// threads over the C ABI — every stage a batched call the library already has —
//     FE  Model2Vec batch            fsgpu_m2v_embed_device            (vectors stay in HBM)
//     FS  fast tier, batched         fsgpu_search_topk_int8_two_pass_batched_device_queries / fsgpu_search_topk_batched_device_queries /
//                                    fsgpu_sharded_search(INT8_TWO_PASS | BATCHED, queries_dev)
//     QE  MiniLM batch               fsgpu_bert_embed_device
//     QS  quality tier, batched      fsgpu_search_topk_batched_device_queries / fsgpu_sharded_search(BATCHED, queries_dev)
// — the two tiers side by side (the quality tier needs nothing of phase 0 in the Retrieved pool), several chunks in flight (three sets
// of device buffers), and a pool of host threads that runs the PER-QUERY fusion of a chunk as soon as its inputs exist: the very
// functions SyncTwoTierSearcher::search runs (hits_from_rows, fuse_initial, fuse_final_retrieved / _rescored), so a query's results
// are those of the per-query call on the same tier answers — and the tier answers are the per-query searches' rows and score bits
// (the batched searches are bit-identical to the exact kernels; Model2Vec is bit-exact whatever the batch; the MiniLM embedding of a
// text is within the encoder's stated tolerance of its single-text embedding: the encoder picks kernels by batch shape).
//
// Two front ends feed it: search_many (arrays in, arrays out: the caller has a queue of requests) and the dynamic batching of
// concurrent fshost_two_tier_search callers (a collector thread turns whatever is queued into the next chunk while the previous
// ones run; ONE wake-up per query — when its Refined list is written — instead of one per stage).
// Host code only: calls nothing but include/fsgpu.h.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include "two_tier_searcher.hpp"

namespace fshost {

namespace {
using clk = std::chrono::steady_clock;
double ms_between(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
}  // namespace

// One query of a chunk: inputs, outputs and (dynamic batching) the caller parked on it.
struct ManyEngine::Query {
    const uint32_t* fast_ids = nullptr;
    uint32_t n_fast = 0;
    const int32_t* quality_ids = nullptr;
    uint32_t n_quality = 0;
    const fsgpu_scored_doc* lexical = nullptr;
    uint32_t n_lexical = 0;
    fshost_hit* initial_out = nullptr;
    uint32_t* n_initial = nullptr;
    fshost_hit* final_out = nullptr;
    uint32_t* n_final = nullptr;
    uint8_t* refinement_failed = nullptr;
    float* fast_vec_out = nullptr;
    float* quality_vec_out = nullptr;
    Waiter* waiter = nullptr;
};

// A blocked fshost_two_tier_search caller.
struct ManyEngine::Waiter {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    fsgpu_status status = FSGPU_OK;
    std::string detail;
    bool refinement_failed = false;
    std::string skip_reason;
    clk::time_point t_submit, t_initial, t_final;
    uint32_t k = 0;
    int parts = 0;   // fusion tasks of this query that have run (the Initial and the Refined one, in either order): the second wakes the caller
    Query q;
};

// What one search_many call waits for.
struct ManyEngine::Batch {
    std::mutex mu;
    std::condition_variable cv;
    uint32_t chunks_left = 0;
    fsgpu_status status = FSGPU_OK;
    std::string detail;
    std::string refinement_detail;
    double t_fe = 0, t_fs = 0, t_qe = 0, t_qs = 0, fusion_busy_ms = 0;
    double first_initial_ms = 0, first_refined_ms = 0;
    uint64_t refinement_failed = 0, fb_fast = 0, fb_quality = 0;
    bool fast_on_dev = false, quality_on_dev = false;
    clk::time_point t_start;
};

struct ManyEngine::Chunk {
    uint32_t n = 0, k = 0, fetch = 0, index = 0;
    std::vector<Query> q;
    std::vector<uint32_t> fast_ids, fast_offs, quality_offs;
    std::vector<int32_t> quality_ids;
    int slot = -1;                     // device buffer set (-1: none held)
    bool fast_on_dev = false, quality_on_dev = false;
    std::vector<float> fvec, qvec;     // host vectors (the paths that need them)
    std::vector<uint32_t> f_rows, f_counts, q_rows, q_counts;
    std::vector<float> f_scores, q_scores;
    std::atomic<int> final_parts{0};   // inputs of the chunk's FINAL fusion that exist (2: both)
    std::atomic<int> searches_left{0}; // tier searches still reading the slot's vectors
    std::atomic<uint32_t> tasks_left{0};
    std::atomic<int> status{FSGPU_OK}; // a failure of phase 0 / of an embedding: every query of the chunk reports it
    std::string detail;
    std::vector<float> qscores;        // re-scored pool: the fast pool's quality scores, [n, fetch] ...
    std::vector<uint8_t> qpresent;     // ... and whether a hit has one
    std::vector<uint8_t> q_refinement_failed;   // ... per query: its scores could not be produced
    bool quality_failed = false;       // the quality pool failed: RefinementFailed for the chunk's queries
    std::string quality_detail;
    Batch* batch = nullptr;
    std::atomic<uint64_t> n_refinement_failed{0};
    std::atomic<uint64_t> fusion_us{0};
};

struct ManyEngine::Task {
    Chunk* c;
    uint32_t q0, q1;
    bool final;
};

ManyEngine::ManyEngine(const SyncTwoTierSearcher& s, uint32_t fusion_threads) : s_(s) {
    const uint32_t hw = std::max(2u, std::thread::hardware_concurrency());
    n_workers_ = fusion_threads ? fusion_threads : std::max(2u, std::min(12u, hw > 6 ? hw - 4 : 2u));
    rescored_ = s_.cfg_.quality_pool == FSHOST_POOL_RESCORED;
    fdim_ = fsgpu_m2v_dimension(s_.m2v_);
    qdim_ = fsgpu_bert_dimension(s_.bert_);
    // which calls answer a tier for a chunk: the batched forms of what search() calls per query (tier_hits): row-level searches when
    // the ids are synthetic or the fast tier runs the int8 two-pass; with doc-id tables the exact searches go through search_hits
    // (WAL merge, shadowing, dedup) query by query, as search() does
    fast_rowlevel_ = s_.cfg_.fast_tier_int8_multiplier != 0 || s_.cfg_.doc_id_mode != 0;
    quality_rowlevel_ = s_.cfg_.doc_id_mode != 0;
    // embeddings stay in device memory when the embedder sits on the tier's (root) device
    const int32_t fast_dev = s_.fast_.index ? fsgpu_index_device(s_.fast_.index) : fsgpu_sharded_device(s_.fast_.sharded, 0);
    const int32_t quality_dev = s_.quality_.index ? fsgpu_index_device(s_.quality_.index) : fsgpu_sharded_device(s_.quality_.sharded, 0);
    fast_dev_ok_ = fast_rowlevel_ && fast_dev >= 0 && fsgpu_m2v_device(s_.m2v_) == fast_dev;
    quality_dev_ok_ = !rescored_ && quality_rowlevel_ && quality_dev >= 0 && fsgpu_bert_device(s_.bert_) == quality_dev;
    fast_dev_ = fast_dev;
    quality_dev_ = quality_dev;
    for (int i = 0; i < kSlots; ++i) free_slots_.push_back(i);
    threads_.emplace_back([this] { embed_stage(true); });
    threads_.emplace_back([this] { embed_stage(false); });
    threads_.emplace_back([this] { search_stage(true); });
    if (!rescored_) threads_.emplace_back([this] { search_stage(false); });
    else threads_.emplace_back([this] { rescore_stage(); });
    for (uint32_t i = 0; i < n_workers_; ++i) threads_.emplace_back([this] { fusion_worker(); });
}

ManyEngine::~ManyEngine() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    {
        std::lock_guard<std::mutex> lk(tmu_);
        pool_stop_ = true;
    }
    tcv_.notify_all();
    {
        std::lock_guard<std::mutex> lk(rmu_);
        server_stop_ = true;
    }
    rcv_.notify_all();
    if (collector_.joinable()) collector_.join();
    {
        std::lock_guard<std::mutex> lk(lmu_);
        lone_stop_ = true;
    }
    lcv_.notify_all();
    if (lone_thread_.joinable()) lone_thread_.join();
    for (std::thread& t : threads_) t.join();
    for (Slot& sl : slots_) {
        if (sl.f) (void)fsgpu_device_free(fast_dev_, sl.f);
        if (sl.q) (void)fsgpu_device_free(quality_dev_, sl.q);
    }
}

// A free set of device buffers for a chunk of up to `cap` queries (blocks while every set is in use: the pipeline's back-pressure).
int ManyEngine::acquire_slot(uint32_t cap) {
    int slot;
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !free_slots_.empty(); });
        if (stop_) return -1;
        slot = free_slots_.front();
        free_slots_.pop_front();
    }
    Slot& sl = slots_[slot];
    if (cap > sl.cap) {   // (grown on demand; a failed allocation leaves the chunk on the host path)
        if (sl.f) (void)fsgpu_device_free(fast_dev_, sl.f);
        if (sl.q) (void)fsgpu_device_free(quality_dev_, sl.q);
        sl.f = sl.q = nullptr;
        sl.cap = 0;
        bool ok = true;
        if (fast_dev_ok_) ok &= fsgpu_device_malloc(fast_dev_, (uint64_t)cap * fdim_ * 4, reinterpret_cast<void**>(&sl.f)) == FSGPU_OK;
        if (quality_dev_ok_) ok &= fsgpu_device_malloc(quality_dev_, (uint64_t)cap * qdim_ * 4, reinterpret_cast<void**>(&sl.q)) == FSGPU_OK;
        if (ok) sl.cap = cap;
    }
    return slot;
}

void ManyEngine::release_slot(int slot) {
    if (slot < 0) return;
    {
        std::lock_guard<std::mutex> lk(mu_);
        free_slots_.push_back(slot);
    }
    cv_.notify_all();
}

// Hands a prepared chunk to both embedding stages.
void ManyEngine::submit(Chunk* c) {
    const Slot& sl = slots_[c->slot];
    bool want_fvec = false, want_qvec = false;
    for (const Query& q : c->q) {
        want_fvec |= q.fast_vec_out != nullptr;
        want_qvec |= q.quality_vec_out != nullptr;
    }
    c->fast_on_dev = fast_dev_ok_ && sl.cap >= c->n && sl.f && !want_fvec;
    c->quality_on_dev = quality_dev_ok_ && sl.cap >= c->n && sl.q && !want_qvec;
    if (!c->fast_on_dev) c->fvec.resize((size_t)c->n * fdim_);
    if (!c->quality_on_dev) c->qvec.resize((size_t)c->n * qdim_);
    c->f_rows.resize((size_t)c->n * c->fetch);
    c->f_scores.resize((size_t)c->n * c->fetch);
    c->f_counts.resize(c->n);
    if (!rescored_) {
        c->q_rows.resize((size_t)c->n * c->fetch);
        c->q_scores.resize((size_t)c->n * c->fetch);
        c->q_counts.resize(c->n);
    }
    c->searches_left.store(rescored_ ? 1 : 2);
    if (c->batch) {
        c->batch->fast_on_dev = c->fast_on_dev;
        c->batch->quality_on_dev = c->quality_on_dev;
    }
    {
        std::lock_guard<std::mutex> lk(mu_);
        fe_q_.push_back(c);
        qe_q_.push_back(c);
    }
    cv_.notify_all();
}

ManyEngine::Chunk* ManyEngine::pop(std::deque<Chunk*>& q) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return stop_ || !q.empty(); });
    if (q.empty()) return nullptr;
    Chunk* c = q.front();
    q.pop_front();
    return c;
}

void ManyEngine::embed_stage(bool fast) {
    for (;;) {
        Chunk* c = pop(fast ? fe_q_ : qe_q_);
        if (!c) return;
        fsgpu_status st = FSGPU_OK;
        const auto b0 = clk::now();
        if (c->status.load() == FSGPU_OK) {
            if (fast)
                st = c->fast_on_dev ? fsgpu_m2v_embed_device(s_.m2v_, c->fast_ids.data(), c->fast_offs.data(), c->n, slots_[c->slot].f)
                                    : fsgpu_m2v_embed(s_.m2v_, c->fast_ids.data(), c->fast_offs.data(), c->n, c->fvec.data());
            else
                st = c->quality_on_dev ? fsgpu_bert_embed_device(s_.bert_, c->quality_ids.data(), c->quality_offs.data(), c->n, slots_[c->slot].q)
                                       : fsgpu_bert_embed(s_.bert_, c->quality_ids.data(), c->quality_offs.data(), c->n, c->qvec.data());
            if (st != FSGPU_OK) {   // an embedding's failure is the search's (embed_sync's error propagates); thread-local detail read here
                int expect = FSGPU_OK;
                const std::string d = fsgpu_last_error();
                if (c->status.compare_exchange_strong(expect, st)) c->detail = d;
            }
        }
        const double d = ms_between(b0, clk::now());
        if (c->batch) {
            std::lock_guard<std::mutex> lk(c->batch->mu);
            (fast ? c->batch->t_fe : c->batch->t_qe) += d;
        }
        if (st == FSGPU_OK) {   // the vectors the caller asked for
            const std::vector<float>& v = fast ? c->fvec : c->qvec;
            const uint32_t dim = fast ? fdim_ : qdim_;
            if (!v.empty())
                for (uint32_t i = 0; i < c->n; ++i) {
                    float* out = fast ? c->q[i].fast_vec_out : c->q[i].quality_vec_out;
                    if (out) std::memcpy(out, v.data() + (size_t)i * dim, (size_t)dim * 4);
                }
        }
        if (!fast && rescored_) {   // (the re-scored pool's second input is the quality VECTOR; no quality-tier search follows)
            final_part_ready(c);
            continue;
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            (fast ? fs_q_ : qs_q_).push_back(c);
        }
        cv_.notify_all();
    }
}

// One tier's answer for a chunk: rows / scores / counts [n, fetch] from device or host vectors.
fsgpu_status ManyEngine::tier_search(const Tier& tier, bool rowlevel, uint32_t int8_mult, const float* vec_dev, const float* vec_host, uint32_t n,
                                     uint32_t dim, uint32_t fetch, uint32_t* rows, float* scores, uint32_t* counts, uint32_t* fb, std::string* detail) {
    fsgpu_status st = FSGPU_OK;
    *fb = 0;
    if (!rowlevel) {   // doc-id tables + an exact search: search_hits per query (WAL merge, shadowing, dedup), as search() does
        for (uint32_t i = 0; i < n && st == FSGPU_OK; ++i)
            st = tier.search_hits(vec_host + (size_t)i * dim, dim, fetch, rows + (size_t)i * fetch, scores + (size_t)i * fetch, &counts[i]);
    } else if (tier.index) {
        if (int8_mult)
            st = vec_dev ? fsgpu_search_topk_int8_two_pass_batched_device_queries(tier.index, vec_dev, n, dim, fetch, int8_mult, rows, scores, counts, fb)
                         : fsgpu_search_topk_int8_two_pass_batched(tier.index, vec_host, n, dim, fetch, int8_mult, rows, scores, counts, fb);
        else
            st = vec_dev ? fsgpu_search_topk_batched_device_queries(tier.index, vec_dev, n, dim, fetch, rows, scores, counts, fb)
                         : fsgpu_search_topk_batched(tier.index, vec_host, n, dim, fetch, nullptr, rows, scores, counts, fb);
    } else {
        fsgpu_sharded_request rq{vec_dev ? nullptr : vec_host, n, dim, fetch, int8_mult ? FSGPU_SHARDED_INT8_TWO_PASS : FSGPU_SHARDED_BATCHED,
                                 int8_mult, nullptr, vec_dev};
        st = fsgpu_sharded_search(tier.sharded, &rq, rows, scores, counts, fb);
    }
    if (st != FSGPU_OK) *detail = fsgpu_last_error();
    return st;
}

void ManyEngine::search_stage(bool fast) {
    for (;;) {
        Chunk* c = pop(fast ? fs_q_ : qs_q_);
        if (!c) return;
        uint32_t fb = 0;
        std::string detail;
        fsgpu_status st = FSGPU_OK;
        const auto b0 = clk::now();
        if (c->status.load() == FSGPU_OK) {
            st = fast ? tier_search(s_.fast_, fast_rowlevel_, s_.cfg_.fast_tier_int8_multiplier, c->fast_on_dev ? slots_[c->slot].f : nullptr,
                                    c->fvec.data(), c->n, fdim_, c->fetch, c->f_rows.data(), c->f_scores.data(), c->f_counts.data(), &fb, &detail)
                      : tier_search(s_.quality_, quality_rowlevel_, 0, c->quality_on_dev ? slots_[c->slot].q : nullptr, c->qvec.data(), c->n, qdim_,
                                    c->fetch, c->q_rows.data(), c->q_scores.data(), c->q_counts.data(), &fb, &detail);
        }
        const auto b1 = clk::now();
        if (st != FSGPU_OK) {
            if (fast) {   // phase 0 failing fails the search (sync_searcher.rs:652-700)
                int expect = FSGPU_OK;
                if (c->status.compare_exchange_strong(expect, st)) c->detail = detail;
            } else {      // the quality pool failing is a RefinementFailed outcome of the chunk's queries (:820-839)
                c->quality_failed = true;
                c->quality_detail = detail;
            }
        }
        if (c->batch) {
            std::lock_guard<std::mutex> lk(c->batch->mu);
            (fast ? c->batch->t_fs : c->batch->t_qs) += ms_between(b0, b1);
            (fast ? c->batch->fb_fast : c->batch->fb_quality) += fb;
            if (c->index == 0) (fast ? c->batch->first_initial_ms : c->batch->first_refined_ms) = ms_between(c->batch->t_start, b1);
        }
        if (c->searches_left.fetch_sub(1) == 1) {   // both tiers have read the slot's vectors
            release_slot(c->slot);
            c->slot = -1;
        }
        if (fast) push_tasks(c, false);
        final_part_ready(c);
    }
}

void ManyEngine::push_tasks(Chunk* c, bool final) {
    constexpr uint32_t kSlice = 32;
    std::lock_guard<std::mutex> lk(tmu_);
    for (uint32_t q = 0; q < c->n; q += kSlice) tasks_.push_back(Task{c, q, std::min(c->n, q + kSlice), final});
    tcv_.notify_all();
}

void ManyEngine::final_part_ready(Chunk* c) {
    if (c->final_parts.fetch_add(1) + 1 != 2) return;
    if (!rescored_) {
        push_tasks(c, true);
        return;
    }
    {   // the re-scored pool: the chunk's quality scores first (one gather launch), then the final fusion
        std::lock_guard<std::mutex> lk(mu_);
        rs_q_.push_back(c);
    }
    cv_.notify_all();
}

// SyncQualityPool::RescoredFastPool for a chunk (sync_searcher.rs:814-818 per query): quality_scores_for_hits of every query's fast pool
// — fsgpu_quality_scores_for_hits_batched: one multi-query gather on the quality tier instead of a blocking call per query (a
// sharded pair: the per-query call, routed to the owning shards).  A query whose scores cannot be produced is a RefinementFailed
// outcome of THAT query: when the batched call fails the chunk is re-scored query by query to find which.
void ManyEngine::rescore_stage() {
    std::vector<Hit> hits;
    std::vector<std::vector<Hit>> all;
    std::vector<fsgpu_scored_doc> flat;
    std::vector<uint32_t> offs;
    std::string detail;
    for (;;) {
        Chunk* c = pop(rs_q_);
        if (!c) return;
        const auto b0 = clk::now();
        if (c->status.load() == FSGPU_OK) {
            const uint32_t n = c->n, fetch = c->fetch;
            c->qscores.assign((size_t)n * fetch + 1, 0.0f);
            c->qpresent.assign((size_t)n * fetch + 1, 0);
            c->q_refinement_failed.assign(n, 0);
            all.resize(n);
            flat.clear();
            offs.assign(1, 0);
            fsgpu_status st = FSGPU_OK;
            for (uint32_t i = 0; i < n && st == FSGPU_OK; ++i) {
                st = s_.hits_from_rows(s_.fast_, c->f_rows.data() + (size_t)i * fetch, c->f_scores.data() + (size_t)i * fetch, c->f_counts[i], &all[i], &detail);
                for (const Hit& h : all[i]) flat.push_back(fsgpu_scored_doc{h.doc_id.data(), (uint32_t)h.doc_id.size(), h.score, h.index});
                offs.push_back((uint32_t)flat.size());
            }
            if (st != FSGPU_OK) {
                int expect = FSGPU_OK;
                if (c->status.compare_exchange_strong(expect, st)) c->detail = detail;
            } else {
                // (flat scores: query i's hit j at offs[i] + j; the fusion reads them at i * fetch + j)
                std::vector<float> sc(flat.size() + 1);
                std::vector<uint8_t> pr(flat.size() + 1);
                bool batched_ok = false;
                if (s_.fast_.index) {
                    batched_ok = fsgpu_quality_scores_for_hits_batched(s_.fast_.index, s_.quality_.index, s_.alignment_, c->qvec.data(), n, qdim_,
                                                                       flat.data(), offs.data(), sc.data(), pr.data()) == FSGPU_OK;
                }
                for (uint32_t i = 0; i < n; ++i) {
                    const uint32_t h0 = offs[i], cnt = offs[i + 1] - h0;
                    if (!batched_ok) {
                        const fsgpu_status s1 =
                            s_.fast_.index ? fsgpu_quality_scores_for_hits(s_.fast_.index, s_.quality_.index, s_.alignment_, c->qvec.data() + (size_t)i * qdim_, qdim_,
                                                                           flat.data() + h0, cnt, sc.data() + h0, pr.data() + h0)
                                           : fsgpu_sharded_quality_scores_for_hits(s_.fast_.sharded, s_.quality_.sharded, s_.alignment_,
                                                                                   c->qvec.data() + (size_t)i * qdim_, qdim_, flat.data() + h0, cnt, sc.data() + h0,
                                                                                   pr.data() + h0);
                        if (s1 != FSGPU_OK) {
                            c->q_refinement_failed[i] = 1;
                            if (c->quality_detail.empty()) c->quality_detail = fsgpu_last_error();
                            continue;
                        }
                    }
                    std::memcpy(c->qscores.data() + (size_t)i * fetch, sc.data() + h0, (size_t)cnt * 4);
                    std::memcpy(c->qpresent.data() + (size_t)i * fetch, pr.data() + h0, cnt);
                }
            }
        }
        if (c->batch) {
            std::lock_guard<std::mutex> lk(c->batch->mu);
            c->batch->t_qs += ms_between(b0, clk::now());
            if (c->index == 0) c->batch->first_refined_ms = ms_between(c->batch->t_start, clk::now());
        }
        push_tasks(c, true);
    }
}

void ManyEngine::fusion_worker() {
    std::vector<Hit> fast_hits, quality_hits;
    std::vector<fshost_hit> fused;
    std::string detail;
    for (;;) {
        Task t;
        {
            std::unique_lock<std::mutex> lk(tmu_);
            tcv_.wait(lk, [&] { return pool_stop_ || !tasks_.empty(); });
            if (tasks_.empty()) return;
            t = tasks_.front();
            tasks_.pop_front();
        }
        Chunk* c = t.c;
        const uint32_t k = c->k, fetch = c->fetch;
        const auto b0 = clk::now();
        auto emit = [&](fshost_hit* out, uint32_t* n_out) {
            const uint32_t n = (uint32_t)std::min<size_t>(fused.size(), k);
            if (n && out) std::memcpy(out, fused.data(), (size_t)n * sizeof(fshost_hit));
            if (n_out) *n_out = n;
        };
        for (uint32_t i = t.q0; i < t.q1; ++i) {
            const Query& q = c->q[i];
            fsgpu_status st = (fsgpu_status)c->status.load();
            bool refinement_failed = false;
            if (st == FSGPU_OK)
                st = s_.hits_from_rows(s_.fast_, c->f_rows.data() + (size_t)i * fetch, c->f_scores.data() + (size_t)i * fetch, c->f_counts[i], &fast_hits, &detail);
            else
                detail = c->detail;
            if (st == FSGPU_OK && !t.final) {
                st = s_.fuse_initial(fast_hits, k, q.lexical, q.n_lexical, &fused, &detail);
                if (st == FSGPU_OK) emit(q.initial_out, q.n_initial);
            } else if (st == FSGPU_OK) {
                refinement_failed = c->quality_failed;
                if (!refinement_failed && rescored_) {
                    refinement_failed = c->q_refinement_failed.empty() || c->q_refinement_failed[i] != 0;
                    if (!refinement_failed)
                        st = s_.fuse_final_rescored_scores(fast_hits, c->qscores.data() + (size_t)i * fetch, c->qpresent.data() + (size_t)i * fetch, k, q.lexical,
                                                           q.n_lexical, &fused, &detail);
                } else if (!refinement_failed) {
                    st = s_.hits_from_rows(s_.quality_, c->q_rows.data() + (size_t)i * fetch, c->q_scores.data() + (size_t)i * fetch, c->q_counts[i], &quality_hits,
                                           &detail);
                    if (st != FSGPU_OK) {   // (a doc id the quality tier cannot resolve: the pool's failure, sync_searcher.rs:820-839)
                        refinement_failed = true;
                        st = FSGPU_OK;
                    } else {
                        st = s_.fuse_final_retrieved(fast_hits, quality_hits, k, q.lexical, q.n_lexical, &fused, &detail);
                    }
                }
                if (st == FSGPU_OK && refinement_failed) {   // final_results = the initial results (recomputed: its own task may not have run yet)
                    st = s_.fuse_initial(fast_hits, k, q.lexical, q.n_lexical, &fused, &detail);
                    c->n_refinement_failed.fetch_add(1);
                }
                if (st == FSGPU_OK) {
                    emit(q.final_out, q.n_final);
                    if (q.refinement_failed) *q.refinement_failed = refinement_failed ? 1 : 0;
                }
            }
            if (st != FSGPU_OK) {   // a failing fusion (a doc id too long for fshost_hit, ...) fails the chunk's call
                int expect = FSGPU_OK;
                if (c->status.compare_exchange_strong(expect, st)) c->detail = detail;
            }
            if (q.waiter) {   // a batched per-query caller: ONE wake-up, by whichever of its two fusion tasks runs second
                Waiter* w = q.waiter;
                std::lock_guard<std::mutex> lk(w->mu);
                if (st != FSGPU_OK && w->status == FSGPU_OK) {
                    w->status = st;
                    w->detail = detail;
                }
                if (t.final) {
                    w->refinement_failed = refinement_failed;
                    if (refinement_failed) w->skip_reason = c->quality_detail;
                    w->t_final = clk::now();
                } else {
                    w->t_initial = clk::now();
                }
                if (++w->parts == 2) {
                    w->done = true;
                    w->cv.notify_one();
                    in_flight_.fetch_sub(1);
                    rcv_.notify_one();   // (the collector's "how many to wait for" moves with the completions)
                }
            }
        }
        c->fusion_us.fetch_add((uint64_t)(ms_between(b0, clk::now()) * 1e3));
        if (c->tasks_left.fetch_sub(1) == 1) finish_chunk(c);
    }
}

// Every task of the chunk has run (tasks_left counts the initial AND the final slices).
void ManyEngine::finish_chunk(Chunk* c) {
    Batch* b = c->batch;
    if (b) {
        std::lock_guard<std::mutex> lk(b->mu);
        if (c->status.load() != FSGPU_OK && b->status == FSGPU_OK) {
            b->status = (fsgpu_status)c->status.load();
            b->detail = c->detail;
        }
        if (c->quality_failed && b->refinement_detail.empty()) b->refinement_detail = c->quality_detail;
        b->refinement_failed += c->n_refinement_failed.load();
        b->fusion_busy_ms += (double)c->fusion_us.load() * 1e-3;
        --b->chunks_left;
        b->cv.notify_all();
    }
    delete c;
}

// ---- front end 1: arrays in, arrays out -------------------------------------------------------------------------------------------
fsgpu_status ManyEngine::run_many(const SyncTwoTierSearcher::ManyArgs& a, fshost_many_result* res) {
    auto keep_detail = [&](const std::string& d) {
        std::strncpy(res->error_detail, d.c_str(), sizeof(res->error_detail) - 1);
        res->error_detail[sizeof(res->error_detail) - 1] = 0;
    };
    const uint32_t nq = a.nq, k = a.k;
    const uint32_t mult = std::max<uint32_t>(s_.cfg_.candidate_multiplier, 1);
    const uint64_t wide_fetch = (uint64_t)k * mult;   // candidate_count (rrf.rs:113-115)
    const uint32_t fetch = std::max<uint32_t>(wide_fetch > 0xffffffffull ? 0xffffffffu : (uint32_t)wide_fetch, k);
    const uint32_t C = std::min(nq, a.chunk ? a.chunk : 1024u);
    const uint32_t n_chunks = (nq + C - 1) / C;
    Batch batch;
    batch.chunks_left = n_chunks;
    batch.t_start = clk::now();
    for (uint32_t ci = 0; ci < n_chunks; ++ci) {
        const uint32_t q0 = ci * C, q1 = std::min(nq, q0 + C);
        std::unique_ptr<Chunk> c(new Chunk);
        c->n = q1 - q0;
        c->k = k;
        c->fetch = fetch;
        c->index = ci;
        c->batch = &batch;
        c->q.resize(c->n);
        c->fast_offs.resize(c->n + 1);
        c->quality_offs.resize(c->n + 1);
        for (uint32_t i = 0; i <= c->n; ++i) {
            c->fast_offs[i] = a.fast_offsets[q0 + i] - a.fast_offsets[q0];
            c->quality_offs[i] = a.quality_offsets[q0 + i] - a.quality_offsets[q0];
        }
        if (c->fast_offs[c->n]) c->fast_ids.assign(a.fast_ids + a.fast_offsets[q0], a.fast_ids + a.fast_offsets[q1]);
        if (c->quality_offs[c->n]) c->quality_ids.assign(a.quality_ids + a.quality_offsets[q0], a.quality_ids + a.quality_offsets[q1]);
        for (uint32_t i = 0; i < c->n; ++i) {
            Query& q = c->q[i];
            const uint32_t g = q0 + i;
            if (a.lexical && a.lexical_offsets) {
                q.lexical = a.lexical + a.lexical_offsets[g];
                q.n_lexical = a.lexical_offsets[g + 1] - a.lexical_offsets[g];
            }
            q.initial_out = a.initial_out ? a.initial_out + (size_t)g * k : nullptr;
            q.n_initial = a.n_initial + g;
            q.final_out = a.final_out ? a.final_out + (size_t)g * k : nullptr;
            q.n_final = a.n_final + g;
            q.refinement_failed = a.refinement_failed ? a.refinement_failed + g : nullptr;
            q.fast_vec_out = a.fast_vectors_out ? a.fast_vectors_out + (size_t)g * fdim_ : nullptr;
            q.quality_vec_out = a.quality_vectors_out ? a.quality_vectors_out + (size_t)g * qdim_ : nullptr;
        }
        c->tasks_left.store(2 * ((c->n + 31) / 32));
        c->slot = acquire_slot(C);
        if (c->slot < 0) {   // the engine is shutting down
            std::lock_guard<std::mutex> lk(batch.mu);
            batch.chunks_left -= n_chunks - ci;
            if (batch.status == FSGPU_OK) batch.status = FSGPU_ERR_DEVICE;
            break;
        }
        submit(c.release());
    }
    {
        std::unique_lock<std::mutex> lk(batch.mu);
        batch.cv.wait(lk, [&] { return batch.chunks_left == 0; });
    }
    const double wall = std::chrono::duration<double>(clk::now() - batch.t_start).count();
    res->wall_seconds = wall;
    res->queries = nq;
    res->chunks = n_chunks;
    res->chunk_queries = C;
    res->queries_per_sec = wall > 0 ? nq / wall : 0.0;
    res->mean_fast_embed_ms = batch.t_fe / n_chunks;
    res->mean_fast_search_ms = batch.t_fs / n_chunks;
    res->mean_quality_embed_ms = batch.t_qe / n_chunks;
    res->mean_quality_search_ms = batch.t_qs / n_chunks;
    res->fusion_busy_ms_per_chunk = batch.fusion_busy_ms / n_chunks;
    res->fusion_threads = n_workers_;
    res->first_chunk_initial_ms = batch.first_initial_ms;
    res->first_chunk_refined_ms = batch.first_refined_ms;
    res->refinement_failed = batch.refinement_failed;
    res->fast_fallbacks = batch.fb_fast;
    res->quality_fallbacks = batch.fb_quality;
    res->device_resident_handoff = (batch.fast_on_dev ? 1u : 0u) | (batch.quality_on_dev ? 2u : 0u);
    if (batch.status != FSGPU_OK) {
        keep_detail(batch.detail);
        return batch.status;
    }
    if (!batch.refinement_detail.empty()) keep_detail("refinement failed: " + batch.refinement_detail);
    return FSGPU_OK;
}

// ---- front end 2: concurrent per-query callers --------------------------------------------------------------------------------------
void ManyEngine::configure_batching(uint32_t max_chunk, uint32_t max_wait_us) {
    std::lock_guard<std::mutex> lk(rmu_);
    max_chunk_ = std::min<uint32_t>(max_chunk, 4096);
    max_wait_us_ = max_wait_us;
    if (max_chunk_ && !collector_.joinable()) {
        collector_ = std::thread([this] { collector(); });
        lone_thread_ = std::thread([this] { lone_lane(); });
    }
}

// Whatever is queued becomes the next chunk: as many requests of the head's k as there are (up to max_chunk), taken as soon as a set of
// device buffers is free — requests keep arriving while the chunks before it run, so under load a chunk is as large as the arrival
// rate makes it, and a lone caller's query leaves at once.  With fewer than max_chunk queued the collector lingers while requests are
// still arriving (until the oldest has waited max_wait_us, or arrivals have paused for an eighth of that).
void ManyEngine::collector() {
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(rmu_);
            rcv_.wait(lk, [&] { return server_stop_ || !requests_.empty(); });
            if (server_stop_) {
                for (Waiter* w : requests_) {
                    std::lock_guard<std::mutex> wl(w->mu);
                    w->status = FSGPU_ERR_DEVICE;
                    w->detail = "the searcher is being destroyed";
                    w->done = true;
                    w->cv.notify_one();
                }
                requests_.clear();
                return;
            }
        }
        const int slot = acquire_slot(std::max<uint32_t>(max_chunk_, 1));
        if (slot < 0) continue;   // (shutting down: the loop's head answers the parked requests)
        std::unique_ptr<Chunk> c(new Chunk);
        {
            std::unique_lock<std::mutex> lk(rmu_);
            const uint32_t cap = std::max<uint32_t>(max_chunk_, 1);
            const auto gap = std::chrono::nanoseconds(std::min<int64_t>((int64_t)max_wait_us_ * 1000 / 8, 40'000));
            // How many to wait for: callers come back when their chunk completes, so the population in play is what is queued plus
            // what is in flight.  A tier's batched search costs the same for 16 queries as for 256 (one pass over the slab + the
            // selections: 0.8 / 1.0 ms at 10M rows), so up to the pass width the WHOLE population rides one chunk — 64 callers: one
            // chunk of 64, not two of 32 at twice the GPU time; above it HALF the population per chunk keeps two chunks alternating
            // between the GPU and the fusion threads (a thousand callers: the cap).  Nothing in flight: whoever is here goes once
            // arrivals pause.  Never longer than max_wait_us past the oldest request.
            constexpr size_t kPassWidth = 256;
            while (!server_stop_ && !requests_.empty() && requests_.size() < cap) {
                const uint64_t in_flight = in_flight_.load(std::memory_order_relaxed);
                const size_t pop = requests_.size() + in_flight;
                const size_t want = std::min<size_t>(cap, std::max<size_t>(1, pop <= kPassWidth ? pop : std::max(kPassWidth, pop / 2)));
                const auto now = clk::now();
                const auto deadline = requests_.front()->t_submit + std::chrono::microseconds(max_wait_us_);
                if (now >= deadline) break;
                if (in_flight == 0) {
                    const auto quiet = last_arrival_ + gap;
                    if (now >= quiet) break;
                    rcv_.wait_until(lk, quiet < deadline ? quiet : deadline);
                } else {
                    if (requests_.size() >= want) break;
                    rcv_.wait_until(lk, deadline);   // (arrivals and completions both notify)
                }
            }
            if (requests_.empty() || server_stop_) {
                lk.unlock();
                release_slot(slot);
                continue;
            }
            if (requests_.size() == 1 && !lone_busy_.load()) {   // a lone request: the per-query flow's latency lanes, on the lone-lane thread
                lone_busy_.store(true);
                Waiter* w = requests_.front();
                requests_.pop_front();
                ++server_chunks_;
                ++server_requests_;
                in_flight_.fetch_add(1);
                lk.unlock();
                release_slot(slot);
                {
                    std::lock_guard<std::mutex> ll(lmu_);
                    lone_q_.push_back(w);
                }
                lcv_.notify_one();
                continue;
            }
            const uint32_t k = requests_.front()->k;
            for (auto it = requests_.begin(); it != requests_.end() && c->q.size() < cap;) {
                if ((*it)->k == k) {
                    (*it)->q.waiter = *it;
                    c->q.push_back((*it)->q);
                    it = requests_.erase(it);
                } else {
                    ++it;
                }
            }
            c->k = k;
            ++server_chunks_;
            server_requests_ += c->q.size();
            in_flight_.fetch_add(c->q.size());
        }
        c->n = (uint32_t)c->q.size();
        const uint32_t mult = std::max<uint32_t>(s_.cfg_.candidate_multiplier, 1);
        const uint64_t wide_fetch = (uint64_t)c->k * mult;
        c->fetch = std::max<uint32_t>(wide_fetch > 0xffffffffull ? 0xffffffffu : (uint32_t)wide_fetch, c->k);
        c->fast_offs.assign(1, 0);
        c->quality_offs.assign(1, 0);
        for (const Query& q : c->q) {
            c->fast_ids.insert(c->fast_ids.end(), q.fast_ids, q.fast_ids + q.n_fast);
            c->fast_offs.push_back((uint32_t)c->fast_ids.size());
            c->quality_ids.insert(c->quality_ids.end(), q.quality_ids, q.quality_ids + q.n_quality);
            c->quality_offs.push_back((uint32_t)c->quality_ids.size());
        }
        c->tasks_left.store(2 * ((c->n + 31) / 32));
        c->slot = slot;
        submit(c.release());
    }
}

fsgpu_status ManyEngine::search_one(const uint32_t* fast_ids, uint32_t n_fast, const int32_t* quality_ids, uint32_t n_quality, uint32_t k,
                                    const fsgpu_scored_doc* lexical, uint32_t n_lexical, Outcome* out, std::string* detail) {
    Waiter w;
    w.k = k;
    w.t_submit = clk::now();
    out->initial.assign(k ? k : 1, fshost_hit{});
    out->final_results.assign(k ? k : 1, fshost_hit{});
    uint32_t ni = 0, nf = 0;
    uint8_t rf = 0;
    Query& q = w.q;
    q.fast_ids = fast_ids;
    q.n_fast = n_fast;
    q.quality_ids = quality_ids;
    q.n_quality = n_quality;
    q.lexical = lexical;
    q.n_lexical = n_lexical;
    q.initial_out = out->initial.data();
    q.n_initial = &ni;
    q.final_out = out->final_results.data();
    q.n_final = &nf;
    q.refinement_failed = &rf;
    {
        std::lock_guard<std::mutex> lk(rmu_);
        if (server_stop_ || !collector_.joinable()) {
            *detail = "dynamic batching is off";
            return FSGPU_ERR_INVALID_CONFIG;
        }
        requests_.push_back(&w);
        last_arrival_ = w.t_submit;
    }
    rcv_.notify_one();
    {
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.done; });
    }
    if (w.status != FSGPU_OK) {
        *detail = w.detail;
        return w.status;
    }
    out->initial.resize(ni);
    out->final_results.resize(nf);
    out->refinement_failed = w.refinement_failed;
    out->skip_reason = w.skip_reason;
    fshost_metrics& m = out->metrics;
    m = fshost_metrics{};
    m.phase1_total_ms = ms_between(w.t_submit, w.t_initial);   // query start -> Initial results written
    m.phase2_total_ms = std::max(0.0, ms_between(w.t_initial, w.t_final));
    return FSGPU_OK;
}

// A chunk of ONE request: the per-query flow as fshost_two_tier_search runs it without batching (lone-query lanes of the tiers, the
// single-text encoder kernels) — a lone caller pays nothing for the batching being on.
void ManyEngine::lone_lane() {
    for (;;) {
        Waiter* w;
        {
            std::unique_lock<std::mutex> lk(lmu_);
            lcv_.wait(lk, [&] { return lone_stop_ || !lone_q_.empty(); });
            if (lone_q_.empty()) return;
            w = lone_q_.front();
            lone_q_.pop_front();
        }
        Outcome out;
        std::string detail;
        const Query& q = w->q;
        const fsgpu_status st = s_.search_unbatched(q.fast_ids, q.n_fast, q.quality_ids, q.n_quality, w->k, q.lexical, q.n_lexical, &out, &detail);
        const auto now = clk::now();
        if (st == FSGPU_OK) {
            const uint32_t ni = (uint32_t)std::min<size_t>(out.initial.size(), w->k), nf = (uint32_t)std::min<size_t>(out.final_results.size(), w->k);
            if (ni) std::memcpy(q.initial_out, out.initial.data(), (size_t)ni * sizeof(fshost_hit));
            if (nf) std::memcpy(q.final_out, out.final_results.data(), (size_t)nf * sizeof(fshost_hit));
            *q.n_initial = ni;
            *q.n_final = nf;
        }
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->status = st;
            w->detail = detail;
            w->refinement_failed = out.refinement_failed;
            w->skip_reason = out.skip_reason;
            w->t_final = now;
            w->t_initial = w->t_submit + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double, std::milli>(
                                             ms_between(w->t_submit, now) - out.metrics.phase2_total_ms));
            w->done = true;
            w->cv.notify_one();
        }
        lone_busy_.store(false);
        in_flight_.fetch_sub(1);
        rcv_.notify_one();
    }
}

void ManyEngine::batching_stats(uint64_t* chunks, uint64_t* requests) {
    std::lock_guard<std::mutex> lk(rmu_);
    *chunks = server_chunks_;
    *requests = server_requests_;
}

// ---- SyncTwoTierSearcher's side ------------------------------------------------------------------------------------------------------
ManyEngine* SyncTwoTierSearcher::engine(uint32_t fusion_threads) const {
    std::lock_guard<std::mutex> lk(engine_mu_);
    if (!engine_) engine_.reset(new ManyEngine(*this, fusion_threads));
    return engine_.get();
}

fsgpu_status SyncTwoTierSearcher::search_many(const ManyArgs& a, fshost_many_result* res) const {
    std::memset(res, 0, sizeof *res);
    if (init_status_ != FSGPU_OK) {
        std::strncpy(res->error_detail, init_detail_.c_str(), sizeof(res->error_detail) - 1);
        return init_status_;
    }
    if (a.nq == 0) return FSGPU_OK;
    return engine(a.fusion_threads)->run_many(a, res);
}

fsgpu_status SyncTwoTierSearcher::set_batching(uint32_t max_chunk, uint32_t max_wait_us) {
    if (init_status_ != FSGPU_OK) return init_status_;
    if (max_chunk == 0) batching_.store(false);   // (off: new callers take the per-query flow at once; what is queued is still answered)
    engine(0)->configure_batching(max_chunk, max_wait_us);
    if (max_chunk != 0) batching_.store(true);
    return FSGPU_OK;
}

void SyncTwoTierSearcher::batching_stats(uint64_t* chunks, uint64_t* requests) const {
    *chunks = *requests = 0;
    std::lock_guard<std::mutex> lk(engine_mu_);
    if (engine_) engine_->batching_stats(chunks, requests);
}

void SyncTwoTierSearcher::stop_engine() {
    std::lock_guard<std::mutex> lk(engine_mu_);
    engine_.reset();
}

}  // namespace fshost
