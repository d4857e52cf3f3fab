// two_tier_many.cpp — SyncTwoTierSearcher::search for MANY queries at once (include/fshost.h, fshost_two_tier_search_many).
//
// The reference's two-phase flow is per query (crates/frankensearch-fusion/src/sync_searcher.rs:616-943: fast embed -> fast-tier
// top-(k x multiplier) -> RRF = Initial; quality embed -> quality-tier top-(k x multiplier) -> blend -> RRF = Refined); its
// many-queries shapes are the batched scan bench (crates/frankensearch-index/benches/batched_query_scan.rs) and the embedder's batch
// coalescer (crates/frankensearch-embed/src/batch_coalescer.rs:18-23).  A GPU wants the whole flow in that shape: a thousand
// blocking per-query calls from a thousand host threads keep the device a fifth busy (bench.py, two_tier.concurrent_1024_threads).
//
// Here the nq queries go through the same stages in CHUNKS of 1,024 (two 512-query passes of the matrix-core scan), as a pipeline of
// four stage threads over the C ABI — every stage a batched call the library already has —
//     FE  Model2Vec batch            fsgpu_m2v_embed_device            (vectors stay in HBM)
//     FS  fast tier, batched         fsgpu_search_topk_int8_two_pass_batched_device_queries / fsgpu_search_topk_batched_device_queries /
//                                    fsgpu_sharded_search(INT8_TWO_PASS | BATCHED, queries_dev)
//     QE  MiniLM batch               fsgpu_bert_embed_device
//     QS  quality tier, batched      fsgpu_search_topk_batched_device_queries / fsgpu_sharded_search(BATCHED, queries_dev)
// — FE / QE one chunk ahead of FS / QS (two buffers each), the two tiers side by side on the GPU (the quality tier needs nothing of
// phase 0 in the Retrieved pool), and a pool of host threads that runs the PER-QUERY fusion of a chunk as soon as its inputs exist:
// the very functions SyncTwoTierSearcher::search runs (hits_from_rows, fuse_initial, fuse_final_retrieved / _rescored), so a query's
// results are those of the per-query call on the same tier answers — and the tier answers are the per-query searches' rows and score
// bits (the batched searches are bit-identical to the exact kernels; Model2Vec is bit-exact whatever the batch; the MiniLM embedding of
// a text is within the encoder's stated tolerance of its single-text embedding: the encoder picks kernels by batch shape).
// Host code only: calls nothing but include/fsgpu.h.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include "two_tier_searcher.hpp"

namespace fshost {

namespace {

using clock_t_ = std::chrono::steady_clock;
double ms_between(clock_t_::time_point a, clock_t_::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

struct DevBuf {   // two device buffers of one embedder's chunk output (freed on scope exit)
    int32_t dev = -1;
    float* p[2] = {nullptr, nullptr};
    ~DevBuf() {
        for (float* q : p)
            if (q) (void)fsgpu_device_free(dev, q);
    }
    bool alloc(int32_t device, uint64_t bytes) {
        dev = device;
        for (float*& q : p)
            if (fsgpu_device_malloc(device, bytes, reinterpret_cast<void**>(&q)) != FSGPU_OK) return false;
        return true;
    }
};

}  // namespace

fsgpu_status SyncTwoTierSearcher::search_many(const ManyArgs& a, fshost_many_result* res) const {
    std::memset(res, 0, sizeof *res);
    auto keep_detail = [&](const std::string& d) {
        std::strncpy(res->error_detail, d.c_str(), sizeof(res->error_detail) - 1);
        res->error_detail[sizeof(res->error_detail) - 1] = 0;
    };
    if (init_status_ != FSGPU_OK) {
        keep_detail(init_detail_);
        return init_status_;
    }
    const uint32_t nq = a.nq, k = a.k;
    if (nq == 0) return FSGPU_OK;
    const uint32_t mult = std::max<uint32_t>(cfg_.candidate_multiplier, 1);
    const uint64_t wide_fetch = (uint64_t)k * mult;   // candidate_count (rrf.rs:113-115)
    const uint32_t fetch = std::max<uint32_t>(wide_fetch > 0xffffffffull ? 0xffffffffu : (uint32_t)wide_fetch, k);
    const uint32_t C = std::min(nq, a.chunk ? a.chunk : 1024u);
    const uint32_t n_chunks = (nq + C - 1) / C;
    const bool rescored = cfg_.quality_pool == FSHOST_POOL_RESCORED;
    const uint32_t fdim = fsgpu_m2v_dimension(m2v_), qdim = fsgpu_bert_dimension(bert_);
    // which calls answer a tier for a chunk: the batched forms of what search() calls per query (tier_hits): row-level searches when
    // the ids are synthetic or the fast tier runs the int8 two-pass; with doc-id tables the exact searches go through search_hits
    // (WAL merge, shadowing, dedup) query by query, as search() does
    const bool fast_rowlevel = cfg_.fast_tier_int8_multiplier != 0 || cfg_.doc_id_mode != 0;
    const bool quality_rowlevel = cfg_.doc_id_mode != 0;
    // embeddings stay in device memory when the embedder sits on the tier's (root) device and the caller does not want them back
    const int32_t fast_dev = fast_.index ? fsgpu_index_device(fast_.index) : fsgpu_sharded_device(fast_.sharded, 0);
    const int32_t quality_dev = quality_.index ? fsgpu_index_device(quality_.index) : fsgpu_sharded_device(quality_.sharded, 0);
    bool fast_on_dev = fast_rowlevel && !a.fast_vectors_out && fsgpu_m2v_device(m2v_) == fast_dev && fast_dev >= 0;
    bool quality_on_dev = !rescored && quality_rowlevel && !a.quality_vectors_out && fsgpu_bert_device(bert_) == quality_dev && quality_dev >= 0;
    DevBuf fbuf, qbuf;
    if (fast_on_dev && !fbuf.alloc(fast_dev, (uint64_t)C * fdim * 4)) fast_on_dev = false;
    if (quality_on_dev && !qbuf.alloc(quality_dev, (uint64_t)C * qdim * 4)) quality_on_dev = false;
    // host-side vectors: the caller's arrays when it asked for them, else a two-chunk ring (the re-scored pool reads a query's quality
    // vector from the fusion threads: a full array)
    std::vector<float> fvec_own, qvec_own;
    float* fvec_all = a.fast_vectors_out;
    float* qvec_all = a.quality_vectors_out;
    if (!fast_on_dev && !fvec_all) fvec_own.resize((size_t)2 * C * fdim);
    if (!quality_on_dev && !qvec_all) {
        if (rescored) {
            qvec_own.resize((size_t)nq * qdim);
            qvec_all = qvec_own.data();
        } else {
            qvec_own.resize((size_t)2 * C * qdim);
        }
    }
    auto fvec_host = [&](uint32_t c) { return fvec_all ? fvec_all + (size_t)c * C * fdim : fvec_own.data() + (size_t)(c & 1) * C * fdim; };
    auto qvec_host = [&](uint32_t c) { return qvec_all ? qvec_all + (size_t)c * C * qdim : qvec_own.data() + (size_t)(c & 1) * C * qdim; };
    // the tiers' answers, whole arrays: the fusion of chunk c reads them while the stages are chunks ahead
    std::vector<uint32_t> f_rows((size_t)nq * fetch), f_counts(nq), q_rows(rescored ? 0 : (size_t)nq * fetch), q_counts(rescored ? 0 : nq);
    std::vector<float> f_scores((size_t)nq * fetch), q_scores(rescored ? 0 : (size_t)nq * fetch);
    std::vector<uint8_t> chunk_quality_failed(n_chunks, 0);
    std::vector<std::string> chunk_quality_detail(n_chunks);

    std::mutex mu;
    std::condition_variable cv;
    int64_t fe_done = -1, fs_done = -1, qe_done = -1, qs_done = -1;
    bool stop = false;
    fsgpu_status failed = FSGPU_OK;
    std::string failed_detail;
    double t_fe = 0, t_fs = 0, t_qe = 0, t_qs = 0;
    uint64_t fb_fast = 0, fb_quality = 0;
    auto fail = [&](fsgpu_status st, const std::string& d) {   // (under mu)
        if (failed == FSGPU_OK) {
            failed = st;
            failed_detail = d;
        }
        stop = true;
    };
    auto chunk_range = [&](uint32_t c) { return std::pair<uint32_t, uint32_t>(c * C, std::min(nq, (c + 1) * C)); };

    // ---- the fusion pool ----------------------------------------------------------------------------------------------------------
    struct Task {
        uint32_t q0, q1;
        bool final;
    };
    std::mutex tmu;
    std::condition_variable tcv, tdone;
    std::deque<Task> tasks;
    bool pool_stop = false;
    uint64_t tasks_open = 0;
    std::atomic<uint64_t> n_refinement_failed{0};
    std::atomic<int32_t> fusion_failed{FSGPU_OK};
    std::string fusion_detail;
    std::atomic<uint64_t> fusion_busy_us{0};
    std::vector<std::atomic<uint8_t>> parts(n_chunks);   // inputs of a chunk's FINAL fusion that exist (2: both)
    for (auto& p : parts) p.store(0);
    auto push_tasks = [&](uint32_t c, bool final) {
        const auto [q0, q1] = chunk_range(c);
        constexpr uint32_t kSlice = 32;
        std::lock_guard<std::mutex> lk(tmu);
        for (uint32_t q = q0; q < q1; q += kSlice) {
            tasks.push_back(Task{q, std::min(q1, q + kSlice), final});
            ++tasks_open;
        }
        tcv.notify_all();
    };
    auto final_part_ready = [&](uint32_t c) {
        if (parts[c].fetch_add(1) + 1 == 2) push_tasks(c, true);
    };
    auto lexical_of = [&](uint32_t q, const fsgpu_scored_doc** lex, uint32_t* n) {
        if (a.lexical && a.lexical_offsets) {
            *lex = a.lexical + a.lexical_offsets[q];
            *n = a.lexical_offsets[q + 1] - a.lexical_offsets[q];
        } else {
            *lex = nullptr;
            *n = 0;
        }
    };
    auto emit = [&](const std::vector<fshost_hit>& hits, fshost_hit* out, uint32_t* n_out, uint32_t q) {
        const uint32_t n = (uint32_t)std::min<size_t>(hits.size(), k);
        if (n) std::memcpy(out + (size_t)q * k, hits.data(), (size_t)n * sizeof(fshost_hit));
        n_out[q] = n;
    };
    auto worker = [&] {
        std::vector<Hit> fast_hits, quality_hits;
        std::vector<fshost_hit> fused;
        std::string detail;
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(tmu);
                tcv.wait(lk, [&] { return pool_stop || !tasks.empty(); });
                if (tasks.empty()) return;
                t = tasks.front();
                tasks.pop_front();
            }
            const auto b0 = clock_t_::now();
            for (uint32_t q = t.q0; q < t.q1 && fusion_failed.load() == FSGPU_OK; ++q) {
                const fsgpu_scored_doc* lex;
                uint32_t n_lex;
                lexical_of(q, &lex, &n_lex);
                fsgpu_status st = hits_from_rows(fast_, f_rows.data() + (size_t)q * fetch, f_scores.data() + (size_t)q * fetch, f_counts[q], &fast_hits, &detail);
                if (st == FSGPU_OK && !t.final) {
                    st = fuse_initial(fast_hits, k, lex, n_lex, &fused, &detail);
                    if (st == FSGPU_OK) emit(fused, a.initial_out, a.n_initial, q);
                } else if (st == FSGPU_OK) {
                    const uint32_t c = q / C;
                    bool refinement_failed = chunk_quality_failed[c] != 0;
                    if (!refinement_failed && rescored) {
                        st = fuse_final_rescored(fast_hits, qvec_all + (size_t)q * qdim, k, lex, n_lex, &fused, &refinement_failed, &detail);
                        if (refinement_failed) st = FSGPU_OK;
                    } else if (!refinement_failed) {
                        st = hits_from_rows(quality_, q_rows.data() + (size_t)q * fetch, q_scores.data() + (size_t)q * fetch, q_counts[q], &quality_hits, &detail);
                        if (st != FSGPU_OK) {   // (a doc id the quality tier cannot resolve: the pool's failure, sync_searcher.rs:820-839)
                            refinement_failed = true;
                            st = FSGPU_OK;
                        } else {
                            st = fuse_final_retrieved(fast_hits, quality_hits, k, lex, n_lex, &fused, &detail);
                        }
                    }
                    if (st == FSGPU_OK && refinement_failed) {   // final_results = the initial results (recomputed: its own task may not have run yet)
                        st = fuse_initial(fast_hits, k, lex, n_lex, &fused, &detail);
                        n_refinement_failed.fetch_add(1);
                    }
                    if (st == FSGPU_OK) {
                        emit(fused, a.final_out, a.n_final, q);
                        if (a.refinement_failed) a.refinement_failed[q] = refinement_failed ? 1 : 0;
                    }
                }
                if (st != FSGPU_OK) {
                    int32_t expect = FSGPU_OK;
                    if (fusion_failed.compare_exchange_strong(expect, st)) {
                        std::lock_guard<std::mutex> lk(tmu);
                        fusion_detail = detail;
                    }
                }
            }
            fusion_busy_us.fetch_add((uint64_t)(ms_between(b0, clock_t_::now()) * 1e3));
            {
                std::lock_guard<std::mutex> lk(tmu);
                if (--tasks_open == 0) tdone.notify_all();
            }
        }
    };
    const uint32_t hw = std::max(2u, std::thread::hardware_concurrency());
    const uint32_t n_workers = a.fusion_threads ? a.fusion_threads : std::max(2u, std::min(12u, hw > 6 ? hw - 4 : 2u));
    std::vector<std::thread> pool;
    for (uint32_t i = 0; i < n_workers; ++i) pool.emplace_back(worker);

    // ---- the four stages ----------------------------------------------------------------------------------------------------------
    const auto t_start = clock_t_::now();
    double first_initial_ms = 0, first_final_ms = 0;
    auto embed_stage = [&](bool fast) {
        std::vector<uint32_t> offs;
        for (uint32_t c = 0; c < n_chunks; ++c) {
            {
                std::unique_lock<std::mutex> lk(mu);   // buffer c & 1 was last read by the search of chunk c - 2
                cv.wait(lk, [&] { return stop || (fast ? fs_done : (rescored ? (int64_t)c : qs_done)) >= (int64_t)c - 2; });
                if (stop) return;
            }
            const auto [q0, q1] = chunk_range(c);
            const uint32_t n = q1 - q0;
            offs.resize(n + 1);
            const uint32_t* src = fast ? a.fast_offsets : a.quality_offsets;
            for (uint32_t i = 0; i <= n; ++i) offs[i] = src[q0 + i] - src[q0];
            const auto b0 = clock_t_::now();
            fsgpu_status st;
            if (fast)
                st = fast_on_dev ? fsgpu_m2v_embed_device(m2v_, a.fast_ids + src[q0], offs.data(), n, fbuf.p[c & 1])
                                 : fsgpu_m2v_embed(m2v_, a.fast_ids + src[q0], offs.data(), n, fvec_host(c));
            else
                st = quality_on_dev ? fsgpu_bert_embed_device(bert_, a.quality_ids + src[q0], offs.data(), n, qbuf.p[c & 1])
                                    : fsgpu_bert_embed(bert_, a.quality_ids + src[q0], offs.data(), n, qvec_host(c));
            const double d = ms_between(b0, clock_t_::now());
            const std::string detail = st != FSGPU_OK ? std::string(fsgpu_last_error()) : std::string();   // thread-local: read here
            {
                std::lock_guard<std::mutex> lk(mu);
                (fast ? t_fe : t_qe) += d;
                if (st != FSGPU_OK) fail(st, detail);   // an embedding's failure is the search's (embed_sync's error propagates)
                else (fast ? fe_done : qe_done) = c;
            }
            cv.notify_all();
            if (st != FSGPU_OK) return;
            if (!fast && rescored) final_part_ready(c);   // (the re-scored pool's second input is the quality VECTOR)
        }
    };
    // one tier's answer for a chunk: rows / scores / counts [n, fetch] from device or host vectors
    auto tier_search = [&](const Tier& tier, bool rowlevel, uint32_t int8_mult, const float* vec_dev, const float* vec_host, uint32_t n, uint32_t dim,
                           uint32_t* rows, float* scores, uint32_t* counts, uint32_t* fb, std::string* detail) -> fsgpu_status {
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
    };
    std::mutex gpu_turn;
    const bool take_turns = std::getenv("FSHOST_MANY_TURNS") != nullptr;
    auto search_stage = [&](bool fast) {
        for (uint32_t c = 0; c < n_chunks; ++c) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || (fast ? fe_done : qe_done) >= (int64_t)c; });
                if (stop) return;
            }
            const auto [q0, q1] = chunk_range(c);
            const uint32_t n = q1 - q0;
            uint32_t fb = 0;
            std::string detail;
            std::unique_lock<std::mutex> turn(gpu_turn, std::defer_lock);
            if (take_turns) turn.lock();
            const auto b0 = clock_t_::now();
            const fsgpu_status st =
                fast ? tier_search(fast_, fast_rowlevel, cfg_.fast_tier_int8_multiplier, fast_on_dev ? fbuf.p[c & 1] : nullptr, fvec_host(c), n, fdim,
                                   f_rows.data() + (size_t)q0 * fetch, f_scores.data() + (size_t)q0 * fetch, f_counts.data() + q0, &fb, &detail)
                     : tier_search(quality_, quality_rowlevel, 0, quality_on_dev ? qbuf.p[c & 1] : nullptr, qvec_host(c), n, qdim,
                                   q_rows.data() + (size_t)q0 * fetch, q_scores.data() + (size_t)q0 * fetch, q_counts.data() + q0, &fb, &detail);
            const auto b1 = clock_t_::now();
            if (take_turns) turn.unlock();
            {
                std::lock_guard<std::mutex> lk(mu);
                (fast ? t_fs : t_qs) += ms_between(b0, b1);
                (fast ? fb_fast : fb_quality) += fb;
                if (st != FSGPU_OK && fast) {
                    fail(st, detail);   // phase 0 failing fails the search (sync_searcher.rs:652-700)
                } else {
                    if (st != FSGPU_OK) {   // the quality pool failing is a RefinementFailed outcome of the chunk's queries (:820-839)
                        chunk_quality_failed[c] = 1;
                        chunk_quality_detail[c] = detail;
                    }
                    (fast ? fs_done : qs_done) = c;
                    if (c == 0) (fast ? first_initial_ms : first_final_ms) = ms_between(t_start, b1);
                }
            }
            cv.notify_all();
            if (st != FSGPU_OK && fast) return;
            if (fast) push_tasks(c, false);
            final_part_ready(c);
        }
    };
    std::thread th_fe(embed_stage, true), th_qe(embed_stage, false), th_fs(search_stage, true);
    std::thread th_qs;
    if (!rescored) th_qs = std::thread(search_stage, false);
    th_fe.join();
    th_qe.join();
    th_fs.join();
    if (th_qs.joinable()) th_qs.join();
    {   // the fusion of whatever was pushed, then the pool goes away
        std::unique_lock<std::mutex> lk(tmu);
        tdone.wait(lk, [&] { return tasks_open == 0; });
        pool_stop = true;
    }
    tcv.notify_all();
    for (std::thread& t : pool) t.join();
    const double wall = std::chrono::duration<double>(clock_t_::now() - t_start).count();

    res->wall_seconds = wall;
    res->queries = nq;
    res->chunks = n_chunks;
    res->chunk_queries = C;
    res->queries_per_sec = wall > 0 ? nq / wall : 0.0;
    res->mean_fast_embed_ms = t_fe / n_chunks;
    res->mean_fast_search_ms = t_fs / n_chunks;
    res->mean_quality_embed_ms = t_qe / n_chunks;
    res->mean_quality_search_ms = t_qs / n_chunks;
    res->fusion_busy_ms_per_chunk = (double)fusion_busy_us.load() * 1e-3 / n_chunks;
    res->fusion_threads = n_workers;
    res->first_chunk_initial_ms = first_initial_ms;
    res->first_chunk_refined_ms = first_final_ms;
    res->refinement_failed = n_refinement_failed.load();
    res->fast_fallbacks = fb_fast;
    res->quality_fallbacks = fb_quality;
    res->device_resident_handoff = (fast_on_dev ? 1u : 0u) | (quality_on_dev ? 2u : 0u);
    if (failed != FSGPU_OK) {
        keep_detail(failed_detail);
        return failed;
    }
    if (fusion_failed.load() != FSGPU_OK) {
        keep_detail(fusion_detail);
        return (fsgpu_status)fusion_failed.load();
    }
    for (uint32_t c = 0; c < n_chunks; ++c)
        if (chunk_quality_failed[c]) {
            keep_detail("refinement failed: " + chunk_quality_detail[c]);
            break;
        }
    return FSGPU_OK;
}

}  // namespace fshost
