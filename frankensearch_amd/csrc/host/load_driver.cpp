// load_driver.cpp — closed-loop load generator over SyncTwoTierSearcher: native threads issuing per-query calls the
// way a multi-threaded Rust host would (see include/fshost.h).  Host code only.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "two_tier_searcher.hpp"

namespace fshost {

namespace {

struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
};

double percentile(std::vector<double>& v, double p) {
    if (v.empty()) return 0.0;
    const size_t i = std::min(v.size() - 1, (size_t)(p * (double)v.size()));
    std::nth_element(v.begin(), v.begin() + (long)i, v.end());
    return v[i];
}

struct ThreadStats {
    std::vector<double> phase0_ms, phase1_ms;
    double fast_embed = 0, fast_search = 0, quality_embed = 0, quality_search = 0, fusion = 0;
    uint64_t completed = 0, failed = 0;
    std::string first_error;
};

}  // namespace

fsgpu_status run_load(const SyncTwoTierSearcher& searcher, const fshost_load_config& cfg, fshost_load_result* res) {
    if (cfg.threads == 0 || cfg.k == 0 || cfg.fast_vocab == 0 || cfg.corpus_rows == 0 || cfg.quality_vocab <= 1000)
        return FSGPU_ERR_INVALID_CONFIG;
    const uint32_t nthreads = cfg.threads;
    std::atomic<int64_t> warm_left((int64_t)cfg.warmup_queries), timed_left((int64_t)cfg.queries);
    std::atomic<uint32_t> at_barrier(0);
    std::atomic<bool> go(false);
    std::vector<ThreadStats> stats(nthreads);
    std::chrono::steady_clock::time_point t_start;

    auto worker = [&](uint32_t tid) {
        SplitMix64 rng{cfg.seed * 0x100000001b3ull + tid};
        ThreadStats& st = stats[tid];
        std::vector<uint32_t> fast_ids;
        std::vector<int32_t> quality_ids;
        std::vector<std::string> lex_ids;
        std::vector<fsgpu_scored_doc> lexical;
        Outcome out;
        std::string detail;
        auto one_query = [&](bool timed) {
            const uint32_t nf = 4 + rng.below(20);
            fast_ids.resize(nf);
            for (auto& t : fast_ids) t = rng.below(cfg.fast_vocab);
            const uint32_t nq = 8 + rng.below(25);  // incl. [CLS] and [SEP]
            quality_ids.resize(nq);
            quality_ids[0] = 101;
            for (uint32_t i = 1; i + 1 < nq; ++i) quality_ids[i] = 1000 + (int32_t)rng.below(cfg.quality_vocab - 1000);
            quality_ids[nq - 1] = 102;
            const uint32_t nl = 3 * cfg.k;  // stub lexical list (BM25 stays on the CPU in the reference)
            lex_ids.resize(nl);
            lexical.resize(nl);
            char buf[32];
            for (uint32_t j = 0; j < nl; ++j) {
                const int n = std::snprintf(buf, sizeof buf, "doc-%08u", (uint32_t)(rng.next() % cfg.corpus_rows));
                lex_ids[j].assign(buf, (size_t)n);
                lexical[j] = fsgpu_scored_doc{lex_ids[j].data(), (uint32_t)lex_ids[j].size(), (float)(nl - j), 0u};
            }
            const auto q0 = std::chrono::steady_clock::now();
            const fsgpu_status s = searcher.search(fast_ids.data(), nf, quality_ids.data(), nq, cfg.k, lexical.data(), nl, &out, &detail);
            const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - q0).count();
            if (!timed) return;
            if (s != FSGPU_OK) {
                if (st.failed++ == 0) st.first_error = "status " + std::to_string(s) + ": " + detail;
                return;
            }
            const fshost_metrics& m = out.metrics;
            st.phase0_ms.push_back(m.phase1_total_ms);
            st.phase1_ms.push_back(total);
            st.fast_embed += m.fast_embed_ms;
            st.fast_search += m.fast_search_ms;
            st.quality_embed += m.quality_embed_ms;
            st.quality_search += m.quality_search_ms;
            st.fusion += total - m.fast_embed_ms - m.fast_search_ms - m.quality_embed_ms - m.quality_search_ms;
            ++st.completed;
        };
        while (warm_left.fetch_sub(1) > 0) one_query(false);
        at_barrier.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        while (timed_left.fetch_sub(1) > 0) one_query(true);
    };

    std::vector<std::thread> threads;
    threads.reserve(nthreads);
    for (uint32_t t = 0; t < nthreads; ++t) threads.emplace_back(worker, t);
    while (at_barrier.load() < nthreads) std::this_thread::sleep_for(std::chrono::microseconds(200));
    t_start = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (auto& th : threads) th.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();

    std::vector<double> p0, p1;
    ThreadStats sum;
    for (auto& s : stats) {
        p0.insert(p0.end(), s.phase0_ms.begin(), s.phase0_ms.end());
        p1.insert(p1.end(), s.phase1_ms.begin(), s.phase1_ms.end());
        sum.fast_embed += s.fast_embed;
        sum.fast_search += s.fast_search;
        sum.quality_embed += s.quality_embed;
        sum.quality_search += s.quality_search;
        sum.fusion += s.fusion;
        sum.completed += s.completed;
        sum.failed += s.failed;
    }
    const double n = sum.completed ? (double)sum.completed : 1.0;
    res->wall_seconds = wall;
    res->queries_per_sec = (double)sum.completed / wall;
    res->phase0_p50_ms = percentile(p0, 0.50);
    res->phase0_p95_ms = percentile(p0, 0.95);
    res->phase0_p99_ms = percentile(p0, 0.99);
    res->phase1_p50_ms = percentile(p1, 0.50);
    res->phase1_p95_ms = percentile(p1, 0.95);
    res->phase1_p99_ms = percentile(p1, 0.99);
    res->mean_fast_embed_ms = sum.fast_embed / n;
    res->mean_fast_search_ms = sum.fast_search / n;
    res->mean_quality_embed_ms = sum.quality_embed / n;
    res->mean_quality_search_ms = sum.quality_search / n;
    res->mean_fusion_ms = sum.fusion / n;
    res->completed = sum.completed;
    res->failed = sum.failed;
    res->first_error[0] = 0;
    for (auto& s : stats)
        if (!s.first_error.empty()) {
            std::snprintf(res->first_error, sizeof res->first_error, "%s", s.first_error.c_str());
            break;
        }
    return FSGPU_OK;
}

// The load generator's queries (same distributions, one stream) handed to search_many in ONE call: what a host with a queue of
// requests does.  Timed: the search_many call (warm-up: one call over warmup_queries first).
fsgpu_status run_load_many(const SyncTwoTierSearcher& searcher, const fshost_load_config& cfg, uint32_t chunk, fshost_many_result* res) {
    if (cfg.k == 0 || cfg.fast_vocab == 0 || cfg.corpus_rows == 0 || cfg.quality_vocab <= 1000 || cfg.queries == 0) return FSGPU_ERR_INVALID_CONFIG;
    const uint32_t total = cfg.queries + cfg.warmup_queries;
    SplitMix64 rng{cfg.seed * 0x100000001b3ull + 0x5eed};
    std::vector<uint32_t> fast_ids, fast_off(total + 1, 0), q_off(total + 1, 0), lex_off(total + 1, 0);
    std::vector<int32_t> quality_ids;
    const uint32_t nl = 3 * cfg.k;
    std::vector<char> lex_text((size_t)total * nl * 12);   // "doc-%08u" is 12 characters
    std::vector<fsgpu_scored_doc> lexical((size_t)total * nl);
    for (uint32_t q = 0; q < total; ++q) {
        const uint32_t nf = 4 + rng.below(20);
        for (uint32_t i = 0; i < nf; ++i) fast_ids.push_back(rng.below(cfg.fast_vocab));
        fast_off[q + 1] = (uint32_t)fast_ids.size();
        const uint32_t nq = 8 + rng.below(25);
        quality_ids.push_back(101);
        for (uint32_t i = 1; i + 1 < nq; ++i) quality_ids.push_back(1000 + (int32_t)rng.below(cfg.quality_vocab - 1000));
        quality_ids.push_back(102);
        q_off[q + 1] = (uint32_t)quality_ids.size();
        for (uint32_t j = 0; j < nl; ++j) {
            char* p = lex_text.data() + ((size_t)q * nl + j) * 12;
            char buf[16];
            std::snprintf(buf, sizeof buf, "doc-%08u", (uint32_t)(rng.next() % cfg.corpus_rows));
            std::memcpy(p, buf, 12);
            lexical[(size_t)q * nl + j] = fsgpu_scored_doc{p, 12u, (float)(nl - j), 0u};
        }
        lex_off[q + 1] = (q + 1) * nl;
    }
    std::vector<fshost_hit> ini((size_t)std::max(cfg.queries, cfg.warmup_queries) * cfg.k), fin(ini.size());
    std::vector<uint32_t> ni(std::max(cfg.queries, cfg.warmup_queries)), nf(ni.size());
    SyncTwoTierSearcher::ManyArgs a;
    a.k = cfg.k;
    a.chunk = chunk;
    a.fusion_threads = cfg.threads;
    a.initial_out = ini.data();
    a.n_initial = ni.data();
    a.final_out = fin.data();
    a.n_final = nf.data();
    auto run = [&](uint32_t first, uint32_t n) {
        a.fast_ids = fast_ids.data();
        a.fast_offsets = fast_off.data() + first;
        a.quality_ids = quality_ids.data();
        a.quality_offsets = q_off.data() + first;
        a.lexical = lexical.data();
        a.lexical_offsets = lex_off.data() + first;
        a.nq = n;
        return searcher.search_many(a, res);
    };
    if (cfg.warmup_queries) {
        const fsgpu_status st = run(0, cfg.warmup_queries);
        if (st != FSGPU_OK) return st;
    }
    const fsgpu_status st = run(cfg.warmup_queries, cfg.queries);
    if (st != FSGPU_OK) return st;
    uint64_t full = 0;
    for (uint32_t q = 0; q < cfg.queries; ++q) full += (ni[q] == cfg.k && nf[q] == cfg.k) ? 1 : 0;
    res->queries_with_k_initial_and_refined_hits = full;
    return FSGPU_OK;
}

}  // namespace fshost
