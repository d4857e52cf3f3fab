// stream_pipeline.cpp — BASELINE config 5's serving loop over the fsgpu C ABI: encode batch g + 1 while batch g is searched
// (include/fshost.h, fshost_embed_search_stream).  Host code only: what a Rust host would write with rayon::join around
// SyncEmbed::embed_batch_sync (crates/frankensearch-core/src/traits.rs:401-582) and VectorIndex::search_top_k (search.rs:192).
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "two_tier_searcher.hpp"

namespace fshost {

namespace {
double ms_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
}
}  // namespace

// encoders: ONE handle (on the index's device, or anywhere with the host hand-off) or — over a sharded handle — one per device for a
// data-parallel encode ("Encoders: data-parallel over the query batch"): encoder e takes the e-th contiguous slice of a
// group's texts on ITS device, the embeddings stay where they were computed, and the sharded search fetches every rank's slice of
// its query group peer to peer (fsgpu_sharded_search_parts).  With the encoder on the root device only, at 8 GPUs device 0 encodes
// every batch AND scans its shard: it is the straggler of every step.
fsgpu_status embed_search_stream(fsgpu_bert* const* encoders, uint32_t n_encoders, fsgpu_index* index, fsgpu_sharded* sharded, const int32_t* ids,
                                 const uint32_t* offsets, uint32_t batch, uint32_t n_batches, uint32_t group, uint32_t k, bool overlap,
                                 bool host_handoff, uint32_t* out_rows, float* out_scores, uint32_t* out_counts, fshost_stream_result* result) {
    using clock = std::chrono::steady_clock;
    fsgpu_bert* encoder = encoders[0];
    const bool data_parallel = n_encoders > 1;
    const uint32_t dim = index ? fsgpu_index_dimension(index) : fsgpu_sharded_dimension(sharded);
    const uint32_t n_groups = (n_batches + group - 1) / group;
    const size_t group_texts = (size_t)group * batch;
    // two embedding buffers: the encoder fills one while the search reads the other.  When the encoder and the index (a sharded
    // handle's root) share a device the vectors never leave HBM: fsgpu_bert_embed_device -> device queries of the search
    // (fsgpu_search_topk_batched_device_queries / fsgpu_sharded_request::queries_dev); `host_handoff` keeps them on the host path.
    const int32_t enc_dev = fsgpu_bert_device(encoder);
    const int32_t idx_dev = index ? fsgpu_index_device(index) : fsgpu_sharded_device(sharded, 0);
    bool on_device = data_parallel || (!host_handoff && enc_dev >= 0 && enc_dev == idx_dev);
    // per encoder: its device and its slice's two buffers (data-parallel: slice e of a group = texts [e * per_enc, ...) of the group)
    const size_t per_enc = data_parallel ? (group_texts + n_encoders - 1) / n_encoders : group_texts;
    struct EncBuf {
        int32_t dev = -1;
        float* p[2] = {nullptr, nullptr};
    };
    std::vector<EncBuf> bufs(n_encoders);
    struct FreeDev {
        std::vector<EncBuf>* b;
        ~FreeDev() {
            for (EncBuf& e : *b)
                for (int i = 0; i < 2; ++i)
                    if (e.p[i]) (void)fsgpu_device_free(e.dev, e.p[i]);
        }
    } free_dev{&bufs};
    if (on_device)
        for (uint32_t e = 0; e < n_encoders && on_device; ++e) {
            bufs[e].dev = fsgpu_bert_device(encoders[e]);
            for (int b = 0; b < 2 && on_device; ++b)
                if (fsgpu_device_malloc(bufs[e].dev, (uint64_t)per_enc * dim * 4, reinterpret_cast<void**>(&bufs[e].p[b])) != FSGPU_OK) on_device = false;
        }
    if (data_parallel && !on_device) return FSGPU_ERR_DEVICE;   // (the data-parallel form has no host path)
    std::vector<float> emb[2];
    if (!on_device) emb[0].resize(group_texts * dim), emb[1].resize(group_texts * dim);
    std::vector<uint32_t> rows(group_texts * k), counts(group_texts), local_offsets;
    std::vector<float> scores(group_texts * k);
    std::mutex mu;
    std::condition_variable cv;
    int64_t encoded = -1, searched = -1;   // last group whose embeddings are ready / whose buffer is free again
    int64_t enc_failed_at = -1;            // the group the encoder failed on (none: -1)
    fsgpu_status enc_status = FSGPU_OK;
    std::string enc_detail;
    double enc_ms = 0, search_ms = 0;
    uint64_t fallbacks = 0;
    result->error_detail[0] = 0;
    auto keep_detail = [&](const std::string& d) {
        std::strncpy(result->error_detail, d.c_str(), sizeof(result->error_detail) - 1);
        result->error_detail[sizeof(result->error_detail) - 1] = 0;
    };

    auto texts_of = [&](uint32_t g) {
        const uint32_t b0 = g * group, b1 = std::min(n_batches, b0 + group);
        return std::pair<size_t, size_t>((size_t)b0 * batch, (size_t)b1 * batch);
    };
    // encoder e's slice of group g: texts [lo, hi) (data-parallel: an even split of the group's texts; else everything)
    auto slice_of = [&](uint32_t g, uint32_t e) {
        const auto [t0, t1] = texts_of(g);
        if (!data_parallel) return std::pair<size_t, size_t>(t0, t1);
        const size_t n = t1 - t0, per = (n + n_encoders - 1) / n_encoders;
        const size_t lo = std::min(n, (size_t)e * per), hi = std::min(n, lo + per);
        return std::pair<size_t, size_t>(t0 + lo, t0 + hi);
    };
    auto encode_slice = [&](uint32_t g, uint32_t e, std::vector<uint32_t>& offs, std::string* detail) -> fsgpu_status {
        const auto [s0, s1] = slice_of(g, e);
        const auto [t0, t1] = texts_of(g);
        (void)t1;
        // one embed call per encoder batch (the reference's embed_batch_sync is called with the caller's batch); a data-parallel
        // slice is one call of its own
        const size_t step = data_parallel ? std::max<size_t>(s1 - s0, 1) : batch;
        for (size_t b = s0; b < s1; b += step) {
            const uint32_t n = (uint32_t)std::min(step, s1 - b);
            offs.resize(n + 1);
            for (uint32_t i = 0; i <= n; ++i) offs[i] = offsets[b + i] - offsets[b];
            const fsgpu_status st = on_device ? fsgpu_bert_embed_device(encoders[e], ids + offsets[b], offs.data(), n, bufs[e].p[g & 1] + (b - s0) * dim)
                                              : fsgpu_bert_embed(encoders[e], ids + offsets[b], offs.data(), n, emb[g & 1].data() + (b - t0) * dim);
            if (st != FSGPU_OK) {
                *detail = fsgpu_last_error();   // thread-local: read on the thread that made the call
                return st;
            }
        }
        return FSGPU_OK;
    };
    auto encode_group = [&](uint32_t g, std::vector<uint32_t>& offs, std::string* detail) -> fsgpu_status {
        if (!data_parallel) return encode_slice(g, 0, offs, detail);
        // one host thread per encoder (fsgpu_bert_embed_device blocks its caller; the encoders run side by side on their devices)
        std::vector<fsgpu_status> st(n_encoders, FSGPU_OK);
        std::vector<std::string> det(n_encoders);
        std::vector<std::thread> helpers;
        for (uint32_t e = 1; e < n_encoders; ++e)
            helpers.emplace_back([&, e] {
                std::vector<uint32_t> o;
                st[e] = encode_slice(g, e, o, &det[e]);
            });
        st[0] = encode_slice(g, 0, offs, &det[0]);
        for (std::thread& t : helpers) t.join();
        for (uint32_t e = 0; e < n_encoders; ++e)
            if (st[e] != FSGPU_OK) {
                *detail = det[e];
                return st[e];
            }
        return FSGPU_OK;
    };
    auto search_group = [&](uint32_t g) -> fsgpu_status {
        const auto [t0, t1] = texts_of(g);
        const uint32_t nq = (uint32_t)(t1 - t0);
        uint32_t fb = 0;
        fsgpu_status st;
        if (index) {
            st = on_device ? fsgpu_search_topk_batched_device_queries(index, bufs[0].p[g & 1], nq, dim, k, rows.data(), scores.data(), counts.data(), &fb)
                           : fsgpu_search_topk_batched(index, emb[g & 1].data(), nq, dim, k, nullptr, rows.data(), scores.data(), counts.data(), &fb);
        } else if (data_parallel) {
            std::vector<const float*> parts;
            std::vector<uint32_t> part_counts;
            std::vector<int32_t> part_devs;
            for (uint32_t e = 0; e < n_encoders; ++e) {
                const auto [s0, s1] = slice_of(g, e);
                if (s1 == s0) continue;
                parts.push_back(bufs[e].p[g & 1]);
                part_counts.push_back((uint32_t)(s1 - s0));
                part_devs.push_back(bufs[e].dev);
            }
            fsgpu_sharded_request rq{nullptr, nq, dim, k, FSGPU_SHARDED_BATCHED, 0, nullptr, nullptr};
            st = fsgpu_sharded_search_parts(sharded, &rq, parts.data(), part_counts.data(), part_devs.data(), (uint32_t)parts.size(), rows.data(),
                                            scores.data(), counts.data(), &fb);
        } else {
            fsgpu_sharded_request rq{on_device ? nullptr : emb[g & 1].data(), nq, dim, k, FSGPU_SHARDED_BATCHED, 0, nullptr,
                                     on_device ? bufs[0].p[g & 1] : nullptr};
            st = fsgpu_sharded_search(sharded, &rq, rows.data(), scores.data(), counts.data(), &fb);
        }
        if (st != FSGPU_OK) {
            keep_detail(fsgpu_last_error());
            return st;
        }
        fallbacks += fb;
        if (out_rows) std::memcpy(out_rows + t0 * k, rows.data(), (size_t)nq * k * 4);
        if (out_scores) std::memcpy(out_scores + t0 * k, scores.data(), (size_t)nq * k * 4);
        if (out_counts) std::memcpy(out_counts + t0, counts.data(), (size_t)nq * 4);
        return FSGPU_OK;
    };

    const auto start = clock::now();
    fsgpu_status status = FSGPU_OK;
    if (!overlap) {
        for (uint32_t g = 0; g < n_groups && status == FSGPU_OK; ++g) {
            const auto a = clock::now();
            std::string detail;
            status = encode_group(g, local_offsets, &detail);
            if (status != FSGPU_OK) keep_detail(detail);
            const auto b = clock::now();
            if (status == FSGPU_OK) status = search_group(g);
            enc_ms += ms_between(a, b);
            search_ms += ms_between(b, clock::now());
        }
    } else {
        bool stop = false;
        std::thread encoder_thread([&] {
            std::vector<uint32_t> offs;
            for (uint32_t g = 0; g < n_groups; ++g) {
                {
                    std::unique_lock<std::mutex> lk(mu);   // buffer g & 1 was last used by group g - 2
                    cv.wait(lk, [&] { return stop || searched >= (int64_t)g - 2; });
                    if (stop) return;
                }
                const auto a = clock::now();
                std::string detail;
                const fsgpu_status st = encode_group(g, offs, &detail);
                const double d = ms_between(a, clock::now());
                std::lock_guard<std::mutex> lk(mu);
                enc_ms += d;
                if (st != FSGPU_OK) {
                    // the groups encoded before this one are still searched (encoded stays where it was); the searcher stops at g
                    enc_status = st;
                    enc_detail = detail;
                    enc_failed_at = (int64_t)g;
                    cv.notify_all();
                    return;
                }
                encoded = g;
                cv.notify_all();
            }
        });
        for (uint32_t g = 0; g < n_groups; ++g) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return encoded >= (int64_t)g || enc_failed_at >= 0; });
                if (encoded < (int64_t)g) {   // the encoder failed on this group (every earlier one has been searched)
                    status = enc_status;
                    keep_detail(enc_detail);
                    break;
                }
            }
            const auto a = clock::now();
            status = search_group(g);
            search_ms += ms_between(a, clock::now());
            {
                std::lock_guard<std::mutex> lk(mu);
                searched = g;
                if (status != FSGPU_OK) stop = true;
            }
            cv.notify_all();
            if (status != FSGPU_OK) break;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        encoder_thread.join();
    }
    const double wall = std::chrono::duration<double>(clock::now() - start).count();
    result->wall_seconds = wall;
    result->queries = (uint64_t)n_batches * batch;
    result->groups = n_groups;
    result->queries_per_sec = wall > 0 ? (double)result->queries / wall : 0.0;
    result->mean_encode_ms = n_groups ? enc_ms / n_groups : 0.0;
    result->mean_search_ms = n_groups ? search_ms / n_groups : 0.0;
    result->exact_fallbacks = fallbacks;
    result->device_resident_handoff = on_device ? 1 : 0;
    result->encoders = n_encoders;
    return status;
}

}  // namespace fshost
