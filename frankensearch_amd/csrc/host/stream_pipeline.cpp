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

fsgpu_status embed_search_stream(fsgpu_bert* encoder, fsgpu_index* index, fsgpu_sharded* sharded, const int32_t* ids,
                                 const uint32_t* offsets, uint32_t batch, uint32_t n_batches, uint32_t group, uint32_t k, bool overlap,
                                 bool host_handoff, uint32_t* out_rows, float* out_scores, uint32_t* out_counts, fshost_stream_result* result) {
    using clock = std::chrono::steady_clock;
    const uint32_t dim = index ? fsgpu_index_dimension(index) : fsgpu_sharded_dimension(sharded);
    const uint32_t n_groups = (n_batches + group - 1) / group;
    const size_t group_texts = (size_t)group * batch;
    // two embedding buffers: the encoder fills one while the search reads the other.  When the encoder and the index (a sharded
    // handle's root) share a device the vectors never leave HBM: fsgpu_bert_embed_device -> device queries of the search
    // (fsgpu_search_topk_batched_device_queries / fsgpu_sharded_request::queries_dev); `host_handoff` keeps them on the host path.
    const int32_t enc_dev = fsgpu_bert_device(encoder);
    const int32_t idx_dev = index ? fsgpu_index_device(index) : fsgpu_sharded_device(sharded, 0);
    float* emb_dev[2] = {nullptr, nullptr};
    bool on_device = !host_handoff && enc_dev >= 0 && enc_dev == idx_dev;
    if (on_device)
        for (int b = 0; b < 2 && on_device; ++b)
            if (fsgpu_device_malloc(enc_dev, (uint64_t)group_texts * dim * 4, reinterpret_cast<void**>(&emb_dev[b])) != FSGPU_OK) on_device = false;
    struct FreeDev {
        int32_t dev;
        float** p;
        ~FreeDev() {
            for (int b = 0; b < 2; ++b)
                if (p[b]) (void)fsgpu_device_free(dev, p[b]);
        }
    } free_dev{enc_dev, emb_dev};
    std::vector<float> emb[2];
    if (!on_device) emb[0].resize(group_texts * dim), emb[1].resize(group_texts * dim);
    std::vector<uint32_t> rows(group_texts * k), counts(group_texts), local_offsets;
    std::vector<float> scores(group_texts * k);
    std::mutex mu;
    std::condition_variable cv;
    int64_t encoded = -1, searched = -1;   // last group whose embeddings are ready / whose buffer is free again
    fsgpu_status enc_status = FSGPU_OK;
    std::string enc_detail;
    double enc_ms = 0, search_ms = 0;
    uint64_t fallbacks = 0;

    auto texts_of = [&](uint32_t g) {
        const uint32_t b0 = g * group, b1 = std::min(n_batches, b0 + group);
        return std::pair<size_t, size_t>((size_t)b0 * batch, (size_t)b1 * batch);
    };
    auto encode_group = [&](uint32_t g, std::vector<uint32_t>& offs) -> fsgpu_status {
        const auto [t0, t1] = texts_of(g);
        // one embed call per encoder batch (the reference's embed_batch_sync is called with the caller's batch)
        for (size_t b = t0; b < t1; b += batch) {
            offs.resize(batch + 1);
            for (uint32_t i = 0; i <= batch; ++i) offs[i] = offsets[b + i] - offsets[b];
            const fsgpu_status st = on_device
                                        ? fsgpu_bert_embed_device(encoder, ids + offsets[b], offs.data(), batch, emb_dev[g & 1] + (b - t0) * dim)
                                        : fsgpu_bert_embed(encoder, ids + offsets[b], offs.data(), batch, emb[g & 1].data() + (b - t0) * dim);
            if (st != FSGPU_OK) return st;
        }
        return FSGPU_OK;
    };
    auto search_group = [&](uint32_t g) -> fsgpu_status {
        const auto [t0, t1] = texts_of(g);
        const uint32_t nq = (uint32_t)(t1 - t0);
        uint32_t fb = 0;
        fsgpu_status st;
        if (index) {
            st = on_device ? fsgpu_search_topk_batched_device_queries(index, emb_dev[g & 1], nq, dim, k, rows.data(), scores.data(), counts.data(), &fb)
                           : fsgpu_search_topk_batched(index, emb[g & 1].data(), nq, dim, k, nullptr, rows.data(), scores.data(), counts.data(), &fb);
        } else {
            fsgpu_sharded_request rq{on_device ? nullptr : emb[g & 1].data(), nq, dim, k, FSGPU_SHARDED_BATCHED, 0, nullptr,
                                     on_device ? emb_dev[g & 1] : nullptr};
            st = fsgpu_sharded_search(sharded, &rq, rows.data(), scores.data(), counts.data(), &fb);
        }
        if (st != FSGPU_OK) return st;
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
            status = encode_group(g, local_offsets);
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
                const fsgpu_status st = encode_group(g, offs);
                const double d = ms_between(a, clock::now());
                std::lock_guard<std::mutex> lk(mu);
                enc_ms += d;
                if (st != FSGPU_OK) {
                    enc_status = st;
                    enc_detail = fsgpu_last_error();   // thread-local: read on the thread that made the call
                    encoded = (int64_t)n_groups;        // release the searcher
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
                cv.wait(lk, [&] { return encoded >= (int64_t)g; });
                if (enc_status != FSGPU_OK) {
                    status = enc_status;
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
    return status;
}

}  // namespace fshost
