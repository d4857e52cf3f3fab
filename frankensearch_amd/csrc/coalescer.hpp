// coalescer.hpp — dynamic batching of concurrent single-item callers.
//
// The reference's seams are per-query calls made from many host threads at once (VectorIndex::search_top_k takes
// &self and is called concurrently, crates/frankensearch-index/src/search.rs:192; the MiniLM backends serialise
// callers on a mutex, crates/frankensearch-rerank/src/native_embedder.rs:40-50).  On the GPU one pass over the slab
// serves 128 queries for the price of one, so the library gathers the callers that are in flight at the same time
// into one batched launch: a waiting thread becomes the leader, waits until the batch is full, the oldest request
// has waited max_wait_us or arrivals have paused for max_wait_us / 8, runs the batch and wakes exactly the callers it
// served plus the next leader (every request sleeps on its own condition variable: with a thousand parked callers a
// shared one is a thundering herd).
// No extra threads are created.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <vector>

namespace fsgpu {

// Base of a request type: the coalescer's bookkeeping.
struct CoalescedRequest {
    bool done = false;
    bool exec_threw = false;  // exec() left through an exception: the request was NOT served (callers report an error)
    int64_t arrival_ns = 0;
    std::condition_variable cv;
};

template <class Req>  // Req derives from CoalescedRequest
class Coalescer {
  public:
    // max_batch == 0 disables coalescing (callers go straight to the unbatched path).
    void configure(uint32_t max_batch, uint32_t max_wait_us) {
        std::lock_guard<std::mutex> lock(mu_);
        max_batch_.store(max_batch, std::memory_order_relaxed);
        max_wait_us_ = max_wait_us;
    }
    bool enabled() const { return max_batch_.load(std::memory_order_relaxed) != 0; }  // read without the lock
    // whether a parked request satisfies pred (a caller that only wants to join an EXISTING compatible batch asks first)
    template <class Pred>
    bool any_pending(Pred&& pred) {
        std::lock_guard<std::mutex> lock(mu_);
        for (Req* r : pending_)
            if (pred(*r)) return true;
        return false;
    }
    void stats(uint64_t* batches, uint64_t* requests) {
        std::lock_guard<std::mutex> lock(mu_);
        *batches = batches_;
        *requests = requests_;
    }

    // Blocks until `r` has been served.  exec(batch) runs on the leader's thread without the coalescer lock and must
    // fill every request's result/status; compatible(a, b) says whether b may ride in a's batch.
    template <class Exec, class Compatible>
    void submit(Req* r, Exec&& exec, Compatible&& compatible) {
        using clock = std::chrono::steady_clock;
        std::unique_lock<std::mutex> lk(mu_);
        r->done = false;
        r->exec_threw = false;
        r->arrival_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now().time_since_epoch()).count();
        pending_.push_back(r);
        last_arrival_ns_ = r->arrival_ns;
        const uint32_t max_batch = max_batch_.load(std::memory_order_relaxed);
        if (leader_ && pending_.size() >= (max_batch ? max_batch : 1)) leader_->cv.notify_one();  // batch is full
        while (!r->done) {
            if (leader_) {
                r->cv.wait(lk);
                continue;
            }
            leader_ = r;
            const uint32_t mb = max_batch_.load(std::memory_order_relaxed);
            const size_t cap = mb ? mb : 1;
            // wait for the batch to fill, but never longer than max_wait_us past the oldest request's arrival — and stop
            // early once arrivals have paused for an eighth of that window: with a handful of callers the batch is
            // complete as soon as all of them are parked, and waiting out the window would only add latency
            const auto deadline = clock::time_point(std::chrono::nanoseconds(pending_.front()->arrival_ns)) +
                                  std::chrono::microseconds(max_wait_us_);
            const auto gap = std::chrono::nanoseconds((int64_t)max_wait_us_ * 1000 / 8);
            for (;;) {
                if (pending_.size() >= cap) break;
                const auto now = clock::now();
                const auto quiet = clock::time_point(std::chrono::nanoseconds(last_arrival_ns_)) + gap;
                if (now >= deadline || now >= quiet) break;
                r->cv.wait_until(lk, quiet < deadline ? quiet : deadline);
            }
            std::vector<Req*> batch;
            Req* head = pending_.front();
            for (auto it = pending_.begin(); it != pending_.end() && batch.size() < cap;) {
                if (*it == head || compatible(*head, **it)) {
                    batch.push_back(*it);
                    it = pending_.erase(it);
                } else {
                    ++it;
                }
            }
            ++batches_;
            requests_ += batch.size();
            lk.unlock();
            // exec must not strand the batch: whatever it throws, every member is released (marked unserved) and the
            // leadership is handed on — otherwise the members would sleep forever behind a leader that no longer exists
            bool threw = false;
            try {
                exec(batch);
            } catch (...) {
                threw = true;
            }
            lk.lock();
            leader_ = nullptr;
            for (Req* b : batch) {
                b->exec_threw = threw;
                b->done = true;
                if (b != r) b->cv.notify_one();
            }
            // hand leadership to the oldest request still waiting (a new arrival may get there first; either way a
            // runnable thread exists whenever requests are pending and nobody leads)
            if (!pending_.empty() && pending_.front() != r) pending_.front()->cv.notify_one();
        }
    }

  private:
    std::mutex mu_;
    std::deque<Req*> pending_;
    Req* leader_ = nullptr;
    std::atomic<uint32_t> max_batch_{0};
    uint32_t max_wait_us_ = 0;
    uint64_t batches_ = 0, requests_ = 0;
    int64_t last_arrival_ns_ = 0;
};

}  // namespace fsgpu
