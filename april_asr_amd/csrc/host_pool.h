// A few helper threads for the per-session HOST copies of a step (lent PCM -> frame fifo, frame windows -> pinned
// staging, fifo compaction).  At a thousand sessions these copies are 5-10 MB per 100 ms feed and sit, single-threaded,
// in front of the first GPU launch of the step; they are independent per session.
// The stepping thread calls run() and takes part itself; small jobs run inline.
//
// Latency matters more than throughput here (a job is 20..500 us of copying, several per tick):
//  - completion is counted per ITEM, not per helper: run() returns when the last item is done, it never waits for a helper
//    that has not woken up yet (a condition-variable wake-up costs 50-100 us -- more than a 256-session job);
//  - helpers spin for a short while after a job (the next job of the same tick follows within microseconds) before they
//    block again.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace aprilx {

class HostPool {
public:
    explicit HostPool(int helpers, int spin_us = 300) : spin_us_(spin_us)
    {
        for (int i = 0; i < helpers; ++i) threads_.emplace_back([this] { worker(); });
    }
    ~HostPool()
    {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; job_.reset(); epoch_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    // fn(i) for i in [0, n); returns when all are done.  Items are handed out in blocks of `grain`.
    void run(size_t n, size_t grain, const std::function<void(size_t)> &fn)
    {
        if (threads_.empty() || n <= grain) { for (size_t i = 0; i < n; ++i) fn(i); return; }
        auto job = std::make_shared<Job>();
        job->fn = &fn; job->n = n; job->grain = grain;
        {
            std::lock_guard<std::mutex> g(mu_);
            job_ = job;
            epoch_.fetch_add(1, std::memory_order_release);
        }
        if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
        work(*job);
        // the blocks still running on helpers are short: spin (fn must stay valid until the last item is done)
        while (job->done.load(std::memory_order_acquire) < n) std::this_thread::yield();
    }

private:
    struct Job {
        const std::function<void(size_t)> *fn = nullptr;
        size_t n = 0, grain = 1;
        std::atomic<size_t> next{0}, done{0};
    };
    static void work(Job &j)
    {
        for (;;) {
            const size_t b = j.next.fetch_add(j.grain, std::memory_order_relaxed);
            if (b >= j.n) return;                              // (a helper that arrives late finds nothing and never touches fn)
            const size_t e = b + j.grain < j.n ? b + j.grain : j.n;
            for (size_t i = b; i < e; ++i) (*j.fn)(i);
            j.done.fetch_add(e - b, std::memory_order_release);
        }
    }
    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            // spin for a moment, then block
            const auto t0 = std::chrono::steady_clock::now();
            while (epoch_.load(std::memory_order_acquire) == seen) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) {
                    std::unique_lock<std::mutex> lk(mu_);
                    sleepers_.fetch_add(1, std::memory_order_release);
                    cv_.wait(lk, [&] { return epoch_.load(std::memory_order_acquire) != seen; });
                    sleepers_.fetch_sub(1, std::memory_order_release);
                    break;
                }
                std::this_thread::yield();
            }
            std::shared_ptr<Job> job;
            {
                std::lock_guard<std::mutex> g(mu_);
                seen = epoch_.load(std::memory_order_acquire);
                if (stop_) return;
                job = job_;
            }
            if (job) work(*job);
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::shared_ptr<Job> job_;
    std::atomic<uint64_t> epoch_{0};
    std::atomic<int> sleepers_{0};
    int spin_us_ = 300;
    bool stop_ = false;
};

}  // namespace aprilx
