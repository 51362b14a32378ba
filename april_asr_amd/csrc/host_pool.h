// A few helper threads for the per-session HOST copies of a step (lent PCM -> frame fifo, frame windows -> pinned
// staging, fifo compaction).  At a thousand sessions these copies are 5-10 MB per 100 ms feed and sit, single-threaded,
// in front of the first GPU launch of the step; they are independent per session.
// The stepping thread calls run() and takes part itself; small jobs run inline.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace aprilx {

class HostPool {
public:
    explicit HostPool(int helpers)
    {
        for (int i = 0; i < helpers; ++i) threads_.emplace_back([this] { worker(); });
    }
    ~HostPool()
    {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; ++epoch_; }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    // fn(i) for i in [0, n); returns when all are done.  Items are handed out in blocks of `grain`.
    void run(size_t n, size_t grain, const std::function<void(size_t)> &fn)
    {
        if (threads_.empty() || n <= grain) { for (size_t i = 0; i < n; ++i) fn(i); return; }
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn; n_ = n; grain_ = grain; next_.store(0); pending_ = (int)threads_.size(); ++epoch_;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void drain()
    {
        for (;;) {
            const size_t b = next_.fetch_add(grain_);
            if (b >= n_) return;
            const size_t e = b + grain_ < n_ ? b + grain_ : n_;
            for (size_t i = b; i < e; ++i) (*fn_)(i);
        }
    }
    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (stop_) return;
            }
            drain();
            { std::lock_guard<std::mutex> g(mu_); if (--pending_ == 0) done_.notify_one(); }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_ = 0, grain_ = 1;
    std::atomic<size_t> next_{0};
    int pending_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

}  // namespace aprilx
