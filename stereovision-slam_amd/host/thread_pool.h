// thread_pool.h — minimal persistent pool with a blocking parallel_for, used for the
// per-stream host bookkeeping of the lockstep pipeline and for the per-problem BA
// structure building inside the library.  Streams / problems are independent, so the
// loop bodies never share mutable state.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <vector>

namespace svs {

class ThreadPool {
public:
    explicit ThreadPool(int nthreads) : n_(nthreads < 1 ? 1 : nthreads)
    {
        for (int i = 1; i < n_; ++i) workers_.emplace_back([this] { worker(); });
    }
    ~ThreadPool()
    {
        {
            std::unique_lock<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    int size() const { return n_; }

    // runs fn(i) for i in [0, n); returns when all are done.  The caller participates.
    void parallel_for(int n, const std::function<void(int)> &fn)
    {
        if (n <= 0) return;
        if (n_ == 1 || n == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
        {
            std::unique_lock<std::mutex> lk(m_);
            fn_ = &fn; total_ = n; next_.store(0); pending_ = n; ++epoch_;
        }
        cv_.notify_all();
        run_chunk();
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void run_chunk()
    {
        int done = 0;
        for (;;) {
            int i = next_.fetch_add(1);
            if (i >= total_) break;
            (*fn_)(i);
            ++done;
        }
        if (done) {
            std::unique_lock<std::mutex> lk(m_);
            pending_ -= done;
            if (pending_ == 0) done_cv_.notify_all();
        }
    }
    void worker()
    {
        (void)pthread_setname_np(pthread_self(), "svs-pool");
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
            }
            run_chunk();
        }
    }
    int n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)> *fn_ = nullptr;
    std::atomic<int> next_{ 0 };
    int total_ = 0, pending_ = 0;
    unsigned long epoch_ = 0;
    bool stop_ = false;
};

} // namespace svs
