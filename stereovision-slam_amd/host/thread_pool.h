// thread_pool.h — minimal persistent pool with a blocking parallel_for, used for the
// per-stream host bookkeeping of the lockstep pipeline and for the per-problem BA
// structure building inside the library.  Streams / problems are independent, so the
// loop bodies never share mutable state.
//
// Every parallel_for publishes ONE immutable job object {fn, total, next, checked_in}
// under the lock; a worker snapshots the shared_ptr together with the epoch, so an index
// it draws can only ever be compared with, and run under, the job it was drawn from — a
// worker that is still between its last fetch_add and the bounds test when the caller
// starts the next loop holds the OLD job and simply finds it exhausted.  The caller returns
// only after every item has finished (remaining == 0); the job object outlives the loop as
// long as any straggler still holds it.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
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
        std::shared_ptr<Job> job = std::make_shared<Job>(fn, n);
        {
            std::unique_lock<std::mutex> lk(m_);
            job_ = job; ++epoch_;
        }
        cv_.notify_all();
        run(*job);
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [&] { return job->remaining.load(std::memory_order_acquire) == 0; });
        if (job_ == job) job_.reset();      // stragglers keep their own reference
    }

private:
    struct Job {
        Job(const std::function<void(int)> &f, int n) : fn(f), total(n), remaining(n) {}
        const std::function<void(int)> fn;   // a copy: stays valid for stragglers of this epoch
        const int total;
        std::atomic<int> next{ 0 };
        std::atomic<int> remaining;
    };
    void run(Job &j)
    {
        int done = 0;
        for (;;) {
            const int i = j.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j.total) break;
            j.fn(i);
            ++done;
        }
        if (done && j.remaining.fetch_sub(done, std::memory_order_acq_rel) == done) {
            std::unique_lock<std::mutex> lk(m_);   // pairs with the waiter's predicate check
            done_cv_.notify_all();
        }
    }
    void worker()
    {
        (void)pthread_setname_np(pthread_self(), "svs-pool");
        unsigned long seen = 0;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
                job = job_;
            }
            if (job) run(*job);
        }
    }
    int n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    std::shared_ptr<Job> job_;
    unsigned long epoch_ = 0;
    bool stop_ = false;
};

} // namespace svs
