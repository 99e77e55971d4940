// Minimal fork-join helper for the embarrassingly parallel per-bubble host phases (the role `-t`
// plays in the reference, src/CDBG.cpp:1723-1780 -- but results are always assembled in task
// order, so the output stays the `-t 1` one).
//
// Work runs on a persistent pool (threads are created once and grow to the largest `threads` ever asked for): a pass
// has a dozen parallel regions per batch and two pipeline stages issuing them concurrently, which is too many for
// thread creation per region.  Several regions may be in flight at once; idle workers join whichever has chunks left.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace pfh {

class WorkPool {
public:
    static WorkPool &instance() {
        static WorkPool p;
        return p;
    }

    // fn(chunk) for chunk = 0 .. n_chunks-1, on the caller plus at most threads-1 pool workers; returns when all ran
    void run(size_t n_chunks, unsigned threads, const std::function<void(size_t)> &fn) {
        if (n_chunks == 0) return;
        if (threads <= 1 || n_chunks == 1) {
            for (size_t c = 0; c < n_chunks; ++c) fn(c);
            return;
        }
        Job job;
        job.fn = &fn;
        job.n_chunks = n_chunks;
        job.slots = (int)std::min<size_t>(threads - 1, n_chunks - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            grow(threads - 1);
            jobs_.push_back(&job);
        }
        cv_.notify_all();
        work_on(job);
        std::unique_lock<std::mutex> lk(mu_);
        for (auto it = jobs_.begin(); it != jobs_.end(); ++it)
            if (*it == &job) { jobs_.erase(it); break; }  // no new worker can join from here on
        done_cv_.wait(lk, [&] { return job.active == 0 && job.completed.load(std::memory_order_acquire) == n_chunks; });
        if (job.failed) std::rethrow_exception(job.failed);
    }

private:
    struct Job {
        const std::function<void(size_t)> *fn = nullptr;
        size_t n_chunks = 0;
        std::atomic<size_t> next{0}, completed{0};
        int slots = 0;   // workers that may still join (guarded by mu_)
        int active = 0;  // workers inside work_on (guarded by mu_)
        std::exception_ptr failed;   // first exception thrown by a chunk (guarded by mu_)
    };

    WorkPool() = default;
    ~WorkPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }

    void grow(unsigned want) {  // mu_ held
        while (workers_.size() < want) workers_.emplace_back([this] { worker(); });
    }

    void work_on(Job &job) {
        for (;;) {
            const size_t c = job.next.fetch_add(1, std::memory_order_relaxed);
            if (c >= job.n_chunks) return;
            try {
                (*job.fn)(c);
            } catch (...) {   // rethrown on the thread that called run(): a worker must not take the process down
                std::lock_guard<std::mutex> lk(mu_);
                if (!job.failed) job.failed = std::current_exception();
            }
            job.completed.fetch_add(1, std::memory_order_release);
        }
    }

    void worker() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            Job *job = nullptr;
            for (Job *j : jobs_)
                if (j->slots > 0 && j->next.load(std::memory_order_relaxed) < j->n_chunks) { job = j; break; }
            if (!job) {
                if (quit_) return;
                cv_.wait(lk);
                continue;
            }
            job->slots--;
            job->active++;
            lk.unlock();
            work_on(*job);
            lk.lock();
            job->active--;
            job->slots++;
            done_cv_.notify_all();
        }
    }

    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::deque<Job *> jobs_;
    std::vector<std::thread> workers_;
    bool quit_ = false;
};

// fn(chunk_index, begin, end) for consecutive chunks of `chunk` items, dynamically scheduled
template <class F>
void parallel_chunks(size_t n, size_t chunk, unsigned threads, F &&fn) {
    if (n == 0) return;
    if (chunk == 0) chunk = 1;
    const size_t n_chunks = (n + chunk - 1) / chunk;
    const std::function<void(size_t)> body = [&](size_t c) { fn(c, c * chunk, std::min(n, (c + 1) * chunk)); };
    WorkPool::instance().run(n_chunks, threads, body);
}

inline size_t n_chunks_of(size_t n, size_t chunk) { return chunk ? (n + chunk - 1) / chunk : 0; }

}  // namespace pfh
