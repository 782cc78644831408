// stage_pool.h -- a few helper threads for the host side of orbx_extract_batch: repacking 64 frames into the pinned staging
// buffer is ~20 MB of memcpy, which one core moves slower than PCIe gen5 does.  Workers keep polling for ~0.3 ms after a job
// (the chunks of one batch arrive back to back), then sleep on a condition variable.
//
// Every job is a self-contained, immutable record {fn, n} with its own counters, published through one shared pointer; a worker
// drains the job it LOADED, never "the current fields of the pool".  A worker that wakes up late for job A therefore either
// finds A exhausted (next >= n: it touches nothing else of A) or helps with whatever job it loads -- it cannot run job B's
// function with an index it took from job A, which the round-2 pool (plain fn_ / n_ / next_ members rewritten per job) allowed.
// Host-only C++ (no HIP): tests/test_sanitizer_cpu.py runs it under ThreadSanitizer with alternating job sizes.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <sched.h>

class StagePool {
public:
    explicit StagePool(int nthreads)
    {
        for (int i = 0; i < nthreads; i++) th_.emplace_back([this] { worker(); });
    }
    ~StagePool()
    {
        { std::lock_guard<std::mutex> lk(mu_); quit_.store(true); gen_.fetch_add(1); }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int threads() const { return (int)th_.size(); }
    // fn(i) for i in [0, n), spread over the workers and the calling thread; returns when all are done
    void parallel_for(int n, const std::function<void(int)> &fn)
    {
        if (n <= 0) return;
        if (th_.empty() || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
        std::shared_ptr<Job> job = std::make_shared<Job>(&fn, n);
        std::atomic_store_explicit(&cur_, job, std::memory_order_release);
        { std::lock_guard<std::mutex> lk(mu_); gen_.fetch_add(1); }
        cv_.notify_all();
        drain(*job);
        while (job->left.load(std::memory_order_acquire) > 0) std::this_thread::yield();
        // fn may die now: a worker that still holds `job` sees next >= n and returns without touching fn
    }
    // the CPUs this process may actually use: the affinity mask, cut down by the cgroup's CPU quota (hardware_concurrency()
    // reports the machine, which in a CPU-limited container makes yield-spinning workers fight the calling thread)
    static int usable_cpus()
    {
        int n = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = c; }
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {            // cgroup v2: "<quota> <period>" or "max <period>"
            char q[64]; long period = 0;
            if (fscanf(f, "%63s %ld", q, &period) == 2 && q[0] != 'm' && period > 0) {
                const long quota = atol(q);
                if (quota > 0) { const int c = (int)((quota + period - 1) / period); if (c > 0 && c < n) n = c; }
            }
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
            long quota = -1, period = 0;
            if (fscanf(g, "%ld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *p = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(p, "%ld", &period) != 1) period = 0; fclose(p); }
            if (quota > 0 && period > 0) { const int c = (int)((quota + period - 1) / period); if (c > 0 && c < n) n = c; }
        }
        return n > 0 ? n : 1;
    }
private:
    struct Job {
        const std::function<void(int)> *fn; const int n;
        std::atomic<int> next{0}, left;
        Job(const std::function<void(int)> *f, int n_) : fn(f), n(n_), left(n_) {}
    };
    static void drain(Job &j)
    {
        for (;;) {
            const int i = j.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j.n) return;
            (*j.fn)(i);
            j.left.fetch_sub(1, std::memory_order_release);
        }
    }
    void worker()
    {
        unsigned long long seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            while (gen_.load(std::memory_order_acquire) == seen) {        // poll first, sleep later
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return gen_.load() != seen; });
                    break;
                }
                std::this_thread::yield();
            }
            seen = gen_.load(std::memory_order_acquire);
            if (quit_.load()) return;
            std::shared_ptr<Job> job = std::atomic_load_explicit(&cur_, std::memory_order_acquire);
            if (job) drain(*job);
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<unsigned long long> gen_{0};
    std::shared_ptr<Job> cur_;                     // accessed with std::atomic_load / atomic_store only
    std::atomic<bool> quit_{false};
};
