// Host-side threading pieces of the worker (no device code, no HIP): the libstdc++ RNG wrapper, the CPU budget, the spin-wait thread pool behind every
// parallel host section, and the background compressor of the Atari records' OBS tags.  A header of their own so that the sanitizer harness
// (tests/csrc/host_check.cpp: ThreadSanitizer / AddressSanitizer builds, `pytest -m "not gpu"`) compiles exactly the code the worker runs.
#pragma once
#include "common_host.h"
#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <limits>
#include <mutex>
#include <numeric>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace mz {

// ---- RNG (ref utils/random.h:9-41): libstdc++'s distributions over one mt19937 ----
struct Rng {
    std::mt19937 gen;
    std::uniform_int_distribution<int> int_dist;
    std::uniform_real_distribution<double> real_dist;
    void seed(int s) { gen.seed(s); }
    __attribute__((always_inline)) int randInt() { return int_dist(gen); }
    double randReal(double range = 1.0f) { return real_dist(gen) * range; }
    __attribute__((always_inline)) void dirichlet(float alpha, int size, std::vector<float>& out)
    {
        out.clear();
        std::gamma_distribution<float> gamma(alpha);
        for (int i = 0; i < size; ++i) { out.emplace_back(gamma(gen)); }
        float sum = std::accumulate(out.begin(), out.end(), 0.0f);
        if (sum < std::numeric_limits<float>::min()) { return; }
        for (int i = 0; i < size; ++i) { out[i] /= sum; }
    }
    void gumbel(int size, std::vector<float>& out)
    {
        out.clear();
        std::extreme_value_distribution<float> ev(0.0, 1.0);
        for (int i = 0; i < size; ++i) {
            float v = ev(gen);
            while (std::isinf(v)) { v = ev(gen); }
            out.emplace_back(v);
        }
    }
};

// ---- persistent thread pool: parallelFor over games ----
// Two short parallel sections per lock-step cycle (~1 ms apart), so wake-up latency matters more than
// anything else: workers spin on an epoch counter (pause) and only fall back to a condition variable
// after ~2 ms without work (worker stopped / between benchmarks).
// CPUs this process may run on, grouped by NUMA node (node of the calling thread first), so that a worker's threads
// share one memory domain with the pinned staging buffers they read and write.
inline std::vector<int> cpuOrder()
{
    cpu_set_t set;
    CPU_ZERO(&set);
    std::vector<int> allowed;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        for (int c = 0; c < CPU_SETSIZE; ++c) { if (CPU_ISSET(c, &set)) { allowed.push_back(c); } }
    }
    auto nodeOf = [](int cpu) {
        for (int node = 0; node < 64; ++node) {
            char path[128];
            snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpu%d", node, cpu);
            if (access(path, F_OK) == 0) { return node; }
        }
        return 0;
    };
    const int here = nodeOf(sched_getcpu());
    std::stable_sort(allowed.begin(), allowed.end(), [&](int a, int b) {
        const int na = nodeOf(a), nb = nodeOf(b);
        return (na != here) < (nb != here) || ((na != here) == (nb != here) && na < nb);
    });
    return allowed;
}

// CPUs this process can actually burn: the affinity mask, capped by the cgroup CPU bandwidth quota (cgroup v2 cpu.max, v1
// cpu.cfs_quota_us).  A spin-wait pool larger than the quota gets the whole container throttled for the rest of each 100-ms
// period (measured on the 1-GPU box: quota 16 CPUs, 32 spinners -> 40-50 ms stalls every ~100 cycles, 270k instead of 470k evals/s).
inline int usableCpus()
{
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = (sched_getaffinity(0, sizeof(set), &set) == 0) ? CPU_COUNT(&set) : static_cast<int>(std::thread::hardware_concurrency());
    if (n < 1) { n = 1; }
    double quota = -1, period = 100000;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) { quota = atof(q); }
        fclose(f);
    } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(f1, "%lf", &quota) != 1) { quota = -1; }
        fclose(f1);
        if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(f2, "%lf", &period) != 1) { period = 100000; }
            fclose(f2);
        }
    }
    if (quota > 0 && period > 0) { n = std::min(n, std::max(1, static_cast<int>(quota / period))); }
    return n;
}

class ThreadPool {
public:
    // cpu_base >= 0: pin the workers to consecutive entries of cpuOrder() starting at cpu_base + 1
    explicit ThreadPool(int n, int cpu_base = -1) : n_(std::max(1, n))
    {
        std::vector<int> cpus;
        if (cpu_base >= 0) { cpus = cpuOrder(); }
        auto pin = [&](pthread_t th, int idx) {
            if (cpus.empty()) { return; }
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(cpus[(cpu_base + idx) % cpus.size()], &one);
            (void)pthread_setaffinity_np(th, sizeof(one), &one);
        };
        // The calling thread is NOT pinned to one CPU: the HIP runtime's helper threads (signal / completion handlers,
        // created lazily) inherit the caller's mask, and a single-CPU mask would put them on the very core the caller
        // busy-waits on.  Only the pool's own spin-wait workers get exclusive CPUs.
        for (int t = 1; t < n_; ++t) {
            threads_.emplace_back([this]() { loop(); });
            pin(threads_.back().native_handle(), t);
        }
    }
    ~ThreadPool()
    {
        {
            std::lock_guard<std::mutex> l(mu_);
            quit_.store(true);
            state_.fetch_add(1ull << 32);
        }
        cv_.notify_all();
        for (auto& t : threads_) { t.join(); }
    }
    // Completion is counted in ITEMS, not in threads: a worker that the OS has descheduled (noisy neighbours on a shared
    // host) delays nothing but the one chunk it holds; workers that wake up late find the epoch gone and go back to waiting.
    void parallelFor(int count, const std::function<void(int)>& fn)
    {
        if (n_ == 1 || count < 2) { for (int i = 0; i < count; ++i) { fn(i); } return; }
        // Close the finished epoch BEFORE fn_ / count_ / chunk_ change.  A worker that the OS descheduled inside work() between its load of state_ = (E, count_old)
        // and its `b >= count_` test would otherwise test against the NEXT call's (larger) count, win the compare-exchange on the still unchanged state and run
        // items of an epoch that does not exist yet — through a function object that may be gone (found by round 5's fuzz sweep under eight-fold CPU
        // oversubscription: a segmentation fault in about one of 3 000 cases).  With the state moved first, that compare-exchange fails and the worker leaves.
        state_.store((state_.load(std::memory_order_relaxed) & 0xffffffff00000000ull) | 0x7fffffffull, std::memory_order_release);
        // (relaxed atomics, not plain fields: a late worker of the closed epoch may still be READING them in work() — its compare-exchange then fails, so the
        // values it saw never matter, but a plain read beside these writes is a data race all the same: ThreadSanitizer, tests/test_sanitizers.py, round 6)
        fn_.store(&fn, std::memory_order_relaxed);
        count_.store(count, std::memory_order_relaxed);
        chunk_.store(std::max(1, count / (n_ * 4)), std::memory_order_relaxed);
        processed_.store(0, std::memory_order_relaxed);
        const uint64_t epoch = (state_.load(std::memory_order_relaxed) >> 32) + 1;
        {
            std::lock_guard<std::mutex> l(mu_); // pairs with the sleepers' predicate check
            state_.store(epoch << 32, std::memory_order_release);
        }
        if (sleepers_.load(std::memory_order_acquire) > 0) { cv_.notify_all(); }
        work(epoch);
        while (processed_.load(std::memory_order_acquire) != count) { __builtin_ia32_pause(); }
    }

private:
    void work(uint64_t epoch)
    {
        while (true) {
            uint64_t s = state_.load(std::memory_order_acquire);
            if ((s >> 32) != epoch) { return; }
            const int b = static_cast<int>(s & 0xffffffffu);
            const int count = count_.load(std::memory_order_relaxed), chunk = chunk_.load(std::memory_order_relaxed);
            const std::function<void(int)>* fn = fn_.load(std::memory_order_relaxed);
            if (b >= count) { return; }
            // success: the state was still (epoch, b), so the epoch had not been closed when the three fields above were read — they are this epoch's
            if (!state_.compare_exchange_weak(s, s + static_cast<uint64_t>(chunk), std::memory_order_acq_rel)) { continue; }
            const int e = std::min(count, b + chunk);
            for (int i = b; i < e; ++i) { (*fn)(i); }
            processed_.fetch_add(e - b, std::memory_order_release);
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        while (true) {
            int spins = 0;
            while ((state_.load(std::memory_order_acquire) >> 32) == seen) {
                __builtin_ia32_pause();
                if (++spins > 60000) { // ~2 ms idle: sleep
                    std::unique_lock<std::mutex> l(mu_);
                    sleepers_.fetch_add(1);
                    cv_.wait(l, [&]() { return (state_.load(std::memory_order_acquire) >> 32) != seen; });
                    sleepers_.fetch_sub(1);
                    break;
                }
            }
            seen = state_.load(std::memory_order_acquire) >> 32;
            if (quit_.load()) { return; }
            work(seen);
        }
    }
    int n_;
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<const std::function<void(int)>*> fn_{nullptr};
    std::atomic<int> count_{0}, chunk_{1};
    std::atomic<uint64_t> state_{0}; // epoch << 32 | next item
    std::atomic<int> processed_{0}, sleepers_{0};
    std::atomic<bool> quit_{false};
};

// One queued stdout line.  An Atari record's OBS tag (gzip + hex of ~6 MB per 200-move sequence: 30 ms of one CPU) is filled in by background threads
// while the worker goes on playing; the line leaves the queue (mz_worker_pop_line) when it is complete, in the order the games finished.
struct OutLine {
    std::string text;
    std::atomic<int> pending{0}; // 1: the OBS placeholder is still to be replaced
    std::atomic<int> failed{0};
};

// Sleeping (not spinning) helper threads for those jobs: every game of a pool reaches its sequence boundary on the same move, i.e. 64 x 30 ms of
// compression are due at once; done between two moves on all host threads they were 1.1 ms per move averaged over a run with 15 threads and 8.8 ms with
// one (zero_num_threads=1, the budget of one rank of eight on a 16-CPU quota: C5 0.68 -> 0.24 M leaf-evals/s).  Here they overlap the following moves' kernels.
class ObsCompressor {
public:
    explicit ObsCompressor(int n)
    {
        for (int t = 0; t < std::max(1, n); ++t) { threads_.emplace_back([this]() { loop(); }); }
    }
    ~ObsCompressor()
    {
        { std::lock_guard<std::mutex> l(mu_); quit_ = true; jobs_.clear(); } // (lines nobody popped go with the worker)
        cv_.notify_all();
        for (auto& t : threads_) { t.join(); }
    }
    void submit(OutLine* line, std::string&& raw, const char* placeholder, size_t placeholder_len)
    {
        line->pending.store(1, std::memory_order_relaxed);
        {
            // back-pressure: a host that cannot compress as fast as the GPU plays (one helper thread against 64 sequences of 6 MB per 200 moves) must not
            // pile up raw observations without bound — the worker waits here, and is then exactly as fast as its compressor
            std::unique_lock<std::mutex> l(mu_);
            done_cv_.wait(l, [&]() { return jobs_.size() < kMaxQueued && (jobs_.empty() || queued_bytes_ + raw.size() <= kMaxQueuedBytes); });
            queued_bytes_ += raw.size();
            jobs_.push_back(Job{line, std::move(raw), placeholder, placeholder_len});
        }
        cv_.notify_one();
    }
    void wait(const OutLine* line) // until the line is complete
    {
        std::unique_lock<std::mutex> l(mu_);
        done_cv_.wait(l, [&]() { return line->pending.load(std::memory_order_acquire) == 0; });
    }

private:
    struct Job { OutLine* line; std::string raw; const char* ph; size_t ph_len; };
    static constexpr size_t kMaxQueued = 192, kMaxQueuedBytes = size_t(512) << 20; // raw observations waiting for a helper: by count and by bytes (a 200-move Atari sequence is 6 MB)
    size_t queued_bytes_ = 0;
    void loop()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [&]() { return quit_ || !jobs_.empty(); });
                if (quit_) { return; }
                j = std::move(jobs_.front());
                jobs_.pop_front();
                queued_bytes_ -= j.raw.size();
            }
            done_cv_.notify_all(); // (a submit() may be waiting for room)
            std::string hex;
            const bool ok = compressToHex(reinterpret_cast<const uint8_t*>(j.raw.data()), j.raw.size(), &hex);
            const size_t at = ok ? j.line->text.find(j.ph) : std::string::npos;
            if (at == std::string::npos) { j.line->failed.store(1); }
            else { j.line->text.replace(at, j.ph_len, hex); }
            { std::lock_guard<std::mutex> l(mu_); j.line->pending.store(0, std::memory_order_release); }
            done_cv_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::deque<Job> jobs_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    bool quit_ = false;
};

} // namespace mz
