// plan_pool.h -- the planner's persistent helper threads (host only, no HIP): Engine::plan runs a step's object ranges as SHARES
// on them.  A header of its own so that tests/test_plan_pool.py can build it with -fsanitize=thread and hammer it.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

namespace pbso {

// Persistent helper threads of the planner: run(n, job) executes job(0) on the caller and job(1..n-1) on
// the workers and returns when all are done.
class PlanPool {
public:
    // pin_near >= 0: helper i is pinned to a core next to that cpu (same 8-core complex), so that the
    // queues the caller fills stay within one last-level cache (PBSO_PLAN_PIN=1)
    explicit PlanPool(int workers, int pin_near = -1) {
        for (int i = 0; i < workers; ++i) {
            th_.emplace_back([this, i] { loop(i + 1); });
            if (pin_near >= 0) {
                cpu_set_t set;
                CPU_ZERO(&set);
                CPU_SET((pin_near & ~7) | ((pin_near + i + 1) & 7), &set);
                (void)pthread_setaffinity_np(th_.back().native_handle(), sizeof(set), &set);
            }
        }
    }
    ~PlanPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    int workers() const { return (int)th_.size(); }
    // Shares are CLAIMED, not assigned: helper i goes for share i (the same objects every step: their queues stay in its cache),
    // the caller for share 0 and then for every share nobody has claimed yet -- so a helper that is slow to wake (the boxes of this
    // pool take 3 - 8 ms now and then to schedule a thread that slept on a condition variable: the outlier runs of the 128 x 512 x
    // 860 share, 2.0 - 2.6 ms per step instead of 1.3, scripts/debug/r05_stall_hunt.sh) costs its share's time on the caller, not
    // its wake-up.  Which thread runs a share does not matter: share t always works in context t.
    void run(int n, const std::function<void(int)> &job) {
        if (n > MAX_SHARES) n = MAX_SHARES;              // (callers pass at most their thread count: 16)
        unsigned long long gen;
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job;
            n_shares_ = n;
            gen = ++gen_;
            completed_.store(0, std::memory_order_release);
        }
        cv_.notify_all();
        if (take(0, gen)) { job(0); completed_.fetch_add(1, std::memory_order_release); }
        for (int t = 1; t < n; ++t)
            if (take(t, gen)) { job(t); completed_.fetch_add(1, std::memory_order_release); }
        // (shares a helper has claimed and not finished yet: short -- a share is a fraction of a millisecond)
        while (completed_.load(std::memory_order_acquire) < n) std::this_thread::yield();
    }

private:
    static constexpr int MAX_SHARES = 64;
    // The round's number goes into the share's word, and only a LARGER number takes it: first come, first served within a round,
    // and a helper that read an earlier round's (job, n, number) under the lock and was scheduled out before claiming finds every
    // word of that round at its number or beyond -- run() returns only when all n shares of its round are done -- and claims
    // nothing (with "!=" it could have run the earlier round's job, whose std::function is gone, beside the new round's).
    bool take(int t, unsigned long long gen) {
        unsigned long long seen = claim_[t].load(std::memory_order_acquire);
        while (seen < gen)
            if (claim_[t].compare_exchange_weak(seen, gen, std::memory_order_acq_rel)) return true;
        return false;
    }
    void loop(int idx) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int)> *job;
            int n;
            unsigned long long gen;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                job = job_;
                n = n_shares_;
                gen = gen_;
            }
#ifdef PBSO_PLAN_POOL_TEST_HOOK
            PBSO_PLAN_POOL_TEST_HOOK(idx);               // (tests/test_plan_pool.py: a helper scheduled out right here)
#endif
            // (the caller is inside run() for as long as a share of ITS round can still be taken: job and the contexts are alive)
            if (idx < n && take(idx, gen)) {
                (*job)(idx);
                completed_.fetch_add(1, std::memory_order_release);
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    const std::function<void(int)> *job_ = nullptr;
    std::atomic<unsigned long long> claim_[MAX_SHARES] = {};
    std::atomic<int> completed_{0};
    unsigned long long gen_ = 0;
    int n_shares_ = 0;
    bool stop_ = false;
};

}  // namespace pbso
