// The second submitting thread of an engine (round 6; pbso_engine_desc::submit_thread).
//
// ModalSolver::step's bookkeeping for launch k + 1 (the planner, modal_solver.h:184-256) and the HIP calls that put launch k on
// the device are both host work; for small scenes -- 64 x 256 with a listener move per buffer: 0.045 ms of planning, 0.043 ms of
// uploads and launches, a 0.057 ms bank -- their sum, not the device, sets the step.  With this queue Engine::step_chunk RECORDS
// the calls of a launch (every argument evaluated then, on the caller's thread: nothing the next plan overwrites is read later)
// and a worker thread makes them, in order, while the caller returns and plans the next launch.
//
// What changes for the caller (hence an opt-in): when pbso_step returns, its launches are not necessarily in their streams yet.
// Every entry point of the C ABI that reads results or touches the engine's streams waits for the queue first; a caller that
// puts work of its OWN on the engine's stream behind a step calls pbso_flush in between.  A failing call surfaces at the next
// entry point that waits (the engine is then marked failed, as a failing step always did).
#pragma once
// (the host-only ThreadSanitizer stress test, tests/cpp/submit_queue_tsan.cpp, defines the two hooks itself: no HIP runtime there)
#ifndef PBSO_SQ_SET_DEVICE
#include <hip/hip_runtime.h>
#define PBSO_SQ_SET_DEVICE(d) (void)hipSetDevice(d)
#define PBSO_SQ_ERRSTR(rc) hipGetErrorString((hipError_t)(rc))
#endif

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

namespace pbso {

struct SubmitOp {
    const char *what;                  // the call, for the error text
    std::function<int()> run;          // returns a hipError_t / launch status as int, 0 = ok
};

// one recorded call: the function and its arguments BY VALUE (references decay to copies: a launch's IirParams is captured whole)
template <class F, class... A>
inline SubmitOp make_submit_op(const char *what, F fn, A &&...args) {
    return SubmitOp{what, [fn, tup = std::make_tuple(std::decay_t<A>(std::forward<A>(args))...)]() mutable -> int {
                        return (int)std::apply(fn, tup);
                    }};
}

class SubmitQueue {
public:
    explicit SubmitQueue(int device) : device_(device), th_([this] { loop(); }) {}
    ~SubmitQueue() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    SubmitQueue(const SubmitQueue &) = delete;
    SubmitQueue &operator=(const SubmitQueue &) = delete;

    // hands a launch's calls over; returns the batch's number (1, 2, ...)
    uint64_t push(std::vector<SubmitOp> &&ops) {
        uint64_t id;
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(std::move(ops));
            id = pushed_.fetch_add(1, std::memory_order_relaxed) + 1;
        }
        cv_.notify_one();
        return id;
    }
    uint64_t pushed() const { return pushed_.load(std::memory_order_relaxed); }
    uint64_t done() const { return done_.load(std::memory_order_acquire); }
    // until batch `id` has been made (all of its calls returned); a short spin first: the worker is usually a few calls behind
    void wait(uint64_t id) {
        for (int i = 0; i < 4000 && done() < id; ++i) std::this_thread::yield();
        if (done() >= id) return;
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return done_.load(std::memory_order_acquire) >= id; });
    }
    void drain() { wait(pushed()); }
    // the first failing call since the queue was made (0: none); the batches behind it were dropped
    int error(std::string *text) const {
        std::lock_guard<std::mutex> lk(m_);
        if (err_ && text) *text = err_text_;
        return err_;
    }

private:
    void loop() {
        PBSO_SQ_SET_DEVICE(device_);
        for (;;) {
            std::vector<SubmitOp> ops;
            {
                std::unique_lock<std::mutex> lk(m_);
                if (q_.empty() && !stop_) {
                    // a burst of steps keeps the worker awake: spin a little before sleeping (a sleeping thread of this pool's boxes
                    // now and then takes milliseconds to wake -- the planner pool's lesson of round 5)
                    lk.unlock();
                    for (int i = 0; i < 20000; ++i) {
                        if (pending_hint()) break;
                        std::this_thread::yield();
                    }
                    lk.lock();
                    cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                }
                if (q_.empty()) return;              // stop_
                ops = std::move(q_.front());
                q_.pop_front();
            }
            if (!err_flag_.load(std::memory_order_relaxed)) {
                for (SubmitOp &op : ops) {
                    const int rc = op.run();
                    if (rc != 0) {
                        std::lock_guard<std::mutex> lk(m_);
                        err_ = rc;
                        err_text_ = std::string(op.what) + ": " + PBSO_SQ_ERRSTR(rc);
                        err_flag_.store(true, std::memory_order_relaxed);
                        break;
                    }
                }
            }
            {
                std::lock_guard<std::mutex> lk(m_);
                done_.store(done_.load(std::memory_order_relaxed) + 1, std::memory_order_release);
            }
            cv_done_.notify_all();
        }
    }
    bool pending_hint() const { return pushed() > done(); }

    int device_;
    mutable std::mutex m_;
    std::condition_variable cv_, cv_done_;
    std::deque<std::vector<SubmitOp>> q_;
    std::atomic<uint64_t> pushed_{0};
    std::atomic<uint64_t> done_{0};
    std::atomic<bool> err_flag_{false};
    int err_ = 0;
    std::string err_text_;
    bool stop_ = false;
    std::thread th_;
};

}  // namespace pbso
