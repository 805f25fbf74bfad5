// Host-only stress test of the engine's second submitting thread (openpbso_amd/csrc/submit_queue.h) under ThreadSanitizer:
//     g++ -std=c++17 -O1 -g -fsanitize=thread -I openpbso_amd/csrc tests/cpp/submit_queue_tsan.cpp -o sq_tsan -lpthread && ./sq_tsan
// The queue's two HIP hooks are defined here, so no HIP runtime is involved: the "calls" are functions that append to a log.
// What is checked: calls are made in the order they were recorded, across batches; arguments are captured BY VALUE at record time
// (the recorder overwrites its locals right after, as Engine::step_chunk's next plan does); wait(id) / drain() return only when the
// batch's calls have returned; a failing call keeps its text, drops the batches behind it and never blocks a waiter; the
// destructor makes what is still queued.  (tests/test_submit_queue.py builds and runs this.)
#include <cstdio>
#include <cstring>
static int g_device_set = -1;
#define PBSO_SQ_SET_DEVICE(d) (g_device_set = (d))
#define PBSO_SQ_ERRSTR(rc) ((rc) == 7 ? "seven" : "other")
#include "submit_queue.h"

#include <chrono>

using namespace pbso;

struct Params {          // stands for IirParams: captured whole
    int a[16];
    long long seq;
};
static std::vector<long long> g_log;          // written by the worker only; read by the test after wait()/drain()
static int record(const Params &p, int tag) {
    long long s = 0;
    for (int v : p.a) s += v;
    g_log.push_back(p.seq * 1000 + tag);
    return s == 16 * (int)(p.seq % 100) ? 0 : 99;       // (a torn / late-read capture shows as an error)
}
static int fails(int code) { return code; }
static int slow(int us) {
    std::this_thread::sleep_for(std::chrono::microseconds(us));
    return 0;
}

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    // 1. order, capture by value, wait / drain
    {
        SubmitQueue q(3);
        const int n_batches = 20000, per_batch = 5;
        Params p;
        uint64_t last = 0;
        for (int b = 0; b < n_batches; ++b) {
            std::vector<SubmitOp> ops;
            for (int k = 0; k < per_batch; ++k) {
                p.seq = b;
                for (int &v : p.a) v = b % 100;
                ops.push_back(make_submit_op("record", record, p, k));
                p.seq = -1;                             // what the recorder does next must not reach the call
                std::memset(p.a, 0x7f, sizeof(p.a));
            }
            last = q.push(std::move(ops));
            REQUIRE(last == (uint64_t)b + 1);
            if (b % 997 == 0) {
                q.wait(last);
                REQUIRE(q.done() >= last);
                REQUIRE(g_log.size() == (size_t)(b + 1) * per_batch);      // (safe to read: the worker is idle, done() was an acquire)
            }
        }
        q.drain();
        REQUIRE(q.done() == (uint64_t)n_batches && q.pushed() == (uint64_t)n_batches);
        REQUIRE(q.error(nullptr) == 0);
        REQUIRE(g_log.size() == (size_t)n_batches * per_batch);
        for (size_t i = 0; i < g_log.size(); ++i) REQUIRE(g_log[i] == (long long)(i / per_batch) * 1000 + (long long)(i % per_batch));
        REQUIRE(g_device_set == 3);
    }
    // 2. a failing call: text kept, later batches dropped (counted as done), waiters released
    {
        g_log.clear();
        SubmitQueue q(0);
        Params p;
        p.seq = 1;
        for (int &v : p.a) v = 1;
        std::vector<SubmitOp> a, b, c;
        a.push_back(make_submit_op("record", record, p, 0));
        b.push_back(make_submit_op("fails", fails, 7));
        b.push_back(make_submit_op("record", record, p, 1));          // behind the failure in its batch: not made
        c.push_back(make_submit_op("record", record, p, 2));          // a later batch: dropped
        q.push(std::move(a));
        q.push(std::move(b));
        const uint64_t id = q.push(std::move(c));
        q.wait(id);
        std::string why;
        REQUIRE(q.error(&why) == 7);
        REQUIRE(why == "fails: seven");
        REQUIRE(g_log.size() == 1 && g_log[0] == 1000);
        REQUIRE(q.done() == 3);
    }
    // 3. the destructor makes what is queued; a sleeping worker wakes for a late batch
    {
        g_log.clear();
        Params p;
        p.seq = 2;
        for (int &v : p.a) v = 2;
        {
            SubmitQueue q(0);
            std::this_thread::sleep_for(std::chrono::milliseconds(30));      // (past the worker's spin: it sleeps on the condition)
            std::vector<SubmitOp> a;
            a.push_back(make_submit_op("slow", slow, 2000));
            a.push_back(make_submit_op("record", record, p, 5));
            q.push(std::move(a));
            std::vector<SubmitOp> b;
            b.push_back(make_submit_op("record", record, p, 6));
            q.push(std::move(b));
        }
        REQUIRE(g_log.size() == 2 && g_log[0] == 2005 && g_log[1] == 2006);
    }
    std::printf("submit queue ok\n");
    return 0;
}
