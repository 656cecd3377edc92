// icet_amd/csrc/icet_multi_sched.h -- the host-side scheduler of icet_multi_* (icet_multi.hip), free of HIP so that the CPU suite can drive it
// with fake devices (tests/cpp/test_multi_sched.cpp; review r5, next 7: no run on N > 1 devices exists, so the protocol that must not hang is tested
// where it can be: N host threads, a rendezvous that blocks like an all-gather, a failure injected on one rank).
//
// One persistent host thread per device entry with a FIFO of jobs (hipSetDevice is per thread; a thread per call cost more than the solve), and the
// two-phase protocol of a call whose gather is a COLLECTIVE: a rank that skipped it would hang the others, so
//   (1) prepare_all: everything that can fail for reasons of the host before the collective (device selection, buffer growth) runs first on every
//       thread and is checked on the calling thread -- nothing has been queued yet when it fails;
//   (2) post_all: solve -> collective -> finish per rank; a rank whose solve FAILED still enters the collective with what it has, and its failure is
//       kept (first failure per rank since the last sync) for sync() to report;
//   sync: waits for every thread's queue, drains every device, returns the first failing rank and clears the record.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace icet_sched {

struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> jobs;
    bool quit = false, busy = false;
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return !jobs.empty() || quit; });
            if (jobs.empty() && quit) return;
            std::function<void()> j = std::move(jobs.front());
            jobs.pop_front(); busy = true;
            lk.unlock();
            j();                                   // jobs catch their own exceptions (nothing may escape a thread)
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
    void post(std::function<void()> j) {
        std::lock_guard<std::mutex> lk(mu);
        jobs.push_back(std::move(j));
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return jobs.empty() && !busy; });
    }
    void stop() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; cv.notify_all(); }
        if (th.joinable()) th.join();
    }
};

// Status codes are the caller's (0 = ok).
struct Sched {
    std::vector<Worker*> workers;
    std::vector<int> st; std::vector<std::string> why;      // first failure of every rank since the last sync
    std::mutex st_mu;
    bool pending = false;

    // n worker threads; false (and nothing left running) when a thread cannot be started
    bool start(int n) {
        try {
            st.assign(n, 0); why.assign(n, std::string());
            workers.reserve(n);
            for (int i = 0; i < n; i++) {
                Worker* w = new Worker();
                workers.push_back(w);                               // reserved above: cannot throw
                w->th = std::thread([w]() { w->loop(); });
            }
        } catch (...) { stop(); return false; }
        return true;
    }
    void stop() { for (Worker* w : workers) { w->stop(); delete w; } workers.clear(); }
    int size() const { return (int)workers.size(); }
    void wait_all() { for (Worker* w : workers) w->wait(); }

    // fn(d) on every rank's thread, then wait for all of them.  Throws what posting throws (std::bad_alloc) after waiting for the jobs already posted.
    template <typename F> void run_all(F fn) {
        const int D = size();
        try { for (int d = 0; d < D; d++) workers[d]->post([fn, d]() { fn(d); }); }
        catch (...) { wait_all(); throw; }
        wait_all();
    }
    // Phase 1.  prep(d) -> status, on every rank's thread; returns the first failing rank (its status in *status) or -1.  -2: cannot post.
    template <typename P> int prepare_all(P prep, int* status) {
        const int D = size();
        std::vector<int> s1;
        try { s1.assign(D, 0); int* p = s1.data(); run_all([=](int d) { p[d] = prep(d); }); } catch (...) { return -2; }
        for (int d = 0; d < D; d++) if (s1[d] != 0) { *status = s1[d]; return d; }
        return -1;
    }
    void record(int d, int status, const std::string& msg) {
        if (status == 0) return;
        std::lock_guard<std::mutex> lk(st_mu);
        if (st[d] == 0) { st[d] = status; try { why[d] = msg; } catch (...) {} }
    }
    // Phase 2, asynchronous: per rank  solve(d, msg) -> status;  collective(d, solve_status, msg) -> status, ALWAYS entered;  finish(d, msg) -> status.
    // false: cannot post (the jobs already posted still run and are collected by sync).
    template <typename S, typename C, typename F> bool post_all(S solve, C collective, F finish) {
        const int D = size();
        pending = true;
        try {
            for (int d = 0; d < D; d++)
                workers[d]->post([this, d, solve, collective, finish]() {
                    std::string msg, m2;
                    int s = 0;
                    try { s = solve(d, msg); } catch (...) { s = -1; msg = "exception in the solve job"; }
                    // (a failure above does NOT skip the collective: the other ranks have queued theirs and would wait for this one for ever)
                    int c = 0;
                    try { c = collective(d, s, m2); } catch (...) { c = -1; m2 = "exception in the gather job"; }
                    if (s == 0 && c != 0) { s = c; msg = m2; }
                    m2.clear();
                    int f = 0;
                    try { f = finish(d, m2); } catch (...) { f = -1; m2 = "exception behind the gather"; }
                    if (s == 0 && f != 0) { s = f; msg = m2; }
                    record(d, s, msg);
                });
        } catch (...) { return false; }
        return true;
    }
    // Waits for every thread, drains every rank (drain(d, msg) -> status; called only when something was posted since the last sync), returns the
    // first failing rank (-1: none) with its status and message, and clears the record.
    template <typename Dr> int sync(Dr drain, int* status, std::string* msg) {
        const int D = size();
        wait_all();
        int first = -1;
        for (int d = 0; d < D; d++) {
            std::string m;
            const int e = pending ? drain(d, m) : 0;
            std::lock_guard<std::mutex> lk(st_mu);
            if (e != 0 && st[d] == 0) { st[d] = e; why[d] = m; }
            if (st[d] != 0 && first < 0) { first = d; *status = st[d]; *msg = why[d]; }
            st[d] = 0; why[d].clear();
        }
        pending = false;
        return first;
    }
};

}  // namespace icet_sched
