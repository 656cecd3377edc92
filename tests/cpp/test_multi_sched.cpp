// tests/cpp/test_multi_sched.cpp -- CPU test of the host-side scheduler icet_multi_* runs on (icet_amd/csrc/icet_multi_sched.h), with FAKE devices: N host threads, a
// gather that blocks until all N ranks have entered it (what an all-gather does), and failures injected on one rank.  No run on N > 1 GPUs has ever happened
// (DESIGN.md section 9); what can be shown without them is that the protocol cannot hang and that every rank's failure reaches the caller.
// Build: g++ -std=c++17 -O1 -pthread -I<repo> tests/cpp/test_multi_sched.cpp -o <out>;  exit code 0 and a last line "ok" = pass; a hang is turned into exit code 2.
#include "icet_amd/csrc/icet_multi_sched.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <future>

namespace {
struct Rendezvous {                       // the fake collective: call k completes on a rank when all n ranks have entered call k
    int n; std::mutex mu; std::condition_variable cv; std::vector<int> arrived;      // arrivals per call index
    explicit Rendezvous(int n_) : n(n_) {}
    void enter(int call) {
        std::unique_lock<std::mutex> lk(mu);
        if ((int)arrived.size() <= call) arrived.resize(call + 1, 0);
        arrived[call]++;
        cv.notify_all();
        cv.wait(lk, [&] { return arrived[call] >= n; });
    }
    int count(int call) { std::lock_guard<std::mutex> lk(mu); return call < (int)arrived.size() ? arrived[call] : 0; }
};
#define CHECK(x) do { if (!(x)) { std::printf("FAILED line %d: %s\n", __LINE__, #x); std::exit(1); } } while (0)

int scenario() {
    const int N = 8;
    icet_sched::Sched s;
    CHECK(s.start(N));
    Rendezvous rv(N);
    std::atomic<int> solved{0}, finished{0};
    auto enqueue = [&](int call, int fail_rank, bool by_exception) {
        return s.post_all(
            [&, call, fail_rank, by_exception](int d, std::string& why) -> int {
                std::this_thread::sleep_for(std::chrono::microseconds(200 * ((d * 7 + call) % 5)));      // ranks arrive out of step
                if (d == fail_rank) { if (by_exception) throw std::runtime_error("boom"); why = "injected failure on rank " + std::to_string(d); return 3; }
                solved++; return 0;
            },
            [&, call](int d, int solve_status, std::string&) -> int { (void)d; (void)solve_status; rv.enter(call); return 0; },      // ALWAYS entered
            [&](int, std::string&) -> int { finished++; return 0; });
    };
    int status = 0; std::string msg;
    auto drain = [](int, std::string&) -> int { return 0; };
    // 1. three calls queued behind each other without a sync in between; rank 5 fails in the second one BEFORE the collective
    CHECK(enqueue(0, -1, false)); CHECK(enqueue(1, 5, false)); CHECK(enqueue(2, -1, false));
    int bad = s.sync(drain, &status, &msg);
    CHECK(bad == 5 && status == 3 && msg == "injected failure on rank 5");
    CHECK(rv.count(0) == N && rv.count(1) == N && rv.count(2) == N);          // the failing rank entered the collective too: nobody waited for ever
    CHECK(solved == 3 * N - 1 && finished == 3 * N);
    // 2. the record is cleared by the sync: a clean call reports nothing
    CHECK(enqueue(3, -1, false));
    CHECK(s.sync(drain, &status, &msg) == -1 && rv.count(3) == N);
    // 3. a job that THROWS is a failure of its rank, not of the process, and still enters the collective
    CHECK(enqueue(4, 2, true));
    bad = s.sync(drain, &status, &msg);
    CHECK(bad == 2 && status == -1 && rv.count(4) == N);
    // 4. two ranks fail in one call: the FIRST (lowest) rank is reported, the other's record is cleared with it
    CHECK(s.post_all([&](int d, std::string& why) -> int { if (d == 6 || d == 1) { why = "r" + std::to_string(d); return 4; } return 0; },
                     [&](int, int, std::string&) -> int { rv.enter(5); return 0; }, [&](int, std::string&) -> int { return 0; }));
    bad = s.sync(drain, &status, &msg);
    CHECK(bad == 1 && status == 4 && msg == "r1" && rv.count(5) == N);
    CHECK(s.sync(drain, &status, &msg) == -1);
    // 5. phase 1 fails on rank 3: reported on the calling thread, NOTHING is queued (no rank is left waiting in a collective)
    int pst = 0;
    bad = s.prepare_all([](int d) -> int { return d == 3 ? 7 : 0; }, &pst);
    CHECK(bad == 3 && pst == 7 && rv.count(6) == 0);
    // 6. a failure found only when the device is drained (the sync's own check) is reported like any other
    CHECK(enqueue(6, -1, false));
    bad = s.sync([](int d, std::string& m) -> int { if (d == 4) { m = "drain failed"; return 9; } return 0; }, &status, &msg);
    CHECK(bad == 4 && status == 9 && msg == "drain failed");
    // 7. stop() with jobs still queued lets them run (they may be inside a collective) and joins
    CHECK(enqueue(7, -1, false));
    s.stop();
    CHECK(rv.count(7) == N);
    return 0;
}
}  // namespace

int main() {
    auto fut = std::async(std::launch::async, scenario);
    if (fut.wait_for(std::chrono::seconds(30)) != std::future_status::ready) { std::printf("HANG: the scheduler did not finish within 30 s\n"); std::fflush(stdout); std::_Exit(2); }
    const int rc = fut.get();
    if (rc == 0) std::printf("ok\n");
    return rc;
}
