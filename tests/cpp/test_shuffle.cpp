// tests/cpp/test_shuffle.cpp -- icet_amd/csrc/icet_shuffle.h against std::shuffle (g++ only, no GPU): the first m entries and the generator's state, over both of
// libstdc++'s branches, chained calls on ONE generator (the node's frames), and random (n, m).
#include "../../icet_amd/csrc/icet_shuffle.h"
#include <cstdio>

int main() {
    if (!icet_shuffle::matches_std_shuffle()) { std::printf("FAIL self-test\n"); return 1; }
    if (!icet_shuffle::fast_matches_std_shuffle()) { std::printf("FAIL self-test of the written-out generator\n"); return 1; }
    std::mt19937 a, b, pick(99); icet_shuffle::FastMt c;                                  // default seed, as the node's generator (simpleMapMaker.cpp:258)
    long checked = 0;
    for (int frame = 0; frame < 60; frame++) {
        const std::size_t n = (frame % 7 == 0) ? (std::size_t)(pick() % 70000) : (std::size_t)(100000 + pick() % 40000);
        const std::size_t m = (frame % 5 == 0) ? (std::size_t)(pick() % 3000) : (std::size_t)2000;
        std::vector<std::size_t> v(n); std::iota(v.begin(), v.end(), (std::size_t)0);
        std::shuffle(v.begin(), v.end(), a);
        std::vector<std::size_t> head;
        icet_shuffle::head_of_shuffled_iota(n, m, b, head);
        if (head.size() != std::min(m, n)) { std::printf("FAIL size frame %d\n", frame); return 1; }
        for (std::size_t k = 0; k < head.size(); k++) if (head[k] != v[k]) { std::printf("FAIL frame %d entry %zu (n %zu m %zu)\n", frame, k, n, m); return 1; }
        std::vector<std::size_t> head2;
        icet_shuffle::head_of_shuffled_iota(n, m, c, head2);
        if (head2 != head) { std::printf("FAIL written-out generator, frame %d (n %zu m %zu)\n", frame, n, m); return 1; }
        checked += (long)head.size();
    }
    { const auto xa = a(), xb = b(); const auto xc = c(); if (xa != xb || xa != xc) { std::printf("FAIL generator state\n"); return 1; } }
    std::printf("OK %ld entries\n", checked);
    return 0;
}
