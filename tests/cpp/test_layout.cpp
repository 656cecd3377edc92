// tests/cpp/test_layout.cpp -- icet_amd/csrc/icet_layout.h (g++ only): the slot order of a ragged batch is a permutation, every XCD (slot % 8) gets the same share of the
// rows to a few per cent -- where the caller's order gives some XCDs twice the others' -- and large and small pairs alternate down a column.
#include "../../icet_amd/csrc/icet_layout.h"
#include <cstdio>
#include <random>

static double worst_over_mean(const std::vector<int64_t>& size, const std::vector<int32_t>& order) {
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tot = 0;
    for (size_t s = 0; s < order.size(); s++) { sum[s % 8] += (double)size[(size_t)order[s]]; tot += (double)size[(size_t)order[s]]; }
    double worst = 0; for (double v : sum) worst = std::max(worst, v);
    return worst / (tot / 8);
}

int main() {
    std::mt19937 rng(7);
    for (int trial = 0; trial < 200; trial++) {
        const int n = 65 + (int)(rng() % 400);
        std::vector<int64_t> size((size_t)n);
        const int kind = trial % 4;
        for (int k = 0; k < n; k++)
            size[(size_t)k] = kind == 0 ? ((k % 2) ? 262144 : 131072)                       // the reference's sample scans, alternating
                            : kind == 1 ? (int64_t)(20000 + rng() % 250000)                   // anything
                            : kind == 2 ? ((k % 8 < 2) ? 300000 : 60000)                      // two heavy XCDs in caller order
                            : (int64_t)(100000 + (rng() % 3) * 70000);
        const std::vector<int32_t> order = icet_layout::balanced_slot_order(size);
        std::vector<int> seen((size_t)n, 0);
        for (int32_t k : order) { if (k < 0 || k >= n || seen[(size_t)k]++) { std::printf("FAIL not a permutation (trial %d)\n", trial); return 1; } }
        std::vector<int32_t> ident((size_t)n); for (int k = 0; k < n; k++) ident[(size_t)k] = k;
        const double w = worst_over_mean(size, order), w0 = worst_over_mean(size, ident);
        if (w > 1.06 + 8.0 * 300000 / (n * 100000.0)) { std::printf("FAIL imbalance %.3f (caller order %.3f) trial %d n %d kind %d\n", w, w0, trial, n, kind); return 1; }
        if ((kind == 0 || kind == 2) && n % 8 == 0 && w0 < 1.2) { std::printf("FAIL the test's own premise: caller order %.3f\n", w0); return 1; }
        // down a column: entries at even depth descend, at odd depth ascend (largest, smallest, second largest, ...)
        for (int x = 0; x < 8; x++) {
            int64_t prev_even = INT64_MAX, prev_odd = -1;
            for (int s = x, t = 0; s < n; s += 8, t++) {
                const int64_t z = size[(size_t)order[(size_t)s]];
                if (t & 1) { if (z < prev_odd) { std::printf("FAIL column order (odd) trial %d\n", trial); return 1; } prev_odd = z; }
                else { if (z > prev_even) { std::printf("FAIL column order (even) trial %d\n", trial); return 1; } prev_even = z; }
            }
        }
    }
    if (icet_layout::is_ragged({100, 100, 120}) || !icet_layout::is_ragged({100, 126}) || icet_layout::is_ragged({})) { std::printf("FAIL is_ragged\n"); return 1; }
    std::printf("OK\n");
    return 0;
}
