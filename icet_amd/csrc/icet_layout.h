// icet_amd/csrc/icet_layout.h -- the slot order of a RAGGED throughput batch (icet_capi.hip solve_device_part).  Pure C++ (tests/cpp/test_layout.cpp).
// decode_block (icet_device_common.h) gives every block of the pair in slot s the XCD `s % 8`, so that a pair's tables stay in one L2.  With pairs of very different sizes in
// caller order the XCDs' shares differ (the reference's sample scans alternate 65 536 and 131 072 rows: all small pairs on the even XCDs, all large ones on the odd ones, and
// a step took what the odd XCDs took).  balanced_slot_order sorts the pairs by size and deals them to the slots of each group of eight in snake order -- every XCD the same
// share to a few per cent -- and then lets large and small pairs alternate down each XCD's column of slots (largest, smallest, second largest, ...): blocks are dispatched in
// slot order, and with all the large pairs' blocks first the point pass was 8 % slower than with sizes mixed.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace icet_layout {

// order[s] = the caller's pair that goes into slot s.  size[k] = rows of pair k (scan 1 + scan 2).
inline std::vector<int32_t> balanced_slot_order(const std::vector<int64_t>& size) {
    const int n = (int)size.size();
    std::vector<int32_t> by_size((size_t)n), order((size_t)n);
    for (int k = 0; k < n; k++) by_size[(size_t)k] = k;
    auto larger = [&](int32_t a, int32_t b) { return size[(size_t)a] > size[(size_t)b]; };
    std::stable_sort(by_size.begin(), by_size.end(), larger);
    for (int i = 0; i < n; i++) {
        const int g = i / 8, r = i % 8, in_group = std::min(8, n - 8 * g);
        const int pos = (g & 1) ? in_group - 1 - r : r;              // snake: the group's largest goes where the previous group put its smallest
        order[(size_t)(8 * g + pos)] = by_size[(size_t)i];
    }
    for (int x = 0; x < 8 && x < n; x++) {
        std::vector<int32_t> col;
        for (int s = x; s < n; s += 8) col.push_back(order[(size_t)s]);
        std::stable_sort(col.begin(), col.end(), larger);
        size_t lo = 0, hi = col.size();
        for (int s = x, t = 0; s < n; s += 8, t++) order[(size_t)s] = (t & 1) ? col[--hi] : col[lo++];
    }
    return order;
}

// Is the batch ragged enough for the layout to matter?  (largest pair more than a quarter above the smallest)
inline bool is_ragged(const std::vector<int64_t>& size) {
    if (size.empty()) return false;
    const auto mm = std::minmax_element(size.begin(), size.end());
    return *mm.second > *mm.first + *mm.first / 4;
}

}  // namespace icet_layout
