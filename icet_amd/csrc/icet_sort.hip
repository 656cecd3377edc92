// icet_amd/csrc/icet_sort.hip -- stable key/value radix sorts used by the keyframe (scan-1) build.
//
// The reference sorts scan 1 by radial distance with std::sort(std::execution::par) on an index
// vector (/root/reference/src/icet.cpp:72-77); tie order is unspecified there and fixed here as
// "stable by original index", which an LSD radix sort gives for free.  A second stable sort by
// angular-bin id recovers the ascending-position order inside each bin that
// sortSphericalCoordinates' push_back produces (src/icet.cpp:534-554).
// Sorting is a plain library operation (rocPRIM through hipCUB); everything else on the path is
// hand-written in icet_kernels.hip.
#include <hipcub/hipcub.hpp>
#include "icet_internal.h"

namespace icet {

size_t sort_temp_bytes(int64_t total_n, int n_segments) {
    size_t a = 0, b = 0;
    uint32_t* k = nullptr; int32_t* off = nullptr;
    hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, a, k, k, k, k, (int)total_n, n_segments, off, off + 1, 0, 32, 0);
    hipcub::DeviceRadixSort::SortPairs(nullptr, b, k, k, k, k, (int)total_n, 0, 32, 0);
    return (a > b ? a : b) + 256;
}

hipError_t sort_pairs_segmented(void* tmp, size_t tmp_bytes, const uint32_t* key_in, uint32_t* key_out,
                                const uint32_t* val_in, uint32_t* val_out, int64_t total_n, int n_segments,
                                const int32_t* d_seg_off, int begin_bit, int end_bit, hipStream_t st) {
    if (total_n == 0) return hipSuccess;
    if (n_segments == 1)
        return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key_in, key_out, val_in, val_out, (int)total_n,
                                                  begin_bit, end_bit, st);
    return hipcub::DeviceSegmentedRadixSort::SortPairs(tmp, tmp_bytes, key_in, key_out, val_in, val_out, (int)total_n,
                                                       n_segments, d_seg_off, d_seg_off + 1, begin_bit, end_bit, st);
}

}  // namespace icet
