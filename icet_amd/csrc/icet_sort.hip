// icet_amd/csrc/icet_sort.hip -- the A/B alternative to the hand-written rank sort: rocPRIM (through hipCUB) radix sorts.
//
// Reached ONLY through icet_set_option("library_sort", 1) (off by default; kept to measure the rank sort of icet_ranksort.hip
// against a library sort).  The reference sorts scan 1 by radial distance with std::sort(std::execution::par) on an index
// vector (/root/reference/src/icet.cpp:72-77); tie order is unspecified there and fixed on this path as "stable by original
// index", which an LSD radix sort gives for free.
#include <hipcub/hipcub.hpp>
#include "icet_internal.h"

namespace icet {

// Device-wide (non-segmented) LSD radix sorts.  A batch of pairs is sorted in ONE pass structure by prefixing
// the pair id to the key: (pair << 32 | r bits) for the radial sort, (pair << vbits | bin) for the bin sort.
// rocPRIM's segmented sort gives each 116k-point segment to a single workgroup and was 5x slower here.
size_t sort_temp_bytes(int64_t total_n) {
    size_t a = 0, b = 0;
    uint32_t* k = nullptr; unsigned long long* k64 = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, k64, k64, k, k, (int)total_n, 0, 64, 0);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, k, k, k, k, (int)total_n, 0, 32, 0);
    return (a > b ? a : b) + 256;
}

hipError_t sort_pairs_u64(void* tmp, size_t tmp_bytes, const unsigned long long* key_in, unsigned long long* key_out,
                          const uint32_t* val_in, uint32_t* val_out, int64_t total_n, int end_bit, hipStream_t st) {
    if (total_n == 0) return hipSuccess;
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key_in, key_out, val_in, val_out, (int)total_n, 0, end_bit, st);
}

hipError_t sort_pairs_u32(void* tmp, size_t tmp_bytes, const uint32_t* key_in, uint32_t* key_out,
                          const uint32_t* val_in, uint32_t* val_out, int64_t total_n, int end_bit, hipStream_t st) {
    if (total_n == 0) return hipSuccess;
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key_in, key_out, val_in, val_out, (int)total_n, 0, end_bit, st);
}

}  // namespace icet
