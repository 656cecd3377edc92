// icet_amd/csrc/icet_block_sort.h -- block-wide stable LSD radix pass over (key, payload) pairs, in LDS or on global scratch.
// Shared by the rank sort's per-bucket sort (icet_ranksort.hip) and by the ordering of oversized angular bins (icet_keyframe.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "icet_device_common.h"

namespace icet {

constexpr int kSortBlock = 256;
constexpr int kSortWaves = kSortBlock / 64;

// LDS layout (words): cells[C]  (the radix fallback keeps cnt[kSortWaves*256] | tot[256] | wsum[4] in the same words)
//                     | red[16] | buf0 keys[kCap] rows[kCap] | buf1 keys[kCap] rows[kCap]
constexpr int kOffCnt = 0, kOffTot = kSortWaves * 256, kOffWsum = kOffTot + 256, kRadixWords = kOffWsum + 4;
constexpr int kRedWords = 16;

// One stable 8-bit LSD pass.  kLds selects, at compile time, LDS arrays (indexed off the extern __shared__ base, so the
// compiler emits ds_* instructions) or the global scratch pointers; `sel` says which of the two buffers is the input.
template <bool kLds>
__device__ __forceinline__ void radix_pass(uint32_t* smem, int offBuf, int kCap, int sel, const uint2* gIn, uint2* gOut, int n, int shift, uint32_t dmask) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* cnt = smem + kOffCnt; uint32_t* tot = smem + kOffTot; int* wsum = reinterpret_cast<int*>(smem + kOffWsum);
    const int inB = offBuf + sel * 2 * kCap, outB = offBuf + (1 - sel) * 2 * kCap;
    auto ld = [&](int i) -> uint2 { if constexpr (kLds) return make_uint2(smem[inB + i], smem[inB + kCap + i]); else return gIn[i]; };
    auto st = [&](uint32_t dest, uint2 kv) {
        if constexpr (kLds) { smem[outB + dest] = kv.x; smem[outB + kCap + dest] = kv.y; } else gOut[dest] = kv;
    };
    int seg = (n + kSortWaves - 1) / kSortWaves; seg = (seg + 63) / 64 * 64;
    const int wlo = wave * seg, whi = min(n, wlo + seg);
    uint32_t* mine = cnt + wave * 256;
    for (int i = lane; i < 256; i += 64) mine[i] = 0u;
    // (same wave zeroes and then adds: LDS operations of one wave complete in order)
    for (int i = wlo + lane; i < whi; i += 64) atomicAdd(&mine[(ld(i).x >> shift) & 255u], 1u);
    __syncthreads();
    {                                                           // kSortBlock == 256: one thread per digit
        const int dgt = threadIdx.x;
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; w++) { const uint32_t c = cnt[w * 256 + dgt]; cnt[w * 256 + dgt] = t; t += c; }
        int incl = (int)t;
        incl = wave_incl_sum(incl);
        if (lane == 63) wsum[wave] = incl;
        tot[dgt] = (uint32_t)(incl - (int)t);                   // exclusive within the wave
    }
    __syncthreads();
    {
        int woff = 0;
        for (int k = 0; k < wave; k++) woff += wsum[k];
        tot[threadIdx.x] += (uint32_t)woff;
    }
    __syncthreads();
    for (int i = lane; i < 256; i += 64) mine[i] += tot[i];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int i0 = wlo; i0 < whi; i0 += 64) {
        const int i = i0 + lane;
        const bool ok = i < whi;
        const uint2 kv = ok ? ld(i) : make_uint2(0u, 0u);
        const uint32_t dgt = (kv.x >> shift) & 255u;
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (!((dmask >> q) & 1u)) continue;                 // block-uniform: every key of the bucket has the same bit here
            const bool bit = (dgt >> q) & 1u;
            const unsigned long long m = __ballot(ok && bit);
            peers &= bit ? m : ~m;
        }
        if (ok) {
            const int rank = __popcll(peers & lt);
            const uint32_t dest = mine[dgt] + (uint32_t)rank;
            if (rank == 0) mine[dgt] += (uint32_t)__popcll(peers);
            st(dest, kv);
        }
    }
    __syncthreads();
}

static_assert(kSortBlock == 256, "radix_pass assigns one thread per digit");

// Block-wide reductions of four words (OR, AND, min, max of the keys) through red[].
__device__ __forceinline__ void block_key_stats(uint32_t* red, uint32_t& vor, uint32_t& vand, uint32_t& vmin, uint32_t& vmax) {
    vor = wave_reduce_or(vor); vand = wave_reduce_and(vand); vmin = wave_reduce_min(vmin); vmax = wave_reduce_max(vmax);   // DPP, not the LDS crossbar
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave] = vor; red[kSortWaves + wave] = vand; red[2 * kSortWaves + wave] = vmin; red[3 * kSortWaves + wave] = vmax; }
    __syncthreads();
    vor = 0u; vand = 0xFFFFFFFFu; vmin = 0xFFFFFFFFu; vmax = 0u;
#pragma unroll
    for (int w = 0; w < kSortWaves; w++) { vor |= red[w]; vand &= red[kSortWaves + w]; vmin = min(vmin, red[2 * kSortWaves + w]); vmax = max(vmax, red[3 * kSortWaves + w]); }
    __syncthreads();
}
static_assert(4 * kSortWaves <= kRedWords, "red[] holds four words per wave");


// Stable sort of n (key, payload) pairs lying in global memory at gA, with gB as the second buffer; returns the buffer that holds
// the result (0 = gA, 1 = gB).  smem: kRadixWords + kRedWords words, red at smem + red_off.  Digits on which all keys agree are skipped.
__device__ __forceinline__ int radix_sort_global(uint32_t* smem, int red_off, uint2* gA, uint2* gB, int n) {
    uint32_t vor = 0u, vand = 0xFFFFFFFFu, vmin = 0xFFFFFFFFu, vmax = 0u;
    for (int i = threadIdx.x; i < n; i += kSortBlock) { const uint32_t k = gA[i].x; vor |= k; vand &= k; }
    block_key_stats(smem + red_off, vor, vand, vmin, vmax);
    const uint32_t differ = vor & ~vand;
    int sel = 0;
    for (int pass = 0; pass < 4; pass++) {
        if (((differ >> (8 * pass)) & 255u) == 0u) continue;    // block-uniform
        radix_pass<false>(smem, 0, 0, sel, sel ? gB : gA, sel ? gA : gB, n, 8 * pass, (differ >> (8 * pass)) & 255u);
        sel ^= 1;
    }
    return sel;
}

}  // namespace icet
