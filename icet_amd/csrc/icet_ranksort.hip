// icet_amd/csrc/icet_ranksort.hip -- hand-written stable rank sort of scan 1 by radial distance, per pair.
//
// The reference sorts an index vector by r with std::sort(std::execution::par) (/root/reference/src/icet.cpp:72-77)
// and then walks it with its one-step swap loop (:78-83).  What the device needs from the sort is s[] (the row with
// rank i) and pred[] = s^-1 (the rank of every row), with ties broken by original index (the oracle's rule).
//
// A device-wide library radix sort has to carry the pair id in the key (40 significant bits -> 5 digit passes over
// 12 bytes per element, ~1.4 ms per 256 pairs) and still needs a separate inverse-permutation pass.  Per pair the
// data is small (116 k keys), so this file does a sample sort instead:
//   k_rs_splitters   one wave per pair: radii of ~2 k sampled rows straight from the Cartesian scan, sorted in registers
//                    (bitonic), <= 255 splitters published -- runs BEFORE k_scan1_spherical, which then
//   [k_scan1_spherical, icet_kernels.hip] also finds every row's bucket (branch-free binary search of the splitters)
//                    and its tile's bucket histogram, at no extra pass over r
//   [k_bin_scan]     (shared with the voxel multi-split) exclusive scan over (bucket, tile)
//   k_rs_scatter     stable multi-split of the rows into their buckets (match-any ranking, as k_bin_scatter)
//   k_rs_bucket_sort one block per (pair, bucket), the bucket's (key, row) pairs in LDS: a COUNTING sort on the bucket's
//                    own key range (a bucket is ~1/128 of a smooth distribution, so its keys are close to uniform
//                    between its min and max: ~0.5 rows per cell of (key - min) >> shift), then every row ranks
//                    itself inside its cell by (key, row) -- 7 barriers instead of the ~15 of an LSD radix sort --
//                    and s[] and pred[] are written: no inverse-permutation kernel.  A bucket whose keys pile up in
//                    one cell takes the LSD radix sort in LDS (stable 8-bit passes, constant digits skipped), one that
//                    does not fit LDS runs that radix sort on global scratch; a bucket of identical keys (the zero rows
//                    of a real scan: thousands of exact r = 0) is already in order because the multi-split is stable.
// Equal keys always fall into one bucket (bucket = number of splitters strictly below the key), a repeated key that
// reaches the sample becomes a bucket of its own, and every stage is stable, so ties end up ordered by row index.
// Traffic ~30 B per row instead of ~130 B.
#include <hip/hip_runtime.h>
#include "icet_internal.h"
#include "icet_device_common.h"
#include "icet_block_sort.h"
#include <algorithm>

namespace icet {
namespace {

#ifndef ICET_RS_SAMPLES
#define ICET_RS_SAMPLES 2048
#endif
constexpr int kSamples = ICET_RS_SAMPLES;          // sampled keys per pair (power of two, sorted by 1024 threads)
constexpr int kMaxBuckets = kRankSortMaxBuckets;   // power of two (icet_internal.h: the workspace is sized with it)
constexpr int kBucketBits = kRankSortBucketBits;
#ifndef ICET_RS_TARGET
#define ICET_RS_TARGET 906
#endif
// Rows per bucket aimed for: a 64-channel scan (~116 k rows) gets ~130 buckets of ~900 rows, a scan above 232 k rows the maximum of 256
// (round 4: 128 -> 256 with the target moved from 512 to 906, i.e. nothing changes for a 64-channel scan and the 485 k-row scans of configs[4]
// get twice the blocks at half the rows: k_rs_bucket_sort 37 -> 25 us of that pair's 0.51 -> 0.49 ms).  The LDS
// capacity of the per-bucket sort is chosen per launch from the largest scan (rank_sort_cap): measured on 256 such pairs,
// 1280 rows (25 KB, 6 blocks per CU) beats the earlier fixed 2560 (45 KB, 3 blocks) by 0.2 ms -- the sort is latency bound
// and wants the occupancy.  With 2048 samples for ~128 buckets the bucket sizes scatter by ~20 % around their mean (944 +- 180 rows on a
// 121 k-row scan), so ~3 % of them exceed 1280 rows and went to the global-scratch path; 1664 rows (30.7 KB with the 1024 cells: five
// blocks per CU) keeps all but ~0.1 % in LDS and is 23 us per 256-pair keyframe faster than 1280, 1408 / 1536: in between,
// 1792 ... 2048 (four blocks per CU): no better than 1280.
constexpr int kBucketTarget = ICET_RS_TARGET;
#ifndef ICET_RS_PER_BLOCK
#define ICET_RS_PER_BLOCK 2
#endif
constexpr int kRsPerBlockBatch = ICET_RS_PER_BLOCK;   // buckets per block of k_rs_bucket_sort in a throughput batch (a small batch: one, it has CUs to spare)
constexpr int kRsPrefetch = 5;                   // (key, row) pairs per thread fetched ahead: 5 x 256 = 1280 rows, more than 97 % of the buckets of a 64-channel scan
constexpr int kCapMin = 1664, kCapMax = 8960;  // 8960 rows + 4096 cells = 156 KB: one block per CU, still far better than global scratch

// ---- splitters -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ float radius_of(float x, float y, float z) {     // the r that k_scan1_spherical stores, bit for bit
    const float r = radius_raw(x, y, z);
    return (r != r) ? 1000.0f : r;
}

// One BLOCK of 8 waves per pair: the 2048 sampled keys live in registers, FOUR per thread (element e = 4 * tid + r), and go through
// the bitonic network: the 12 stages whose partner distance is 1 or 2 are register-to-register, the 39 with distance 4 .. 128 exchange
// inside a wave (wave_xor: DPP and gfx950's permlane swaps, no trip through the LDS crossbar), and only the 6 with distance 256 .. 1024 cross waves through LDS (two barriers each).  Round 2 kept
// all 2048 keys in ONE wave (32 per lane, no barrier at all): 2112 compare-exchanges and 672 shuffles per lane -- 17 us of a single
// pair's 22 us for this kernel, and the rest of the block idle; here a lane does 264 + 156.
static_assert(kSamples == 2048, "k_rs_splitters holds 4 keys per thread of a 512-thread block");
constexpr int kSplitLoadThreads = 512;
// (n1_dev: scan-1 row counts that only the device knows -- the descriptor holds an upper bound; this is the keyframe's first kernel, so it also
// does k_patch_counts' job for its pair: one launch less in front of a sequential caller's keyframe)
__global__ __launch_bounds__(kSplitLoadThreads) void k_rs_splitters(PairDesc* __restrict__ desc,
                                                     uint32_t* __restrict__ splitters, int32_t* __restrict__ n_buckets, const int32_t* __restrict__ n1_dev, int32_t* __restrict__ flags, int32_t* __restrict__ zero_rows, const PairDesc* __restrict__ h_desc, const int32_t* __restrict__ h_seg, int32_t* __restrict__ seg, int n_pairs) {
    __shared__ uint32_t sm[kSamples];
    const int pair = blockIdx.x, tid = threadIdx.x;
    if (tid < 4) zero_rows[4 * pair + tid] = 0;                 // the pair's exact-zero row counts (k_scan1_spherical adds, k_fit_cluster reads): cleared by the keyframe's first kernel
    if (tid == 0) flags[pair] = 0;                              // the pair's "bounded walk overflowed" flag (k_exec_flags / k_scramble_src set it): cleared here, the keyframe's first kernel, instead of by a memset node
    // (h_desc: the descriptors still sit in the pinned staging -- a small batch, whose first kernel this is: every thread reads its pair's record from there and one
    // of them puts it, and the pair's segment offset, where the later kernels look; a copy command in front of this kernel costs the stream 10 - 60 us)
    const PairDesc d = h_desc ? h_desc[pair] : desc[pair];
    const int n = n1_dev ? max(0, min(n1_dev[pair], d.n1)) : d.n1;
    if (h_desc && tid == 0) { PairDesc dn = d; dn.n1 = n; desc[pair] = dn; seg[pair] = h_seg[pair]; if (pair == 0) seg[n_pairs] = h_seg[n_pairs]; }
    else if (n1_dev && tid == 0) desc[pair].n1 = n;             // (every thread of this block took both values itself; later kernels read the descriptor)
    const int stride = max(1, (n + kSamples - 1) / kSamples);
    const int ns = n > 0 ? (n + stride - 1) / stride : 0;
    uint32_t x[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {                               // element e = 4 * tid + r: each thread loads its own four samples
        const int j = 4 * tid + r;
        // sample j comes from a pseudo-random place inside the j-th stride, not from its start: a lidar scan is periodic in its row index (64 rings x 1024 or 2048
        // azimuth steps), and every 32nd / 64th row of a column-major scan is ONE ring -- the reference's sample scans then got buckets of 2 - 7 k rows against a mean of 900
        const size_t row = (size_t)min(n - 1, j * stride + (int)((((uint32_t)j * 2654435761u) >> 12) % (uint32_t)stride));
        x[r] = (j < ns) ? __float_as_uint(radius_of(d.s1[row], d.s1[d.ld1 + row], d.s1[2 * (size_t)d.ld1 + row])) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int k = 2; k <= kSamples; k <<= 1) {                   // bitonic sort, ascending
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 4) {                                       // partner in another thread, same register
                const int tj = j >> 2;
                const bool up = ((4 * tid) & k) == 0;
                const bool low = (tid & tj) == 0;               // this thread holds the lower index of the pair
                uint32_t o[4];
                if (tj < 64) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {               // (tj is a compile-time constant once the loops are unrolled: the switch folds)
                        switch (tj) {
                            case 1: o[r] = wave_xor<1>(x[r]); break;
                            case 2: o[r] = wave_xor<2>(x[r]); break;
                            case 4: o[r] = wave_xor<4>(x[r]); break;
                            case 8: o[r] = wave_xor<8>(x[r]); break;
                            case 16: o[r] = wave_xor<16>(x[r]); break;
                            default: o[r] = wave_xor<32>(x[r]); break;
                        }
                    }
                } else {                                        // another wave: through LDS
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < 4; r++) sm[4 * tid + r] = x[r];
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = sm[4 * (tid ^ tj) + r];
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t lo = min(x[r], o[r]), hi = max(x[r], o[r]);
                    x[r] = (low == up) ? lo : hi;
                }
            } else {                                            // partner in this thread: registers r and r ^ j
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (r & j) continue;
                    // direction bit: from the register index when k <= 2 (compile time), from the thread otherwise
                    const bool up = (k < 4) ? ((r & k) == 0) : (((4 * tid) & k) == 0);
                    const uint32_t lo = min(x[r], x[r ^ j]), hi = max(x[r], x[r ^ j]);
                    x[r] = up ? lo : hi; x[r ^ j] = up ? hi : lo;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) sm[4 * tid + r] = x[r];
    __syncthreads();
    int nb = (n + kBucketTarget - 1) / kBucketTarget;
    nb = min(max(nb, 1), kMaxBuckets);
    if (n == 0) nb = 0;
    if (threadIdx.x == 0) n_buckets[pair] = nb;
    // splitters[j], j = 1..nb-1 ; slot 0 unused ; unused slots = +max so that the search never counts them
    for (int j = threadIdx.x; j < kMaxBuckets; j += kSplitLoadThreads)
        splitters[(size_t)pair * kMaxBuckets + j] = (j >= 1 && j < nb) ? sm[(int)(((long long)j * ns) / nb)] : 0xFFFFFFFFu;
}

// Stable multi-split of the rows of one tile into their buckets; wave w owns the w-th quarter of the tile.
// (The ballot form: what a device that failed the LDS-atomic order self-test gets, and the tests' cross-check; k_rs_scatter_staged below otherwise.)
constexpr int kScatterRounds = kKfMaxPtsPerThread;   // a tile is at most 4 waves x this many rounds x 64 positions
__global__ __launch_bounds__(kBlock) void k_rs_scatter(const PairDesc* __restrict__ desc, const float* __restrict__ r1, const uint8_t* __restrict__ bkt,
                                                       const uint32_t* __restrict__ tile_base, const int32_t* __restrict__ bucket_start,
                                                       uint2* __restrict__ bkv, int n_pairs, int chunks) {
    __shared__ uint32_t lb[4 * kMaxBuckets];
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;
    const PairDesc d = desc[pair];
    int cs_ = (d.n1 + chunks - 1) / chunks; cs_ = (cs_ + kBlock - 1) / kBlock * kBlock;
    const int lo_ = chunk * cs_, hi_ = min(d.n1, lo_ + cs_);
    if (lo_ >= hi_) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qs = cs_ / 4;
    const int wlo = lo_ + wave * qs, whi = min(hi_, wlo + qs);
    const int rounds = qs / 64;
    for (int j = threadIdx.x; j < 4 * kMaxBuckets; j += kBlock) lb[j] = 0u;
    __syncthreads();
    const size_t o = d.off1;
    uint32_t bb[kScatterRounds], key[kScatterRounds]; bool ok[kScatterRounds];
    uint32_t* mine = lb + wave * kMaxBuckets;
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++) {                       // all loads first (clamped, unconditional: see k_bin_scatter), then the atomics
        const int v = wlo + 64 * k + lane;
        ok[k] = (k < rounds) & (v < whi);
        const size_t vv = o + (size_t)(ok[k] ? v : lo_);
        bb[k] = bkt[vv]; key[k] = __float_as_uint(r1[vv]);
    }
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++) {
        if (!ok[k]) { bb[k] = 0u; key[k] = 0u; }
        if (ok[k]) atomicAdd(&mine[bb[k]], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kMaxBuckets; b += kBlock) {
        const uint32_t c0 = lb[b], c1 = lb[kMaxBuckets + b], c2 = lb[2 * kMaxBuckets + b];
        const uint32_t base = (uint32_t)bucket_start[(size_t)pair * (kMaxBuckets + 1) + b] + tile_base[((size_t)pair * chunks + chunk) * kMaxBuckets + b];
        lb[b] = base; lb[kMaxBuckets + b] = base + c0; lb[2 * kMaxBuckets + b] = base + c0 + c1; lb[3 * kMaxBuckets + b] = base + c0 + c1 + c2;
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++) {
        unsigned long long peers = __ballot(ok[k]);
#pragma unroll
        for (int q = 0; q < kBucketBits; q++) {
            const bool bit = (bb[k] >> q) & 1u;
            const unsigned long long m = __ballot(ok[k] && bit);
            peers &= bit ? m : ~m;
        }
        if (ok[k]) {
            const int rank = __popcll(peers & lt);
            const uint32_t dest = mine[bb[k]] + (uint32_t)rank;
            if (rank == 0) mine[bb[k]] += (uint32_t)__popcll(peers);
            bkv[o + dest] = make_uint2(key[k], (uint32_t)(wlo + 64 * k + lane));      // (key, row): ONE scattered 8-byte store
        }
    }
}

// The same multi-split with ranks from the values the LDS atomics hand back (see k_bin_scatter, icet_keyframe.hip) and the tile's rows first
// grouped by bucket in LDS: a tile's ~16 rows of one bucket then leave as ONE run of consecutive 8-byte stores by consecutive lanes instead of
// 16 scattered ones -- the scattered form is bound by the number of store requests, not by instructions (ranks from LDS atomics alone: 116 -> 112 us
// per 256 pairs; staged: 101; with all loads issued before the first atomic: 83).
static_assert(kMaxBuckets <= kBlock && kMaxBuckets % 64 == 0 && kMaxBuckets <= 256, "k_rs_scatter_staged: one thread per bucket scans them, ids travel in 8 bits");
// Bucket 0 of a pair whose first splitter is 0 holds exactly the rows with r == 0 -- the invalid returns of a real scan, 5 k - 24 k rows -- and a stable
// multi-split leaves them in row order, which IS their final order (ties by row index): their s[] / pred[] entries are written here, where every row
// knows its place, and k_rs_bucket_sort skips the bucket (it used to load the 24 k equal keys through one block's global-scratch path).
__device__ __forceinline__ bool zero_bucket_done(const uint32_t* __restrict__ splitters, int pair, const int32_t* __restrict__ n_buckets) {
    return n_buckets[pair] >= 2 && splitters[(size_t)pair * kMaxBuckets + 1] == 0u;
}
__global__ __launch_bounds__(kBlock) void k_rs_scatter_staged(const PairDesc* __restrict__ desc, const float* __restrict__ r1, const uint8_t* __restrict__ bkt,
                                                              const uint32_t* __restrict__ tile_base, const int32_t* __restrict__ bucket_start,
                                                              uint2* __restrict__ bkv, int n_pairs, int chunks,
                                                              const uint32_t* __restrict__ splitters, const int32_t* __restrict__ n_buckets, uint32_t* __restrict__ s_out, int32_t* __restrict__ pred_out) {
    __shared__ uint32_t lb[4 * kMaxBuckets];                         // per wave and bucket: rows counted, then the wave's first slot in the stage
    __shared__ int32_t gdelta[kMaxBuckets];                          // bucket: (global position - stage position) of its rows of this tile
    __shared__ uint32_t wtot[kMaxBuckets / 64];
    __shared__ uint2 stage[kBlock * kScatterRounds];                 // 16 KB: (key, row), grouped by bucket, the tile's order inside a bucket
    __shared__ uint8_t stage_b[kBlock * kScatterRounds];
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;
    const PairDesc d = desc[pair];
    int cs_ = (d.n1 + chunks - 1) / chunks; cs_ = (cs_ + kBlock - 1) / kBlock * kBlock;
    const int lo_ = chunk * cs_, hi_ = min(d.n1, lo_ + cs_);
    if (lo_ >= hi_) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qs = cs_ / 4;
    const int wlo = lo_ + wave * qs, whi = min(hi_, wlo + qs);
    const int rounds = qs / 64;
    for (int j = threadIdx.x; j < 4 * kMaxBuckets; j += kBlock) lb[j] = 0u;
    uint32_t gbase = 0u;                                             // (in flight during the counting step)
    if (threadIdx.x < kMaxBuckets) gbase = (uint32_t)bucket_start[(size_t)pair * (kMaxBuckets + 1) + threadIdx.x] + tile_base[((size_t)pair * chunks + chunk) * kMaxBuckets + threadIdx.x];
    __syncthreads();
    const size_t o = d.off1;
    uint32_t bb[kScatterRounds], key[kScatterRounds]; bool ok[kScatterRounds];
    uint32_t* mine = lb + wave * kMaxBuckets;
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++) {                       // all loads first (clamped, unconditional: see k_bin_scatter), then the atomics
        const int v = wlo + 64 * k + lane;
        ok[k] = (k < rounds) & (v < whi);
        const size_t vv = o + (size_t)(ok[k] ? v : lo_);
        bb[k] = bkt[vv]; key[k] = __float_as_uint(r1[vv]);
    }
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++) {
        if (!ok[k]) { bb[k] = 0u; key[k] = 0u; }
        if (ok[k]) bb[k] |= atomicAdd(&mine[bb[k]], 1u) << 8;       // earlier rows of this wave in the bucket (ids are 8 bits)
    }
    __syncthreads();
    uint32_t c0 = 0u, c1 = 0u, c2 = 0u, tot = 0u; int incl = 0;
    if (threadIdx.x < kMaxBuckets) {
        const int b = threadIdx.x;
        c0 = lb[b]; c1 = lb[kMaxBuckets + b]; c2 = lb[2 * kMaxBuckets + b]; tot = c0 + c1 + c2 + lb[3 * kMaxBuckets + b];
        incl = wave_incl_sum((int)tot);
        if (lane == 63) wtot[wave] = (uint32_t)incl;
    }
    __syncthreads();
    if (threadIdx.x < kMaxBuckets) {
        const int b = threadIdx.x;
        uint32_t ls = (uint32_t)incl - tot;
        for (int q = 0; q < wave; q++) ls += wtot[q];
        gdelta[b] = (int32_t)gbase - (int32_t)ls;
        lb[b] = ls; lb[kMaxBuckets + b] = ls + c0; lb[2 * kMaxBuckets + b] = ls + c0 + c1; lb[3 * kMaxBuckets + b] = ls + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++)
        if (ok[k]) {
            const uint32_t b = bb[k] & 255u, j = mine[b] + (bb[k] >> 8);
            stage[j] = make_uint2(key[k], (uint32_t)(wlo + 64 * k + lane));
            stage_b[j] = (uint8_t)b;
        }
    __syncthreads();
    const int nt = hi_ - lo_;
    const bool zero_done = zero_bucket_done(splitters, pair, n_buckets);
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++) {
        const int j = k * kBlock + (int)threadIdx.x;
        if (j < nt) {
            const int b = stage_b[j];
            const int dest = gdelta[b] + j;
            bkv[(int64_t)o + dest] = stage[j];
            if (zero_done && b == 0) { s_out[o + dest] = stage[j].y; pred_out[o + stage[j].y] = dest; }      // (bucket 0 starts at rank 0)
        }
    }
}

// ---- per-bucket sort ------------------------------------------------------------------------------------------------
// LDS layout (words): cells[C]  (the radix fallback keeps its counters, kRadixWords words, in the same place) | red[kRedWords]
//                     | buf0 keys[kCap] rows[kCap] | buf1 keys[kCap] rows[kCap]
// words reserved for the cells: the radix fallback keeps its counters (kRadixWords) in the same place
__host__ __device__ constexpr int cell_region(int C) { return C > kRadixWords ? C : (kRadixWords + 3) / 4 * 4; }

// Counting sort of the n (key, row) pairs in buf0 on the bucket's own key range; true = done (s / pred written), false = some cell
// is too crowded (block-uniform; nothing written): the caller falls back to the radix sort.
__device__ __forceinline__ bool counting_sort_lds(uint32_t* smem, int Clayout, int logCmax, int kCap, int n, uint32_t kmin, uint32_t kmax,
                                                  int lo, size_t off1, uint32_t* s_out, int32_t* pred_out, int max_cell) {
    uint32_t* cells = smem; uint32_t* red = smem + cell_region(Clayout);
    const int offBuf = cell_region(Clayout) + kRedWords;
    // cells for THIS bucket: a launch whose LDS is sized for large buckets (small batches: every bucket up to 8960 rows stays in LDS) still
    // sorts an ordinary bucket over 1024 cells (~0.5 - 2 rows per cell), not over the 4096 the largest one needs
    const int logC = (n <= 2 * kSortBlock * 4) ? min(logCmax, 10) : logCmax;
    const int C = 1 << logC;
    const uint32_t* K0 = smem + offBuf; const uint32_t* I0 = K0 + kCap;
    uint32_t* K1 = smem + offBuf + 2 * kCap; uint32_t* I1 = K1 + kCap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t range = kmax - kmin;                                   // > 0 here
    const int sh = max(0, 32 - __clz(range) - logC);                      // (range >> sh) < C
    for (int j = threadIdx.x; j < C; j += kSortBlock) cells[j] = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kSortBlock) atomicAdd(&cells[(K0[i] - kmin) >> sh], 1u);
    __syncthreads();
    {   // exclusive scan of the cell counts in place (thread t owns C / 256 consecutive cells) + the largest count
        const int per = C / kSortBlock, c0 = threadIdx.x * per;
        uint32_t sum = 0u, big = 0u;
        for (int j = 0; j < per; j++) { const uint32_t c = cells[c0 + j]; sum += c; big = max(big, c); }
        const int incl = wave_incl_sum((int)sum);
        big = wave_reduce_max(big);
        if (lane == 63) { red[wave] = (uint32_t)incl; red[kSortWaves + wave] = big; }
        __syncthreads();
        uint32_t base = (uint32_t)incl - sum;
        for (int k = 0; k < wave; k++) base += red[k];
        big = 0u;
#pragma unroll
        for (int w = 0; w < kSortWaves; w++) big = max(big, red[kSortWaves + w]);
        if (big > (uint32_t)max_cell) { __syncthreads(); return false; }          // block-uniform
        for (int j = 0; j < per; j++) { const uint32_t c = cells[c0 + j]; cells[c0 + j] = base; base += c; }
    }
    __syncthreads();
    // placement by cell, in arrival order inside a cell; afterwards cells[c] = END of cell c (its begin = cells[c - 1])
    for (int i = threadIdx.x; i < n; i += kSortBlock) {
        const uint32_t k = K0[i];
        const uint32_t pos = atomicAdd(&cells[(k - kmin) >> sh], 1u);
        K1[pos] = k; I1[pos] = I0[i];
    }
    __syncthreads();
    // every row ranks itself among the rows of its cell by (key, row): ties in row order, the oracle's rule
    for (int p = threadIdx.x; p < n; p += kSortBlock) {
        const uint32_t k = K1[p], id = I1[p];
        const uint32_t c = (k - kmin) >> sh;
        const uint32_t b = c ? cells[c - 1] : 0u, e = cells[c];
        uint32_t r = b;
        for (uint32_t j = b; j < e; j++) { const uint32_t kj = K1[j], ij = I1[j]; r += ((kj < k) | ((kj == k) & (ij < id))) ? 1u : 0u; }
        s_out[off1 + lo + r] = id;
        pred_out[off1 + id] = lo + (int)r;
    }
    return true;
}

// The same counting sort for a bucket that is ENTIRELY in the threads' registers (n <= kRsPrefetch x 256 rows: 97 % of the buckets of a
// 64-channel scan): the keys never go through the first LDS buffer, and every phase is straight-line code over the thread's <= 5 rows -- five
// independent LDS atomics, then five placements, then five rank walks -- instead of a loop whose every trip waits for its own LDS reads.
// The cells must have been zeroed before the barriers of block_key_stats.  Same cells, same ranks, same bits as counting_sort_lds.
__device__ __forceinline__ bool counting_sort_regs(uint32_t* smem, int Clayout, int logC, int kCap, int n, const uint2 (&pf)[kRsPrefetch], uint32_t kmin, uint32_t kmax,
                                                   int lo, size_t off1, uint32_t* s_out, int32_t* pred_out, int max_cell) {
    uint32_t* cells = smem; uint32_t* red = smem + cell_region(Clayout);
    const int offBuf = cell_region(Clayout) + kRedWords;
    const int C = 1 << logC;
    // key << 32 | row as ONE 64-bit word: "sorts before me" is a single 64-bit compare.  The first buffer is free on this path (offBuf is even:
    // 8-byte aligned; it holds 2 x kCap such words, n + 4 are used)
    unsigned long long* KI = reinterpret_cast<unsigned long long*>(smem + offBuf);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t range = kmax - kmin;                                   // > 0 here
    const int sh = max(0, 32 - __clz(range) - logC);                      // (range >> sh) < C
    uint32_t c[kRsPrefetch]; bool ok[kRsPrefetch];
#pragma unroll
    for (int k = 0; k < kRsPrefetch; k++) { ok[k] = (int)threadIdx.x + k * kSortBlock < n; c[k] = ok[k] ? (pf[k].x - kmin) >> sh : 0u; }
#pragma unroll
    for (int k = 0; k < kRsPrefetch; k++) if (ok[k]) atomicAdd(&cells[c[k]], 1u);
    __syncthreads();
    {   // exclusive scan of the cell counts in place (thread t owns C / 256 consecutive cells) + the largest count
        const int per = C / kSortBlock, c0 = threadIdx.x * per;
        uint32_t sum = 0u, big = 0u;
        for (int j = 0; j < per; j++) { const uint32_t x = cells[c0 + j]; sum += x; big = max(big, x); }
        const int incl = wave_incl_sum((int)sum);
        big = wave_reduce_max(big);
        if (lane == 63) { red[wave] = (uint32_t)incl; red[kSortWaves + wave] = big; }
        __syncthreads();
        uint32_t base = (uint32_t)incl - sum;
        for (int k = 0; k < wave; k++) base += red[k];
        big = 0u;
#pragma unroll
        for (int w = 0; w < kSortWaves; w++) big = max(big, red[kSortWaves + w]);
        if (big > (uint32_t)max_cell) { __syncthreads(); return false; }          // block-uniform
        for (int j = 0; j < per; j++) { const uint32_t x = cells[c0 + j]; cells[c0 + j] = base; base += x; }
    }
    __syncthreads();
    // placement by cell, in arrival order inside a cell; afterwards cells[c] = END of cell c (its begin = cells[c - 1])
#pragma unroll
    for (int k = 0; k < kRsPrefetch; k++)
        if (ok[k]) KI[atomicAdd(&cells[c[k]], 1u)] = ((unsigned long long)pf[k].x << 32) | pf[k].y;
    if (threadIdx.x < 4) KI[n + (int)threadIdx.x] = ~0ull;               // what the look-ahead of the last cell reads: sorts after everything
    __syncthreads();
    uint32_t b[kRsPrefetch], e[kRsPrefetch];
#pragma unroll
    for (int k = 0; k < kRsPrefetch; k++) { b[k] = (ok[k] & (c[k] != 0u)) ? cells[c[k] - (c[k] != 0u ? 1u : 0u)] : 0u; e[k] = ok[k] ? cells[c[k]] : 0u; }
#pragma unroll
    for (int k = 0; k < kRsPrefetch; k++) {
        if (!ok[k]) continue;
        const uint32_t id = pf[k].y;
        const unsigned long long me = ((unsigned long long)pf[k].x << 32) | id;
        uint32_t r = b[k];
        // the first kCellAhead members of the cell in ONE round of reads (a cell holds 1 - 3 rows; walked one member per trip, the wave makes as
        // many dependent LDS round trips as its fullest cell has rows), the rest of a crowded cell in a loop.  No "inside my cell" test: what
        // lies behind the cell belongs to later cells -- larger keys -- or is the padding, and never sorts before me.
        constexpr uint32_t kCellAhead = 4;
        unsigned long long m[kCellAhead];
#pragma unroll
        for (uint32_t t = 0; t < kCellAhead; t++) m[t] = KI[b[k] + t];
#pragma unroll
        for (uint32_t t = 0; t < kCellAhead; t++) r += (m[t] < me) ? 1u : 0u;
        for (uint32_t j = b[k] + kCellAhead; j < e[k]; j++) r += (KI[j] < me) ? 1u : 0u;
        s_out[off1 + lo + r] = id;
        pred_out[off1 + id] = lo + (int)r;
    }
    return true;
}

// pf: the bucket's first kRsPrefetch x 256 (key, row) pairs, already in registers (fetched while the block sorted its previous bucket)
template <bool kLds>
__device__ __forceinline__ void bucket_sort_body(uint32_t* smem, int C, int logC, int kCap, const uint2* bkv, uint2* gA, uint2* gB,
                                                 int n, int lo, size_t off1, uint32_t* s_out, int32_t* pred_out, int max_cell, const uint2 (&pf)[kRsPrefetch],
                                                 uint32_t key_lo, uint32_t key_hi) {      // key_lo <= key_hi: bounds of the bucket's keys known from the splitters
    const size_t o = off1 + lo;
    const int offBuf = cell_region(C) + kRedWords;
    uint32_t vor = 0u, vand = 0xFFFFFFFFu, vmin = 0xFFFFFFFFu, vmax = 0u;
    if (kLds && n <= kRsPrefetch * kSortBlock) {                              // block-uniform: the whole bucket is in registers
        const int logCr = min(logC, 10);
        for (int j = threadIdx.x; j < (1 << logCr); j += kSortBlock) smem[j] = 0u;     // the cells
        if (key_lo <= key_hi) {
            // an inner bucket: its keys lie in (splitter, next splitter] -- the counting sort only needs A range that holds them, so the
            // block-wide min / max (four wave reductions, two barriers) is left to the first and the last bucket of the pair
            vmin = key_lo; vmax = key_hi; vor = key_hi; vand = key_lo == key_hi ? key_hi : 0u;
            __syncthreads();
        } else {
#pragma unroll
            for (int k = 0; k < kRsPrefetch; k++)
                if ((int)threadIdx.x + k * kSortBlock < n) { const uint32_t x = pf[k].x; vor |= x; vand &= x; vmin = min(vmin, x); vmax = max(vmax, x); }
            block_key_stats(smem + cell_region(C), vor, vand, vmin, vmax);    // (its barriers also publish the zeroed cells)
        }
        if ((vor & ~vand) == 0u) {                                             // identical keys: already in order (the multi-split is stable)
#pragma unroll
            for (int k = 0; k < kRsPrefetch; k++) { const int i = (int)threadIdx.x + k * kSortBlock; if (i < n) { s_out[o + i] = pf[k].y; pred_out[off1 + pf[k].y] = lo + i; } }
            return;
        }
        if (counting_sort_regs(smem, C, logCr, kCap, n, pf, vmin, vmax, lo, off1, s_out, pred_out, max_cell)) return;
        vor = 0u; vand = 0xFFFFFFFFu; vmin = 0xFFFFFFFFu; vmax = 0u;            // a crowded cell: the general path below, from the start
    }
#pragma unroll
    for (int k = 0; k < kRsPrefetch; k++) {
        const int i = threadIdx.x + k * kSortBlock;
        if (i < n) {
            const uint2 kv = pf[k];
            if constexpr (kLds) { smem[offBuf + i] = kv.x; smem[offBuf + kCap + i] = kv.y; }
            vor |= kv.x; vand &= kv.x; vmin = min(vmin, kv.x); vmax = max(vmax, kv.x);
        }
    }
    for (int i = threadIdx.x + kRsPrefetch * kSortBlock; i < n; i += kSortBlock) {
        const uint2 kv = bkv[o + i];
        if constexpr (kLds) { smem[offBuf + i] = kv.x; smem[offBuf + kCap + i] = kv.y; }
        vor |= kv.x; vand &= kv.x; vmin = min(vmin, kv.x); vmax = max(vmax, kv.x);
    }
    block_key_stats(smem + cell_region(C), vor, vand, vmin, vmax);             // (its barriers also publish buf0)
    // a digit position where every key agrees needs no radix pass; a bucket of identical keys -- the zero rows of a real scan -- is
    // already in order (the multi-split is stable)
    const uint32_t differ = vor & ~vand;
    int sel = 0;
    if (differ != 0u) {
        if constexpr (kLds) { if (counting_sort_lds(smem, C, logC, kCap, n, vmin, vmax, lo, off1, s_out, pred_out, max_cell)) return; }
        for (int pass = 0; pass < 4; pass++) {
            if (((differ >> (8 * pass)) & 255u) == 0u) continue;    // block-uniform
            radix_pass<kLds>(smem, offBuf, kCap, sel, sel ? gB : gA, sel ? gA : gB, n, 8 * pass, (differ >> (8 * pass)) & 255u);
            sel ^= 1;
        }
    }
    for (int i = threadIdx.x; i < n; i += kSortBlock) {
        uint32_t row;
        if constexpr (kLds) row = smem[offBuf + sel * 2 * kCap + kCap + i]; else row = (sel ? gB : gA)[i].y;
        s_out[o + i] = row;
        pred_out[off1 + row] = lo + i;
    }
}

template <int kRsPerBlock>
__global__ __launch_bounds__(kSortBlock) void k_rs_bucket_sort(const PairDesc* __restrict__ desc, const int32_t* __restrict__ bucket_start,
                                                               const int32_t* __restrict__ n_buckets, uint2* __restrict__ bkv, uint2* __restrict__ alt,
                                                               uint32_t* __restrict__ s_out, int32_t* __restrict__ pred_out, int kCap, int logC, int max_cell, int n_pairs,
                                                               const uint32_t* __restrict__ splitters, int zero_in_scatter) {
    extern __shared__ uint32_t smem[];
    // all buckets of a pair on one XCD (decode_block): their scattered 4-byte writes of pred[] then complete whole cache lines in
    // ONE L2 instead of leaving partial lines in eight
    int pair, part;
    if (!decode_block(n_pairs, kMaxBuckets / kRsPerBlock, pair, part)) return;
    const int nb = n_buckets[pair];
    if (part >= nb) return;
    const size_t off1 = (size_t)desc[pair].off1;
    const int C = 1 << logC;
    // A block sorts kRsPerBlock buckets (part, part + stride, ...) and fetches the next bucket's pairs into registers before it sorts
    // the current one: the load of a bucket (its 7 KB arrive after two dependent reads) overlaps the previous bucket's sort instead
    // of standing in front of its own.
    constexpr int kStride = kMaxBuckets / kRsPerBlock;
    const int32_t* bs = bucket_start + (size_t)pair * (kMaxBuckets + 1);
    int los[kRsPerBlock], ns[kRsPerBlock];
    uint32_t klo[kRsPerBlock], khi[kRsPerBlock];
#pragma unroll
    for (int q = 0; q < kRsPerBlock; q++) {
        const int bkt = part + q * kStride;
        const bool ok = bkt < nb;
        los[q] = ok ? bs[bkt] : 0; ns[q] = ok ? bs[bkt + 1] - los[q] : 0;
        if (zero_in_scatter && bkt == 0 && zero_bucket_done(splitters, pair, n_buckets)) ns[q] = 0;      // the bucket of exact zeros was finished by the multi-split
        // bucket = number of splitters strictly below the key: an inner bucket holds splitter[bkt] < key <= splitter[bkt + 1]
        const bool inner = ok && bkt >= 1 && bkt + 1 < nb;
        const uint32_t a = inner ? splitters[(size_t)pair * kMaxBuckets + bkt] : 0xFFFFFFFFu, b2 = inner ? splitters[(size_t)pair * kMaxBuckets + bkt + 1] : 0u;
        klo[q] = (inner && a < b2) ? a + 1u : 1u; khi[q] = (inner && a < b2) ? b2 : 0u;           // (1, 0): not known
    }
    auto fetch = [&](int lo, int n, uint2 (&pf)[kRsPrefetch]) {
#pragma unroll
        for (int k = 0; k < kRsPrefetch; k++) { const int i = threadIdx.x + k * kSortBlock; pf[k] = (i < n) ? bkv[off1 + lo + i] : make_uint2(0u, 0u); }
    };
    uint2 pf[kRsPrefetch];
    fetch(los[0], ns[0], pf);
#pragma unroll
    for (int q = 0; q < kRsPerBlock; q++) {
        uint2 nf[kRsPrefetch];
        if (q + 1 < kRsPerBlock) fetch(los[q + 1], ns[q + 1], nf);
        const int lo = los[q], n = ns[q];
        if (n > 0) {                                                   // block-uniform
            if (n <= kCap) bucket_sort_body<true>(smem, C, logC, kCap, bkv, nullptr, nullptr, n, lo, off1, s_out, pred_out, max_cell, pf, klo[q], khi[q]);
            else bucket_sort_body<false>(smem, C, logC, kCap, bkv, bkv + off1 + lo, alt + off1 + lo, n, lo, off1, s_out, pred_out, max_cell, pf, 1u, 0u);
            __syncthreads();                                           // the LDS buffers are reused by the next bucket
        }
        if (q + 1 < kRsPerBlock) {
#pragma unroll
            for (int k = 0; k < kRsPrefetch; k++) pf[k] = nf[k];
        }
    }
}

}  // namespace

#define ICET_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

// LDS rows of the per-bucket sort for scans of at most max_n rows: 1.3 x the mean bucket, rounded up to 128
static int rank_sort_cap(int max_n, int forced, int n_pairs) {
    int nb = (max_n + kBucketTarget - 1) / kBucketTarget; nb = nb < 1 ? 1 : (nb > kMaxBuckets ? kMaxBuckets : nb);
    int cap = (int)(1.3 * (double)max_n / nb); cap = (cap + 127) / 128 * 128;
    cap = cap < kCapMin ? kCapMin : (cap > kCapMax ? kCapMax : cap);
    // A small batch has a CU per block anyway: every bucket up to 8960 rows sorts in LDS.  Real scans need it: their thousands of exact-zero
    // rows take ~10 of the splitters, the other buckets grow to 3 - 12x the mean (frame_804: 6157 rows against a mean of 512) and
    // went to the global-scratch radix sort (65 us for this kernel against 16 on a synthetic pair).
    if (n_pairs <= 4) cap = kCapMax;
    if (forced > 0) cap = forced < 64 ? 64 : (forced > kCapMax ? kCapMax : forced);     // Tuning::rs_cap (tests: force the global-scratch path)
    return cap;
}
// cells of the counting sort: 1024 for the ~900-row buckets of a 64-channel scan (1 - 2 rows per cell), 4096 for large ones.  Measured
// on 256 pairs: 2048 cells 320 us, 1024 cells 279 us (half the scan work and 26 instead of 29 KB: six blocks per CU), 512 cells 291 us;
// 1024 cells with a u16 order array instead of the second (key, row) buffer (18 KB, eight blocks per CU) 288 us -- the extra
// indirection in the rank loop costs more than the occupancy brings.
#ifndef ICET_RS_LOGC_SMALL
#define ICET_RS_LOGC_SMALL 10
#endif
#ifndef ICET_RS_SMALL_CAP
#define ICET_RS_SMALL_CAP 1664
#endif
static int rank_sort_log_cells(int cap) { return cap <= ICET_RS_SMALL_CAP ? ICET_RS_LOGC_SMALL : 12; }
static size_t rank_sort_lds_bytes(int cap) { return (size_t)(cell_region(1 << rank_sort_log_cells(cap)) + kRedWords + 4 * cap) * 4; }

// Outputs: w.valB = s (row with rank i), w.pred = rank of every row.  Scratch: w.key64A (bucket-grouped (key, row) pairs),
// w.key64B (second buffer of an overflow bucket), w.bkt, w.counts / w.tile_base, w.splitters, w.n_buckets, w.bucket_start.
hipError_t init_rank_sort_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rs_bucket_sort<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rank_sort_lds_bytes(kCapMax));
    if (e == hipSuccess && kRsPerBlockBatch != 1) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rs_bucket_sort<kRsPerBlockBatch>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rank_sort_lds_bytes(kCapMax));
    return e;
}

hipError_t launch_rank_sort_splitters(const Workspace& w, const LaunchCfg& c, hipStream_t st, const int32_t* d_n1) {
    k_rs_splitters<<<c.n_pairs, kSplitLoadThreads, 0, st>>>(w.desc, w.splitters, w.n_buckets, d_n1, w.flags, w.zero_rows, c.h_desc_up, c.h_seg_up, w.seg_off, c.n_pairs);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_rank_sort(const Workspace& w, const LaunchCfg& c, hipStream_t st) {
    const int chunks = c.kf_chunks, np = c.n_pairs;
    const int groups = grid_groups(np);
    dim3 grid(groups * chunks), blk(kBlock);
    // (buckets bkt[] and the per-tile histograms counts[] were produced by k_scan1_spherical)
    // (its per-pair scan block also reduces the tiles' voxel ranges, written by k_scan1_spherical, to the pair's)
    hipError_t e = launch_class_scan(w.counts, w.tile_base, w.bucket_start, kMaxBuckets, chunks, np, st, nullptr, nullptr, 0, nullptr, nullptr, w.tile_vr, w.vrange);
    if (e != hipSuccess) return e;
    if (c.lds_rank) k_rs_scatter_staged<<<grid, blk, 0, st>>>(w.desc, w.r1, w.bkt, w.tile_base, w.bucket_start, reinterpret_cast<uint2*>(w.key64A), np, chunks, w.splitters, w.n_buckets, w.valB, w.pred);
    else k_rs_scatter<<<grid, blk, 0, st>>>(w.desc, w.r1, w.bkt, w.tile_base, w.bucket_start, reinterpret_cast<uint2*>(w.key64A), np, chunks);
    ICET_LAUNCH_CHECK();
    const int cap = rank_sort_cap(c.max_n1, c.rs_cap, c.n_pairs);
    const int zero_in_scatter = c.lds_rank ? 1 : 0;                       // (the ballot form of the multi-split does not write the zero bucket's ranks)
    // two buckets per block (the second one's pairs in flight during the first one's sort) once the launch fills the chip several times over;
    // a small batch -- one pair is at most 256 blocks on 256 CUs -- keeps a block per bucket
    if (groups * kMaxBuckets >= 16 * 256 && kRsPerBlockBatch != 1)
        k_rs_bucket_sort<kRsPerBlockBatch><<<dim3(groups * (kMaxBuckets / kRsPerBlockBatch)), kSortBlock, rank_sort_lds_bytes(cap), st>>>(w.desc, w.bucket_start, w.n_buckets, reinterpret_cast<uint2*>(w.key64A),
                                                                                             reinterpret_cast<uint2*>(w.key64B), w.valB, w.pred, cap, rank_sort_log_cells(cap), c.rs_max_cell, np, w.splitters, zero_in_scatter);
    else
        k_rs_bucket_sort<1><<<dim3(groups * kMaxBuckets), kSortBlock, rank_sort_lds_bytes(cap), st>>>(w.desc, w.bucket_start, w.n_buckets, reinterpret_cast<uint2*>(w.key64A),
                                                                                             reinterpret_cast<uint2*>(w.key64B), w.valB, w.pred, cap, rank_sort_log_cells(cap), c.rs_max_cell, np, w.splitters, zero_in_scatter);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace icet
