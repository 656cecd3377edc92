// icet_amd/csrc/icet_kernels.hip -- hand-written gfx950 kernels for the ICET hot path.
//
// Reference path (all under /root/reference): ICET::ICET src/icet.cpp:29-63 = fitScan1 (:68-107) ->
// prepScan2 (:254-277) -> runlen x fitScan2 (:372-436).  The device formulation is NOT a translation:
//
//   keyframe build (once per pair, scan 1)
//     k_scan1_spherical   utils::cartesianToSpherical (src/utils.cpp:93-119) + sort keys
//     [stable radix sort by r]                         (src/icet.cpp:72-77)
//     k_inverse_perm / k_exec_flags / k_scramble_src   the reference's one-step swap loop
//                                                      (src/icet.cpp:78-83) in parallel closed form
//     k_bin_hist + k_bin_scan + k_bin_scatter           stable multi-split of positions by voxel =
//                                                      sortSphericalCoordinates (src/icet.cpp:534-554)
//     k_fit_scan1         fitCells1 (src/icet.cpp:109-252): findCluster (:557-607), bounds filter
//                         (:609-652), mean/covariance, 3x3 eigen, sigma-point test (:654-696) -> L
//     k_compact_slots     dense voxel table -> compact "slots" of active voxels
//   Gauss-Newton loop (runlen x)
//     k_gn_accumulate     transform (:375-378) + c2s (:387) + binning (:388) + bounds filter (:299)
//                         + LDS-staged partial sums for the per-voxel mean/covariance (:303-306)
//     k_gn_solve          per-voxel fitCells2 algebra (:314-338), block reduction of H^T W H and
//                         H^T W dz (:401-402), 6x6 covariance (:410-417), conditioning (:443-492),
//                         dx and X += dx (:427-433)
//
// No MFMA anywhere: the largest contraction is 3x6; the point kernel is HBM/VALU co-limited and the
// rest is latency-bound bookkeeping.
#include <hip/hip_runtime.h>
#include <math.h>
#include "icet_internal.h"
#include "icet_device_math.h"

namespace icet {
namespace {

constexpr int kBlock = 256;
#ifndef ICET_ACC_BLOCK
#define ICET_ACC_BLOCK 512
#endif
#ifndef ICET_ACC_WAVES
#define ICET_ACC_WAVES 6
#endif
#ifndef ICET_ACC_PTS
#define ICET_ACC_PTS 4
#endif
constexpr int kAccPts = ICET_ACC_PTS;                      // consecutive points per lane per trip (two dwordx4 loads per coordinate)
constexpr int kAccBlock = ICET_ACC_BLOCK;        // k_gn_accumulate: the waves of a block share one copy of the pair's LDS tables
constexpr int kAccWavesPerSimd = ICET_ACC_WAVES; // register budget: 6 -> 84 VGPRs, no spills, three 512-thread blocks per CU (with 1536 blocks per 256-pair launch: 130 -> 121 us); 8 spills
constexpr double kTwoPi = 6.283185307179586476925286766559;
constexpr double kPi = 3.14159265358979323846;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// utils::cartesianToSpherical for one point (src/utils.cpp:93-119).  r is computed without
// contraction and with the correctly rounded sqrt so that its BITS match a plain IEEE evaluation:
// the radial sort + swap loop downstream is chaotic in the order of r.
__device__ __forceinline__ void c2s_point(float x, float y, float z, float& r, float& th, float& ph) {
    {
#pragma clang fp contract(off)
        float s = x * x + y * y;
        s = s + z * z;
        r = sqrtf(s);
    }
    th = atan2f(y, x);
    if (th < 0.0f) th = (float)((double)th + kTwoPi);
    ph = acosf(z / r);
    if (r != r) r = 1000.0f;
    if (th != th) th = 1000.0f;
    if (ph != ph) ph = 1000.0f;
}

__device__ __forceinline__ void s2c_point(float r, float th, float ph, float& x, float& y, float& z) {
    float sp, cp, st, ct;
    sincosf(ph, &sp, &cp); sincosf(th, &st, &ct);          // one range reduction per angle
    x = r * sp * ct; y = r * sp * st; z = r * cp;
}

// sortSphericalCoordinates' bin of one (theta, phi) pair: double arithmetic on float angles,
// truncation, modulo (src/icet.cpp:545-546).
__device__ __forceinline__ int voxel_of(float th, float ph, int T, int P) {
    int bt = static_cast<int>(((double)th / kTwoPi) * (double)T) % T;
    int bp = static_cast<int>(((double)ph / kPi) * (double)P) % P;
    return T * bp + bt;
}

__device__ __forceinline__ bool inside_bounds(float r, float az, float el, float az0, float az1, float el0, float el1, float inner, float outer) {
    return az >= az0 && az <= az1 && el >= el0 && el <= el1 && r >= inner && r <= outer;
}

// 1-D grid -> (pair, chunk).  With >= 8 pairs every chunk of a pair gets the same blockIdx % 8, i.e. (as the
// dispatcher is observed to deal blocks round-robin over the 8 XCDs) the same XCD and the same 4 MiB L2, and
// consecutive block ids walk through ONE group of 8 pairs before touching the next: the pointer-chasing keyframe
// kernels (rank / scramble / gather, ~1 MB of randomly accessed tables per pair) then find their pair's tables in
// L2 instead of HBM.  Speed only -- nothing depends on where a block actually lands.
__device__ __forceinline__ bool decode_block(int n_pairs, int chunks, int& pair, int& chunk) {
    if (n_pairs >= 8) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        pair = (j / chunks) * 8 + xcd; chunk = j % chunks;
        return pair < n_pairs;
    }
    pair = blockIdx.x / chunks; chunk = blockIdx.x % chunks;
    return true;
}
#define ICET_FOR_CHUNK_OF_SCAN1(i)                                                         \
    int pair, chunk;                                                                       \
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;                               \
    const PairDesc d = desc[pair];                                                         \
    int cs_ = (d.n1 + chunks - 1) / chunks; cs_ = (cs_ + kBlock - 1) / kBlock * kBlock;     \
    const int lo_ = chunk * cs_, hi_ = min(d.n1, lo_ + cs_);                               \
    for (int i = lo_ + threadIdx.x; i < hi_; i += kBlock)

// ------------------------------------------------------------------------------------------------
// keyframe build
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_scan1_spherical(const PairDesc* __restrict__ desc, float* __restrict__ r1, float* __restrict__ th1,
                                                            float* __restrict__ ph1, unsigned long long* __restrict__ key64, uint32_t* __restrict__ key32,
                                                            uint32_t* __restrict__ val, uint16_t* __restrict__ bin16, int T, int P, int n_pairs, int chunks,
                                                            const uint32_t* __restrict__ splitters, uint8_t* __restrict__ bkt, uint32_t* __restrict__ counts) {
    // First step of the rank sort fused in (icet_ranksort.hip): the pair's splitters are already known (k_rs_splitters
    // samples the radii straight from the Cartesian rows), so each row's bucket and this tile's bucket histogram cost no
    // extra pass over r1[].  splitters == nullptr: library-sort diagnostic path, nothing of this is needed.
    __shared__ uint32_t sp[kRankSortMaxBuckets];
    __shared__ uint32_t lh[kRankSortMaxBuckets];
    if (splitters) {
        int pair_, chunk_;
        if (decode_block(n_pairs, chunks, pair_, chunk_))
            for (int j = threadIdx.x; j < kRankSortMaxBuckets; j += kBlock) { sp[j] = splitters[(size_t)pair_ * kRankSortMaxBuckets + j]; lh[j] = 0u; }
        __syncthreads();
    }
    ICET_FOR_CHUNK_OF_SCAN1(i) {
        const float* x = d.s1; const float* y = d.s1 + d.ld1; const float* z = d.s1 + 2 * (size_t)d.ld1;
        float r, th, ph;
        c2s_point(x[i], y[i], z[i], r, th, ph);
        size_t o = (size_t)d.off1 + i;
        r1[o] = r; th1[o] = th; ph1[o] = ph;
        if (splitters) {
            const int b = rank_sort_bucket_of(__float_as_uint(r), sp);
            bkt[o] = (uint8_t)b;
            atomicAdd(&lh[b], 1u);
        }
        // r >= +0 (or 1000 for NaN): the bit pattern orders like the float; the pair id in the high word keeps
        // every pair's points contiguous, so one device-wide sort handles the whole batch
        if (key64) key64[o] = ((unsigned long long)pair << 32) | (unsigned long long)__float_as_uint(r);
        else if (key32) key32[o] = __float_as_uint(r);
        if (key64 || key32) val[o] = (uint32_t)i;               // library-sort path only
        bin16[o] = (uint16_t)voxel_of(th, ph, T, P);
    }
    if (splitters) {
        __syncthreads();
        for (int j = threadIdx.x; j < kRankSortMaxBuckets; j += kBlock) counts[((size_t)pair * chunks + chunk) * kRankSortMaxBuckets + j] = lh[j];
    }
}

// pred[s[i]] = i : rank of every original row.
__global__ __launch_bounds__(kBlock) void k_inverse_perm(const PairDesc* __restrict__ desc, const uint32_t* __restrict__ s, int32_t* __restrict__ pred,
                                                         int n_pairs, int chunks) {
    ICET_FOR_CHUNK_OF_SCAN1(i) pred[(size_t)d.off1 + s[(size_t)d.off1 + i]] = i;
}

// The reference "sorts" rows in place with
//     for i: if (index[i] != i) { swap(row i, row index[i]); swap(index[i], index[index[i]]); }
// (src/icet.cpp:78-83), which executes ONE step of each permutation cycle instead of following it.
// Visiting order makes step i execute iff row i is not a fixed point and was not frozen by an
// executed step i' = pred(i) < i.  So exec(v) is the parity of the length of the descending chain
// v, pred(v), pred(pred(v)), ... taken while pred(u) < u.
// Both walks below are pointer chases through ~1 MB of per-pair tables that sit in the XCD's L2 (see decode_block): they
// are bound by load latency, not bandwidth.  Each thread therefore advances EIGHT independent chains in lock step, so
// that eight loads are in flight per thread instead of one.
#ifndef ICET_WALK
#define ICET_WALK 8
#endif
constexpr int kWalk = ICET_WALK;

// The flag goes into bit 15 of the row's voxel id (V <= 32768): k_scramble_src needs "did step u execute" and "which voxel is
// row u in" for the same u, so one 2-byte random read serves both.
constexpr uint16_t kExecBit = 0x8000u, kBinMask = 0x7FFFu;

__global__ __launch_bounds__(kBlock) void k_exec_flags(const PairDesc* __restrict__ desc, const uint32_t* __restrict__ s, const int32_t* __restrict__ pred,
                                                       uint16_t* __restrict__ bin16, int32_t* __restrict__ flags, int max_walk, int n_pairs, int chunks) {
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;
    const PairDesc d = desc[pair];
    int cs_ = (d.n1 + chunks - 1) / chunks; cs_ = (cs_ + kBlock - 1) / kBlock * kBlock;
    const int lo_ = chunk * cs_, hi_ = min(d.n1, lo_ + cs_);
    const size_t o = d.off1;
    for (int base = lo_ + threadIdx.x; base < hi_; base += kWalk * kBlock) {
        int u[kWalk], len[kWalk]; bool act[kWalk], moved[kWalk];
#pragma unroll
        for (int k = 0; k < kWalk; k++) {
            const int v = base + k * kBlock;
            const bool valid = v < hi_;
            moved[k] = false;                                    // a fixed point of the permutation: pred[v] == v (known after the first load)
            act[k] = valid; u[k] = v; len[k] = 0;
        }
        bool any = true;
        while (any) {
            int p[kWalk];
#pragma unroll
            for (int k = 0; k < kWalk; k++) p[k] = act[k] ? pred[o + u[k]] : 0;
            any = false;
#pragma unroll
            for (int k = 0; k < kWalk; k++) {
                if (act[k]) {
                    if (len[k] == 0) moved[k] = (p[k] != u[k]);   // first step reads pred[v] itself: no separate pass over s[]
                    if (p[k] >= u[k]) act[k] = false;
                    else { u[k] = p[k]; len[k]++; if (len[k] > max_walk) { atomicOr(&flags[pair], 1); act[k] = false; } }
                }
                any |= act[k];
            }
        }
#pragma unroll
        for (int k = 0; k < kWalk; k++) {
            const int v = base + k * kBlock;
            if (v < hi_ && moved[k] && !(len[k] & 1)) bin16[o + v] |= kExecBit;      // k_scan1_spherical wrote the id with the bit clear
        }
    }
}

// src[v] = original row that ends at position v after the swap loop.  Position v receives row
// pred(v), except at the head of a run of executed steps, where the row arrives from the end of
// the forward chain v -> s[v] -> s[s[v]] ... while the steps executed.
// Fused with the first step of the voxel multi-split (k_bin_hist): the row that lands on a position is known here, so
// its voxel id and this tile's voxel histogram cost no extra pass over src[].
#ifndef ICET_SCR_WAVES
#define ICET_SCR_WAVES 6      /* <= 80 VGPRs: 6 waves per SIMD for a latency-bound walk (measured: -35 us per 256-pair keyframe; 8 spills) */
#endif
__global__ __launch_bounds__(kBlock, ICET_SCR_WAVES) void k_scramble_src(const PairDesc* __restrict__ desc, const uint32_t* __restrict__ s, const int32_t* __restrict__ pred,
                                                         int32_t* __restrict__ src, int32_t* __restrict__ flags, int max_walk,
                                                         const uint16_t* __restrict__ bin16, uint16_t* __restrict__ binpos, uint32_t* __restrict__ counts, int V,
                                                         int n_pairs, int chunks) {
    extern __shared__ uint32_t lh[];
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;
    const PairDesc d = desc[pair];
    int cs_ = (d.n1 + chunks - 1) / chunks; cs_ = (cs_ + kBlock - 1) / kBlock * kBlock;
    const int lo_ = chunk * cs_, hi_ = min(d.n1, lo_ + cs_);
    const size_t o = d.off1;
    for (int b = threadIdx.x; b < V; b += kBlock) lh[b] = 0u;
    __syncthreads();
    for (int base = lo_ + threadIdx.x; base < hi_; base += kWalk * kBlock) {
        int f[kWalk], u[kWalk], len[kWalk]; bool act[kWalk];
        int pv[kWalk]; uint16_t wv[kWalk], fb[kWalk];    // fb: packed word of the row f[] (its voxel id is what the histogram needs)
#pragma unroll
        for (int k = 0; k < kWalk; k++) {
            const int v = base + k * kBlock;
            const bool valid = v < hi_;
            const size_t i = o + (valid ? v : lo_);
            pv[k] = pred[i]; wv[k] = bin16[i];
        }
        uint16_t wp[kWalk];
#pragma unroll
        for (int k = 0; k < kWalk; k++) wp[k] = bin16[o + pv[k]];
#pragma unroll
        for (int k = 0; k < kWalk; k++) {
            const int v = base + k * kBlock;
            const bool valid = v < hi_;
            const bool moved = valid && pv[k] != v;             // s[v] != v  <=>  pred[v] != v (fixed points of a permutation)
            f[k] = moved ? pv[k] : v; fb[k] = moved ? wp[k] : wv[k];
            act[k] = moved && (wv[k] & kExecBit) && !(wp[k] & kExecBit);   // head of a run of executed steps
            u[k] = v; len[k] = 0;
        }
        bool any = false;
#pragma unroll
        for (int k = 0; k < kWalk; k++) any |= act[k];
        while (any) {
            int nu[kWalk];
#pragma unroll
            for (int k = 0; k < kWalk; k++) nu[k] = act[k] ? (int)s[o + u[k]] : 0;
            uint16_t ne[kWalk];
#pragma unroll
            for (int k = 0; k < kWalk; k++) ne[k] = act[k] ? bin16[o + nu[k]] : (uint16_t)0;
            any = false;
#pragma unroll
            for (int k = 0; k < kWalk; k++) {
                if (act[k]) {
                    u[k] = nu[k]; len[k]++;
                    if (!(ne[k] & kExecBit)) { f[k] = u[k]; fb[k] = ne[k]; act[k] = false; }
                    else if (len[k] > max_walk) { atomicOr(&flags[pair], 1); f[k] = u[k]; fb[k] = ne[k]; act[k] = false; }
                }
                any |= act[k];
            }
        }
#pragma unroll
        for (int k = 0; k < kWalk; k++) {
            const int v = base + k * kBlock;
            if (v < hi_) {
                src[o + v] = f[k];
                const uint16_t b = fb[k] & kBinMask;
                binpos[o + v] = b;
                atomicAdd(&lh[b], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* out = counts + ((size_t)pair * chunks + chunk) * V;
    for (int b = threadIdx.x; b < V; b += kBlock) out[b] = lh[b];
}

// Serial fallback for adversarial permutations (walks longer than max_walk): one lane replays the
// literal swap loop on indices.  Never taken on lidar data (observed walk depth <= 14).
__global__ void k_scramble_serial(const PairDesc* __restrict__ desc, const uint32_t* __restrict__ s, int32_t* __restrict__ idx_tmp,
                                  int32_t* __restrict__ src, const int32_t* __restrict__ flags) {
    const int pair = blockIdx.x;
    if (threadIdx.x != 0 || !(flags[pair] & 1)) return;
    const PairDesc d = desc[pair];
    const size_t o = d.off1;
    for (int i = 0; i < d.n1; i++) { idx_tmp[o + i] = (int)s[o + i]; src[o + i] = i; }
    for (int i = 0; i < d.n1; i++) {
        int j = idx_tmp[o + i];
        if (j != i) {
            int t = src[o + i]; src[o + i] = src[o + j]; src[o + j] = t;
            idx_tmp[o + i] = idx_tmp[o + j]; idx_tmp[o + j] = j;
        }
    }
}

// ---- grouping scan-1 rows by voxel, in ascending POSITION order inside each voxel -------------------------------
// sortSphericalCoordinates appends point indices to per-voxel vectors while walking the (scrambled) array front to back
// (src/icet.cpp:539-550), so findCluster later sees each voxel's rows in ascending position.  That is a STABLE
// multi-split of the positions by voxel id.  Done here in three small kernels instead of a second library sort:
//   k_bin_hist     per tile of positions: histogram of voxel ids (LDS), voxel id of every position
//   k_bin_scan     per pair: exclusive scan over (voxel, tile) -> bin_start[] and each tile's base offset per voxel
//   k_bin_scatter  one wave per tile walks its positions 64 at a time; lanes holding the same voxel find each other
//                  with a ballot per id bit (match-any), so rank = popcount of lower peers -- stable by construction --
//                  and the row's spherical coordinates are written straight to their final place.
__global__ __launch_bounds__(kBlock) void k_bin_hist(const PairDesc* __restrict__ desc, const int32_t* __restrict__ src, const uint16_t* __restrict__ bin16,
                                                     uint16_t* __restrict__ binpos, uint32_t* __restrict__ counts, const int32_t* __restrict__ flags,
                                                     int V, int n_pairs, int chunks, int force) {
    extern __shared__ uint32_t lh[];
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;
    if (!force && !(flags[pair] & 1)) return;   // k_scramble_src already produced binpos / counts; redo only after the serial replay (or when it did not run)
    const PairDesc d = desc[pair];
    int cs_ = (d.n1 + chunks - 1) / chunks; cs_ = (cs_ + kBlock - 1) / kBlock * kBlock;
    const int lo_ = chunk * cs_, hi_ = min(d.n1, lo_ + cs_);
    for (int b = threadIdx.x; b < V; b += kBlock) lh[b] = 0u;
    __syncthreads();
    const size_t o = d.off1;
    for (int v = lo_ + threadIdx.x; v < hi_; v += kBlock) {
        const uint16_t b = bin16[o + src[o + v]] & kBinMask;
        binpos[o + v] = b;
        atomicAdd(&lh[b], 1u);
    }
    __syncthreads();
    uint32_t* out = counts + ((size_t)pair * chunks + chunk) * V;
    for (int b = threadIdx.x; b < V; b += kBlock) out[b] = lh[b];
}

// Exclusive scan over (class, tile) in class-major order, in two steps so that a single large pair (7200 voxels x 240 tiles)
// is not scanned by ONE block: k_bin_tiles (one thread per class, blocks over classes) turns each class's per-tile counts into
// per-tile offsets and leaves the class total in class_start[]; k_bin_scan (one block per pair) scans the totals in place.
__global__ __launch_bounds__(kBlock) void k_bin_tiles(const uint32_t* __restrict__ counts, uint32_t* __restrict__ tile_base, int32_t* __restrict__ class_start,
                                                      int V, int chunks) {
    const int pair = blockIdx.y, b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= V) return;
    const uint32_t* c = counts + (size_t)pair * chunks * V + b;
    uint32_t* tb = tile_base + (size_t)pair * chunks * V + b;
    int tot = 0, t = 0;
    for (; t + 8 <= chunks; t += 8) {                         // 8 independent loads in flight
        uint32_t x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = c[(size_t)(t + k) * V];
#pragma unroll
        for (int k = 0; k < 8; k++) { tb[(size_t)(t + k) * V] = (uint32_t)tot; tot += (int)x[k]; }
    }
    for (; t < chunks; t++) { const uint32_t x = c[(size_t)t * V]; tb[(size_t)t * V] = (uint32_t)tot; tot += (int)x; }
    class_start[(size_t)pair * (V + 1) + b] = tot;
}

__global__ __launch_bounds__(kBlock) void k_bin_scan(int32_t* __restrict__ class_start, int V) {
    __shared__ int wave_tot[kBlock / 64];
    __shared__ int base;
    const int pair = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int v0 = 0; v0 < V; v0 += kBlock) {
        const int b = v0 + threadIdx.x;
        const int tot = (b < V) ? class_start[(size_t)pair * (V + 1) + b] : 0;
        int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; k++) woff += wave_tot[k];
        const int bb = base;
        if (b < V) class_start[(size_t)pair * (V + 1) + b] = bb + woff + incl - tot;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) base = bb + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) class_start[(size_t)pair * (V + 1) + V] = base;
}

// One block per tile (<= 2048 positions); wave w owns the w-th quarter (<= 8 rounds of 64 positions).  Everything
// global is loaded up front and stored at the end, so the only serial chain is 8 rounds of ballots + LDS.
constexpr int kScatterRounds = kKfMaxPtsPerThread;   // a tile is at most 4 waves x this many rounds x 64 positions
__global__ __launch_bounds__(kBlock) void k_bin_scatter(const PairDesc* __restrict__ desc, const int32_t* __restrict__ src, const uint16_t* __restrict__ binpos,
                                                        const uint32_t* __restrict__ tile_base, const int32_t* __restrict__ bin_start,
                                                        uint32_t* __restrict__ sorted_row, int V, int vbits, int n_pairs, int chunks) {
    extern __shared__ uint32_t lb[];                                   // 4 x V : per-wave counts, then per-wave running offsets
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;
    const PairDesc d = desc[pair];
    int cs_ = (d.n1 + chunks - 1) / chunks; cs_ = (cs_ + kBlock - 1) / kBlock * kBlock;      // multiple of 256, <= 64 * 4 * kScatterRounds
    const int lo_ = chunk * cs_, hi_ = min(d.n1, lo_ + cs_);
    if (lo_ >= hi_) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qs = cs_ / 4;                                            // multiple of 64
    const int wlo = lo_ + wave * qs, whi = min(hi_, wlo + qs);
    const int rounds = qs / 64;
    for (int i = threadIdx.x; i < 4 * V; i += kBlock) lb[i] = 0u;
    __syncthreads();
    const size_t o = d.off1;
    uint32_t bb[kScatterRounds]; int row[kScatterRounds]; bool ok[kScatterRounds];
    uint32_t* mine = lb + wave * V;
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++) {
        const int v = wlo + 64 * k + lane;
        ok[k] = (k < rounds) & (v < whi);
        bb[k] = ok[k] ? (uint32_t)binpos[o + v] : 0u;
        row[k] = ok[k] ? src[o + v] : 0;
        if (ok[k]) atomicAdd(&mine[bb[k]], 1u);
    }
    __syncthreads();
    {
        const uint32_t* tb = tile_base + ((size_t)pair * chunks + chunk) * V;
        const int32_t* bst = bin_start + (size_t)pair * (V + 1);
        for (int b = threadIdx.x; b < V; b += kBlock) {
            const uint32_t c0 = lb[b], c1 = lb[V + b], c2 = lb[2 * V + b];
            const uint32_t base = (uint32_t)bst[b] + tb[b];
            lb[b] = base; lb[V + b] = base + c0; lb[2 * V + b] = base + c0 + c1; lb[3 * V + b] = base + c0 + c1 + c2;
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t dest[kScatterRounds];
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++) {
        unsigned long long peers = __ballot(ok[k]);
        for (int q = 0; q < vbits; q++) {
            const bool bit = (bb[k] >> q) & 1u;
            const unsigned long long m = __ballot(ok[k] && bit);
            peers &= bit ? m : ~m;
        }
        dest[k] = 0u;
        if (ok[k]) {
            const int rank = __popcll(peers & lt);
            dest[k] = mine[bb[k]] + (uint32_t)rank;
            if (rank == 0) mine[bb[k]] += (uint32_t)__popcll(peers);  // one leader per distinct voxel in this round
        }
    }
    // one scattered 4-byte store per row; k_fit_scan1 gathers the coordinates through it (element-wise scattered stores of
    // the three coordinate arrays cost 4x more than gathering them, and a separate gather pass 0.15 ms more than gathering
    // inside the fit, whose independent loads hide the latency)
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++)
        if (ok[k]) sorted_row[o + dest[k]] = (uint32_t)row[k];
}

// fitCells1 (src/icet.cpp:109-252): one wavefront per angular bin.
// The per-bin tail -- 3x3 eigen-decomposition, the six sigma points, the slot records -- is scalar work: it runs in
// k_fit_finish with one LANE per bin instead of here with one WAVE per bin (measured: 0.21 ms of the 0.42 ms this kernel
// took on 256 pairs was 64 lanes executing the same eigen-solve).
#ifndef ICET_FIT_WAVES
#define ICET_FIT_WAVES 7      /* <= 72 VGPRs: one more wave per SIMD hides the gather latency (measured: -47 us per 256-pair keyframe; 8 is worse) */
#endif
__global__ __launch_bounds__(kBlock, ICET_FIT_WAVES) void k_fit_scan1(const PairDesc* __restrict__ desc, const int32_t* __restrict__ bin_start,
                                                      const uint32_t* __restrict__ sorted_row,
                                                      const float* __restrict__ r1, const float* __restrict__ th1, const float* __restrict__ ph1,
                                                      FitMid* __restrict__ midD, int T, int P, int n, float thresh, float buff) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int V = T * P;
    const int v = blockIdx.x * (kBlock / 64) + wave;
    const int pair = blockIdx.y;
    if (v >= V) return;
    const PairDesc d = desc[pair];
    const int theta = v % T, phi = v / T;
    // src/icet.cpp:136-139: (float / int) -> float, times a double constant, stored to float
    const float az0 = (float)((double)((float)theta / (float)T) * kTwoPi);
    const float az1 = (float)((double)((float)(theta + 1) / (float)T) * kTwoPi);
    const float el0 = (float)((double)((float)phi / (float)P) * kPi);
    const float el1 = (float)((double)((float)(phi + 1) / (float)P) * kPi);
    const int bs = bin_start[(size_t)pair * (V + 1) + v];
    const int cnt = bin_start[(size_t)pair * (V + 1) + v + 1] - bs;
    const size_t base = (size_t)d.off1 + bs;
    // rows of this bin in (scrambled) position order: sorted_row[base + i] is the row of the pair's input-order tables
    const size_t po = (size_t)d.off1;
    auto RS = [&](int i) { return r1[po + sorted_row[base + i]]; };

    float inner = 0.f, outer = 0.f;
    int has_fit = 0;
    float mean[3] = {0.f, 0.f, 0.f};
    float cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // xx xy xz yy yz zz

    if (cnt >= n) {
        // The bin's first 4 x 64 rows are fetched up front with independent loads (most bins hold ~100-400 rows), so
        // the serial walks below run on registers instead of paying one memory round trip per 64 rows.
        constexpr int kCache = 4;
        float pr[kCache], pth[kCache], pph[kCache];
#pragma unroll
        for (int k = 0; k < kCache; k++) {
            const int i = lane + 64 * k;
            const bool okk = i < cnt;
            const uint32_t row = okk ? sorted_row[base + i] : 0u;
            pr[k] = okk ? r1[po + row] : 0.f; pth[k] = okk ? th1[po + row] : 0.f; pph[k] = okk ? ph1[po + row] : 0.f;
        }
        // ---- findCluster (src/icet.cpp:557-607): first run of >= n consecutive points whose
        // successive |dr| <= thresh, walking the bin in stored (scrambled) order.
        int run_start = 0; float front = 0.f; float carry_prev = 0.f; bool found = false;
        // A break is a point that does not continue the current run; a run is reported at the first break that closes
        // >= n points.  In the scrambled order most points are breaks, so instead of visiting the breaks of a 64-point
        // chunk one after another, every break lane looks up the break before it with a prefix-max scan and the first
        // lane whose run is long enough is picked with a ballot.
        auto walk = [&](int c0, float r) {
            const int i = c0 + lane; const bool valid = i < cnt;
            float prev = __shfl_up(r, 1);
            if (lane == 0) prev = carry_prev;
            const bool brk = valid && (i == 0 || !(fabsf(prev - r) <= thresh));
            int pm = brk ? i : -1;                                   // inclusive prefix max of break positions
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(pm, o); if (lane >= o) pm = max(pm, t); }
            int prevb = __shfl_up(pm, 1);                            // last break strictly before this lane ...
            if (lane == 0) prevb = -1;
            prevb = max(prevb, run_start);                           // ... or the run carried in from earlier chunks
            const unsigned long long hit = __ballot(brk && (i - prevb >= n));
            if (hit) {
                const int b = __ffsll((long long)hit) - 1;
                const int rs0 = __shfl(prevb, b);                    // start of the run that this break closes
                const float back = (b > 0) ? __shfl(r, b - 1) : carry_prev;
                const float fr = (rs0 >= c0) ? __shfl(r, rs0 - c0) : front;
                inner = fr - buff; outer = back + buff; found = true;
            } else {
                const int last = __shfl(pm, 63);                     // last break of this chunk, if any
                if (last >= 0) { run_start = last; front = __shfl(r, last - c0); }
            }
            carry_prev = __shfl(r, 63);
        };
#pragma unroll
        for (int k = 0; k < kCache; k++) if (64 * k < cnt && !found) walk(64 * k, pr[k]);
        for (int c0 = 64 * kCache; c0 < cnt && !found; c0 += 64) walk(c0, (c0 + lane < cnt) ? RS(c0 + lane) : 0.f);
        if (!found && cnt - run_start >= n) {
            if (front != 0.f) { const float back = RS(cnt - 1); inner = front - buff; outer = back + buff; }
            else { inner = 0.f; outer = 0.f; }
        }
        // ---- filterPointsInsideCluster + sphericalToCartesian + mean (src/icet.cpp:155-160).  The Cartesian
        // coordinates of the cached rows stay in registers for the second (centred) pass.
        float cx[kCache], cy[kCache], cz[kCache]; bool cin[kCache];
        // Sums in double: the reference's float sums run in Eigen's (unspecified, vectorised) order, so there is no
        // order to reproduce; the exactly rounded value is the one every order approximates, and it keeps the
        // eigenvectors -- and with them the sigma-point masks -- as stable as the data allows.
        double sx = 0.0, sy = 0.0, sz = 0.0; int rows = 0;
#pragma unroll
        for (int k = 0; k < kCache; k++) {
            const int i = lane + 64 * k;
            cin[k] = false; cx[k] = cy[k] = cz[k] = 0.f;
            if (i < cnt && inside_bounds(pr[k], pth[k], pph[k], az0, az1, el0, el1, inner, outer)) {
                s2c_point(pr[k], pth[k], pph[k], cx[k], cy[k], cz[k]); cin[k] = true;
                sx += (double)cx[k]; sy += (double)cy[k]; sz += (double)cz[k]; rows++;
            }
        }
        for (int i = lane + 64 * kCache; i < cnt; i += 64) {
            const uint32_t row = sorted_row[base + i];
            const float r = r1[po + row], th = th1[po + row], ph = ph1[po + row];
            if (inside_bounds(r, th, ph, az0, az1, el0, el1, inner, outer)) {
                float x, y, z; s2c_point(r, th, ph, x, y, z);
                sx += (double)x; sy += (double)y; sz += (double)z; rows++;
            }
        }
        sx = wave_sum_d(sx); sy = wave_sum_d(sy); sz = wave_sum_d(sz); rows = wave_sum_i(rows);
        if ((double)outer > 0.1 && rows * 3 >= n) {       // src/icet.cpp:158 (size() counts coefficients)
            has_fit = 1;
            mean[0] = (float)(sx / (double)rows); mean[1] = (float)(sy / (double)rows); mean[2] = (float)(sz / (double)rows);
            double c[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < kCache; k++) {
                if (cin[k]) {
                    const float dx = cx[k] - mean[0], dy = cy[k] - mean[1], dz = cz[k] - mean[2];
                    c[0] += (double)(dx * dx); c[1] += (double)(dx * dy); c[2] += (double)(dx * dz);
                    c[3] += (double)(dy * dy); c[4] += (double)(dy * dz); c[5] += (double)(dz * dz);
                }
            }
            for (int i = lane + 64 * kCache; i < cnt; i += 64) {
                const uint32_t row = sorted_row[base + i];
            const float r = r1[po + row], th = th1[po + row], ph = ph1[po + row];
                if (inside_bounds(r, th, ph, az0, az1, el0, el1, inner, outer)) {
                    float x, y, z; s2c_point(r, th, ph, x, y, z);
                    const float dx = x - mean[0], dy = y - mean[1], dz = z - mean[2];
                    c[0] += (double)(dx * dx); c[1] += (double)(dx * dy); c[2] += (double)(dx * dz);
                    c[3] += (double)(dy * dy); c[4] += (double)(dy * dz); c[5] += (double)(dz * dz);
                }
            }
            const float den = (float)(rows - 1);
#pragma unroll
            for (int k = 0; k < 6; k++) cov[k] = (float)wave_sum_d(c[k]) / den;
        }
    }
    // one 64-byte record per bin, staged through LDS so that 16 lanes store it with one coalesced instruction
    __shared__ float stage[kBlock / 64][16];
    if (lane == 0) {
        float* g = stage[wave];
        g[0] = mean[0]; g[1] = mean[1]; g[2] = mean[2];
#pragma unroll
        for (int k = 0; k < 6; k++) g[3 + k] = cov[k];
        g[9] = inner; g[10] = outer; g[11] = __int_as_float(cnt); g[12] = __int_as_float(has_fit); g[13] = g[14] = g[15] = 0.f;
    }
    // same wave wrote and reads: LDS operations of one wave complete in order
    if (lane < 16) reinterpret_cast<float*>(midD + (size_t)pair * V + v)[lane] = stage[wave][lane];
}

// fitCells1's per-bin tail (src/icet.cpp:181-252), one lane per angular bin: eigen-decomposition, U = eigenvectors^T, the six
// sigma points and their inside test -> L, the scan-1 half of the gate at :290, and the records of the active voxels.
__global__ __launch_bounds__(kBlock) void k_fit_finish(const FitMid* __restrict__ midD, SlotHot* __restrict__ hotD, SlotFit* __restrict__ fitD,
                                                       int32_t* __restrict__ activeD, AuxDev aux, int T, int P, int n) {
    const int V = T * P;
    const int v = blockIdx.x * kBlock + threadIdx.x;
    const int pair = blockIdx.y;
    if (v >= V) return;
    const size_t o = (size_t)pair * V + v;
    const FitMid m = midD[o];
    const int theta = v % T, phi = v / T;
    // src/icet.cpp:136-139: (float / int) -> float, times a double constant, stored to float
    const float az0 = (float)((double)((float)theta / (float)T) * kTwoPi);
    const float az1 = (float)((double)((float)(theta + 1) / (float)T) * kTwoPi);
    const float el0 = (float)((double)((float)phi / (float)P) * kPi);
    const float el1 = (float)((double)((float)(phi + 1) / (float)P) * kPi);
    const float inner = m.inner, outer = m.outer;
    float ev[3] = {0.f, 0.f, 0.f}, Vm[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float Ld[3] = {0.f, 0.f, 0.f};
    int active = 0;
    if (m.has_fit) {
        icetdev::eig3_sym(m.cov[0], m.cov[1], m.cov[3], m.cov[2], m.cov[4], m.cov[5], ev, Vm);
        // sigma points mu +- 2 sqrt(lambda_k) * (row k of V): rotated = diag(2 sqrt(lambda)) * U^T = diag(.) * V (src/icet.cpp:187-202).
        // testSigmaPoints walks j = 0..5 and leaves the loop AFTER testing the first point with r > outer (:669-686).
        bool inside[6] = {false, false, false, false, false, false};
        bool done = false;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int k = j >> 1;
            const float al = 2.0f * sqrtf(ev[k]);
            const float sgn = (j & 1) ? -1.f : 1.f;
            const float px = m.mean[0] + sgn * (al * Vm[3 * k]), py = m.mean[1] + sgn * (al * Vm[3 * k + 1]), pz = m.mean[2] + sgn * (al * Vm[3 * k + 2]);
            if (!done) {
                float r, az, el; c2s_point(px, py, pz, r, az, el);
                inside[j] = inside_bounds(r, az, el, az0, az1, el0, el1, inner, outer);
                done = r > outer;
            }
        }
        Ld[0] = (inside[0] || inside[1]) ? 1.f : 0.f; Ld[1] = (inside[2] || inside[3]) ? 1.f : 0.f; Ld[2] = (inside[4] || inside[5]) ? 1.f : 0.f;
        active = (m.cnt > n && outer > 1.f) ? 1 : 0;      // scan-1 half of the gate at src/icet.cpp:290
    }
    // Records are only needed for ACTIVE voxels (k_compact_slots copies nothing else)
    if (active) {
        SlotHot h; h.az0 = az0; h.az1 = az1; h.el0 = el0; h.el1 = el1; h.inner = inner; h.outer = outer;
        h.mu[0] = m.mean[0]; h.mu[1] = m.mean[1]; h.mu[2] = m.mean[2]; h.v = v; h.pad[0] = h.pad[1] = 0;
        hotD[o] = h;
        SlotFit f; f.mu[0] = m.mean[0]; f.mu[1] = m.mean[1]; f.mu[2] = m.mean[2];
        const float d1 = (float)(m.cnt - 1);
#pragma unroll
        for (int k = 0; k < 6; k++) f.s1n[k] = m.cov[k] / d1;
#pragma unroll
        for (int k = 0; k < 3; k++) { f.M[3 * k] = Ld[k] * Vm[3 * k]; f.M[3 * k + 1] = Ld[k] * Vm[3 * k + 1]; f.M[3 * k + 2] = Ld[k] * Vm[3 * k + 2]; }
        f.n1 = m.cnt; f.v = v;
        fitD[o] = f;
    }
    activeD[o] = active;
    if (aux.bounds) { float* b = aux.bounds + o * 6; b[0] = az0; b[1] = az1; b[2] = el0; b[3] = el1; b[4] = inner; b[5] = outer; }
    if (aux.n1_raw) aux.n1_raw[o] = m.cnt;
    if (aux.has_fit) aux.has_fit[o] = m.has_fit;
    if (aux.mu1) { aux.mu1[o * 3] = m.mean[0]; aux.mu1[o * 3 + 1] = m.mean[1]; aux.mu1[o * 3 + 2] = m.mean[2]; }
    if (aux.sigma1) { float* sg = aux.sigma1 + o * 9; sg[0] = m.cov[0]; sg[1] = m.cov[1]; sg[2] = m.cov[2]; sg[3] = m.cov[1]; sg[4] = m.cov[3]; sg[5] = m.cov[4]; sg[6] = m.cov[2]; sg[7] = m.cov[4]; sg[8] = m.cov[5]; }
    if (aux.evecs1) for (int k = 0; k < 9; k++) aux.evecs1[o * 9 + k] = Vm[k];
    if (aux.l_diag) { aux.l_diag[o * 3] = Ld[0]; aux.l_diag[o * 3 + 1] = Ld[1]; aux.l_diag[o * 3 + 2] = Ld[2]; }
}

// Dense per-voxel records -> compact slots in voxel order (phi-major, theta inner: the reference's
// accumulation order, src/icet.cpp:391-404).
__global__ __launch_bounds__(kBlock) void k_compact_slots(const SlotHot* __restrict__ hotD, const SlotFit* __restrict__ fitD, const int32_t* __restrict__ activeD,
                                                          SlotHot* __restrict__ hotS, SlotFit* __restrict__ fitS, int16_t* __restrict__ slot_of_voxel,
                                                          int32_t* __restrict__ n_slots, uint32_t* __restrict__ acc, int V) {
    __shared__ int wave_tot[kBlock / 64];
    __shared__ int base;
    const int pair = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int v0 = 0; v0 < V; v0 += kBlock) {
        const int v = v0 + threadIdx.x;
        const int a = (v < V) ? activeD[(size_t)pair * V + v] : 0;
        const unsigned long long m = __ballot(a != 0);
        const int excl = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; k++) woff += wave_tot[k];
        const int b = base;
        if (v < V) {
            if (a) {
                const int s = b + woff + excl;
                slot_of_voxel[(size_t)pair * ((V + 1) & ~1) + v] = (int16_t)s;
                hotS[(size_t)pair * V + s] = hotD[(size_t)pair * V + v];
                fitS[(size_t)pair * V + s] = fitD[(size_t)pair * V + v];
            } else {
                slot_of_voxel[(size_t)pair * ((V + 1) & ~1) + v] = (int16_t)-1;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < kBlock / 64; k++) t += wave_tot[k]; base = b + t; }
        __syncthreads();
    }
    const int ns = base;
    if (threadIdx.x == 0) n_slots[pair] = ns;
    for (int i = threadIdx.x; i < ns * kAccWords; i += kBlock) acc[(size_t)pair * V * kAccWords + i] = 0u;
}

// ------------------------------------------------------------------------------------------------
// Gauss-Newton loop
// ------------------------------------------------------------------------------------------------
// Per-pair transform record, kXf floats: t[3] | R[9] (utils::R, src/utils.cpp:144-152, row-major) | angles[3] | pad |
// J[27] (get_H's three derivative matrices, src/icet.cpp:507-529).  Written once per iteration by the lane that updates
// X, so the six sin/cos are evaluated once and serve both the next point pass (R) and the next voxel pass (J).
constexpr int kXf = 48;
__device__ __forceinline__ void write_xf(float* xf, const float X[6]) {
    const float phi = X[3], theta = X[4], psi = X[5];
    float sph, cph, sth, cth, sps, cps;
    sincosf(phi, &sph, &cph); sincosf(theta, &sth, &cth); sincosf(psi, &sps, &cps);
    xf[0] = X[0]; xf[1] = X[1]; xf[2] = X[2];
    xf[3] = cth * cps;  xf[4] = sps * cph + sph * sth * cps;  xf[5] = sph * sps - sth * cph * cps;
    xf[6] = -sps * cth; xf[7] = cph * cps - sph * sth * sps;  xf[8] = sph * cps + sth * sps * cph;
    xf[9] = sth;        xf[10] = -sph * cth;                  xf[11] = cph * cth;
    xf[12] = phi; xf[13] = theta; xf[14] = psi; xf[15] = 0.f;
    float* J = xf + 16;
    J[0] = 0.f; J[1] = -sps * sph + cph * sth * cps; J[2] = cph * sps + sth * sph * cps;
    J[3] = 0.f; J[4] = -sph * cps - cph * sth * sps; J[5] = cph * cps - sth * sps * sph;
    J[6] = 0.f; J[7] = -cph * cth;                   J[8] = -sph * cth;
    J[9] = -sth * cps;  J[10] = cth * sph * cps;  J[11] = -cth * cph * cps;
    J[12] = sps * sth;  J[13] = -cth * sph * sps; J[14] = cth * sps * cph;
    J[15] = cth;        J[16] = sph * sth;        J[17] = -sth * cph;
    J[18] = -cth * sps; J[19] = cps * cph - sph * sth * sps;  J[20] = cps * sph + sth * cph * sps;
    J[21] = -cps * cth; J[22] = -sps * cph - sph * sth * cps; J[23] = -sph * sps + sth * cps * cph;
    J[24] = 0.f; J[25] = 0.f; J[26] = 0.f;
}

__global__ void k_init_state(const float* __restrict__ x0, float* __restrict__ X, float* __restrict__ xf, int n_pairs) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    float x[6];
    for (int k = 0; k < 6; k++) { x[k] = x0 ? x0[p * 6 + k] : 0.f; X[p * 6 + k] = x[k]; }
    write_xf(xf + p * kXf, x);
}

// sortSphericalCoordinates' bin index WITHOUT the double divide: thr[k] is the smallest float whose
// reference bin (double arithmetic, src/icet.cpp:545-546) is >= k, built on the host with exactly that
// arithmetic, so "largest k with thr[k] <= a" is bit-for-bit the reference's truncation.  The float
// product only proposes a candidate (off by at most one).  Returns nb when a lies beyond the last
// edge (a == float(2 pi), float(pi) or the 1000 sentinel): the caller then takes the literal formula.
__device__ __forceinline__ int bin_from_table(float a, const float* __restrict__ thr, int nb, float scale) {
    int k = static_cast<int>(a * scale);
    k = min(k, nb - 1);
    const float lo = thr[k], hi = thr[k + 1];
    k += (a >= hi) ? 1 : 0;
    k -= (a < lo) ? 1 : 0;
    return k;
}

// One pass of fitScan2's point work over a chunk of one pair's scan 2.
//   points2 = (points2_OG.rowwise() + t) * R          src/icet.cpp:375-378
//   cartesianToSpherical, sortSphericalCoordinates     src/icet.cpp:387-388
//   filterPointsInsideCluster                          src/icet.cpp:299
// and the sums that give mean / covariance of the surviving points (src/icet.cpp:303-306), taken
// about the voxel's scan-1 mean so that one pass in float keeps its digits.
//
// The literal evaluation costs ~250 VALU instructions per point (atan2f, acosf, a divide, an exact
// sqrt) for 12 bytes of traffic.  Every one of its DECISIONS, though, is a comparison of an angle
// or a range against a voxel edge, so the kernel classifies each point on two monotone stand-ins
// that need no transcendental --
//     polar bin   : w  = -z / |q|                      (monotone in phi   = acos(z/|q|))
//     azimuth bin : pa = "diamond angle" of (x, y)     (monotone in theta = atan2(y, x))
// -- through LDS look-up tables whose cells are narrower than half a bin: a cell names the one
// edge a point in it can be near, one compare picks the side.  A point closer to an edge than a
// guard band (a few float ulps, covering the rounding of both evaluations) is re-done with the
// literal formulas (classify_exact), so the result is the literal evaluation's, decision for
// decision.  Away from the edges the azimuth/polar bounds of filterPointsInsideCluster hold by
// construction and only the radial test remains (same guard-band rule).
//
// Each lane takes 4 CONSECUTIVE points per trip (three 16-byte loads) and keeps a run-length
// accumulator in registers: lidar storage order puts neighbours in the same voxel, so a lane
// flushes to LDS about once per trip instead of once per point, and no cross-lane reduction is
// needed.  LDS holds the pair's voxel->slot map, the two LUTs, the hot slot records and the block's
// partial sums, which are flushed with one global atomic per touched word at the end.
// Block -> (pair, chunk) is XCD-aware: all chunks of a pair have equal blockIdx % 8, i.e. share an
// XCD and therefore its L2 copy of the pair's tables (speed only, never correctness).
struct LutCell { float edge; int32_t idx; };       // nearest edge (in w / pa units) and its index

struct PointClass { int s; bool inb; float dx, dy, dz; };

// Literal evaluation of one transformed point: c2s, bin, slot look-up, 6-sided bounds test.
__device__ __forceinline__ void classify_exact(float qx, float qy, float qz, const int16_t* map, const float* __restrict__ thr, int T, int P,
                                            const SlotHot* __restrict__ hs, PointClass& out) {
    float r, th, ph;
    c2s_point(qx, qy, qz, r, th, ph);
    const float scale_t = (float)((double)T / kTwoPi), scale_p = (float)((double)P / kPi);
    int bt = bin_from_table(th, thr, T, scale_t);
    int bp = bin_from_table(ph, thr + T + 1, P, scale_p);
    if (bt >= T) bt = static_cast<int>(((double)th / kTwoPi) * (double)T) % T;
    if (bp >= P) bp = static_cast<int>(((double)ph / kPi) * (double)P) % P;
    const int s = map[T * bp + bt];
    out.s = s; out.inb = false; out.dx = out.dy = out.dz = 0.f;
    if (s >= 0) {
        const SlotHot h = hs[s];
        out.inb = inside_bounds(r, th, ph, h.az0, h.az1, h.el0, h.el1, h.inner, h.outer);
        out.dx = qx - h.mu[0]; out.dy = qy - h.mu[1]; out.dz = qz - h.mu[2];
    }
}

// Partial sums of a slot that has no LDS row go straight to HBM.  Kept out of line so that the LDS update above
// stays a ds_add_* (a select between the two pointers would turn both into flat atomics).
// float -> 64-bit fixed point (floor(v * 2^30), two's complement) in 6 VALU instructions: the scaling is exact (power of
// two), h = floor(x / 2^32) is a small integer held exactly in a float, and x - h * 2^32 is exact under fma and lies in
// [0, 2^32).  Any fixed rounding rule would do; what matters is that integer addition is associative.
__device__ __forceinline__ unsigned long long to_fix(float v) {
    const float x = v * kFixScale;
    const float h = floorf(x * 2.3283064365386963e-10f);            // 2^-32
    const float lo = fmaf(h, -4294967296.0f, x);
    return ((unsigned long long)(uint32_t)(int)h << 32) | (unsigned long long)(uint32_t)lo;
}

// Two values at once: the three multiplies / the fma are packed-FP32 instructions (v_pk_mul_f32, v_pk_fma_f32).
typedef float vfloat2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void to_fix2(float a, float b, unsigned long long& fa, unsigned long long& fb) {
    const vfloat2 v = {a, b};
    const vfloat2 x = v * kFixScale;
    vfloat2 h = x * 2.3283064365386963e-10f;                        // 2^-32
    h.x = floorf(h.x); h.y = floorf(h.y);
    const vfloat2 lo = __builtin_elementwise_fma(h, (vfloat2){-4294967296.0f, -4294967296.0f}, x);
    fa = ((unsigned long long)(uint32_t)(int)h.x << 32) | (unsigned long long)(uint32_t)lo.x;
    fb = ((unsigned long long)(uint32_t)(int)h.y << 32) | (unsigned long long)(uint32_t)lo.y;
}

__device__ __noinline__ void spill_flush(uint32_t* A, uint32_t nraw, uint32_t nin, float S0, float S1, float S2, float S3, float S4,
                                         float S5, float S6, float S7, float S8) {
    unsigned long long* F = reinterpret_cast<unsigned long long*>(A + 2);
    atomicAdd(reinterpret_cast<unsigned long long*>(A), (unsigned long long)nraw | ((unsigned long long)nin << 32));   // A[0] raw, A[1] in: one 64-bit add
    if (nin) {
        atomicAdd(&F[0], to_fix(S0)); atomicAdd(&F[1], to_fix(S1)); atomicAdd(&F[2], to_fix(S2)); atomicAdd(&F[3], to_fix(S3)); atomicAdd(&F[4], to_fix(S4));
        atomicAdd(&F[5], to_fix(S5)); atomicAdd(&F[6], to_fix(S6)); atomicAdd(&F[7], to_fix(S7)); atomicAdd(&F[8], to_fix(S8));
    }
}

typedef __attribute__((address_space(1))) const float gfloat;
typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const vfloat4 gfloat4;

template <bool kVec4>
__global__ __launch_bounds__(kAccBlock, kAccWavesPerSimd) void k_gn_accumulate(const PairDesc* __restrict__ desc, const float* __restrict__ xf_all,
                                                          const int16_t* __restrict__ slot_of_voxel, const int32_t* __restrict__ n_slots,
                                                          const SlotHot* __restrict__ hotS, uint32_t* __restrict__ acc,
                                                          const float* __restrict__ thr, const LutCell* __restrict__ lut,
                                                          int T, int P, int Mt, int Mp, float guard_t, float guard_p,
                                                          int lds_slots, int chunks, int n_pairs, int force_exact) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int V = T * P;
    int pair, chunk;
    if (n_pairs >= 8) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        pair = (j / chunks) * 8 + xcd; chunk = j % chunks;
        if (pair >= n_pairs) return;
    } else {
        pair = blockIdx.x / chunks; chunk = blockIdx.x % chunks;
    }
    const PairDesc d = desc[pair];
    int cs = (d.n2 + chunks - 1) / chunks;
    cs = (cs + kAccPts * kAccBlock - 1) / (kAccPts * kAccBlock) * (kAccPts * kAccBlock);   // whole trips of kAccPts points per lane
    const int begin = chunk * cs;
    if (begin >= d.n2) return;
    const int end = min(d.n2, begin + cs);

    LutCell* lut_t = reinterpret_cast<LutCell*>(smem);                    // Mt + 1 cells (the spare one catches pa == 4)
    LutCell* lut_p = lut_t + (Mt + 1);                                    // Mp + 1 cells (w == 1)
    // per slot 10 x u64, the layout of the HBM accumulator: [raw | in << 32], then the 9 fixed-point sums -- constant offsets
    // inside a flush, one 64-bit add for the two counts
    unsigned long long* lacc = reinterpret_cast<unsigned long long*>(lut_p + (Mp + 1));
    float* hot = reinterpret_cast<float*>(lacc + 10 * lds_slots);         // lds_slots x 5: inner, outer, mu1
    int16_t* map = reinterpret_cast<int16_t*>(hot + lds_slots * 5);
    const int map_words = (V + 1) / 2;
    const int ns = n_slots[pair];
    const int nl = min(ns, lds_slots);
    const SlotHot* hs = hotS + (size_t)pair * V;
    {
        const uint32_t* gm = reinterpret_cast<const uint32_t*>(slot_of_voxel + (size_t)pair * ((V + 1) & ~1));   // rows padded to even length
        uint32_t* lm = reinterpret_cast<uint32_t*>(map);
        for (int i = threadIdx.x; i < map_words; i += kAccBlock) lm[i] = gm[i];
        for (int i = 2 * map_words + threadIdx.x; i < V + T + 2; i += kAccBlock) map[i] = (int16_t)-1;   // bt == T or bp == P land here
        const uint2* gl = reinterpret_cast<const uint2*>(lut);
        uint2* ll = reinterpret_cast<uint2*>(lut_t);
        for (int i = threadIdx.x; i < Mt + Mp + 2; i += kAccBlock) ll[i] = gl[i];
        for (int i = threadIdx.x; i < nl * 5; i += kAccBlock) { int s = i / 5, k = i - s * 5; hot[i] = reinterpret_cast<const float*>(hs + s)[4 + k]; }
        for (int i = threadIdx.x; i < 10 * nl; i += kAccBlock) lacc[i] = 0ull;   // only the rows in use
    }
    const float* xf = xf_all + pair * kXf;
    const float tx = xf[0], ty = xf[1], tz = xf[2];
    const float R00 = xf[3], R01 = xf[4], R02 = xf[5], R10 = xf[6], R11 = xf[7], R12 = xf[8], R20 = xf[9], R21 = xf[10], R22 = xf[11];
    const float cell_t = (float)Mt * 0.25f, cell_p = (float)Mp * 0.5f;
    __syncthreads();

    gfloat* px = (gfloat*)d.s2; gfloat* py = px + d.ld2; gfloat* pz = px + 2 * (size_t)d.ld2;   // scans live in HBM: global_load, not flat
    uint32_t* gacc = acc + (size_t)pair * V * kAccWords;

    auto load4 = [&](int i0, float (&X)[4], float (&Y)[4], float (&Z)[4]) {
        if (kVec4 && i0 + 3 < end) {
            // read-once stream: non-temporal, so the scans do not evict the pair tables from L2
            const vfloat4 a = __builtin_nontemporal_load((gfloat4*)(px + i0)), b = __builtin_nontemporal_load((gfloat4*)(py + i0)), c = __builtin_nontemporal_load((gfloat4*)(pz + i0));
            X[0] = a.x; X[1] = a.y; X[2] = a.z; X[3] = a.w; Y[0] = b.x; Y[1] = b.y; Y[2] = b.z; Y[3] = b.w; Z[0] = c.x; Z[1] = c.y; Z[2] = c.z; Z[3] = c.w;
        } else {
#pragma unroll
            // past the end: NaN coordinates fail every guard-band test below, so the point drops out in phase B without a per-point range check
            for (int j = 0; j < 4; j++) { const bool ok = i0 + j < end; X[j] = ok ? px[i0 + j] : __builtin_nanf(""); Y[j] = ok ? py[i0 + j] : 0.f; Z[j] = ok ? pz[i0 + j] : 0.f; }
        }
    };
    // Software pipeline across trips: the loads of trip t+1 are in flight while trip t is classified.  With the scans
    // streaming from HBM (256 distinct pairs = 745 MB, well past the Infinity Cache) a wave that loads, waits and then
    // computes leaves the memory pipe idle: measured 0.18 ms per launch without any prefetch and 0.146 with it; the
    // cache-resident floor (VALU-issue-bound, 16 distinct pairs) is 0.13.  More points per trip would amortise the LDS
    // flushes better but the second set of registers then spills (kAccPts 8: 128 VGPRs + scratch, 0.164 ms).
    // The pipeline runs over GROUPS of 4 points (one dwordx4 per coordinate): while a group is classified the next group of the
    // same lane -- the next 4 of its kAccPts consecutive points, or the first 4 of its next trip -- is already in flight, so the
    // prefetch costs 12 registers however many points a lane takes per trip.
    float XN[4], YN[4], ZN[4];
    load4(begin + kAccPts * (int)threadIdx.x, XN, YN, ZN);
    for (int t0 = begin + kAccPts * threadIdx.x; t0 < begin + cs; t0 += kAccPts * kAccBlock) {   // whole waves iterate together
      // Run state of this lane for the whole trip: the current run, and a stash holding one finished run (see phase C).
      int cur = -1; uint32_t nraw = 0, nin = 0;
      float S0 = 0.f, S1 = 0.f, S2 = 0.f, S3 = 0.f, S4 = 0.f, S5 = 0.f, S6 = 0.f, S7 = 0.f, S8 = 0.f;
      int bs = -1; uint32_t braw = 0, bin = 0;
      float B0 = 0.f, B1 = 0.f, B2 = 0.f, B3 = 0.f, B4 = 0.f, B5 = 0.f, B6 = 0.f, B7 = 0.f, B8 = 0.f;
      auto flush = [&](int slot, uint32_t cr, uint32_t ci, float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float a8) {
          if (slot >= 0) {
              if (slot < nl) {
                  unsigned long long* F = lacc + slot * 10;
                  atomicAdd(&F[0], (unsigned long long)cr | ((unsigned long long)ci << 32));
                  if (ci) {
                      unsigned long long f0, f1, f2, f3, f4, f5, f6, f7;
                      to_fix2(a0, a1, f0, f1); to_fix2(a2, a3, f2, f3); to_fix2(a4, a5, f4, f5); to_fix2(a6, a7, f6, f7);
                      atomicAdd(&F[1], f0); atomicAdd(&F[2], f1); atomicAdd(&F[3], f2); atomicAdd(&F[4], f3); atomicAdd(&F[5], f4);
                      atomicAdd(&F[6], f5); atomicAdd(&F[7], f6); atomicAdd(&F[8], f7); atomicAdd(&F[9], to_fix(a8));
                  }
              } else {
                  spill_flush(gacc + (size_t)slot * kAccWords, cr, ci, a0, a1, a2, a3, a4, a5, a6, a7, a8);
              }
          }
      };
      // The lane's kAccPts CONSECUTIVE points are taken 4 at a time.
#pragma unroll
      for (int g = 0; g < kAccPts / 4; g++) {
        const int i0 = t0 + 4 * g;
        float X[4], Y[4], Z[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { X[j] = XN[j]; Y[j] = YN[j]; Z[j] = ZN[j]; }
        if (g + 1 < kAccPts / 4) load4(i0 + 4, XN, YN, ZN);
        else if (t0 + kAccPts * kAccBlock < begin + cs) load4(t0 + kAccPts * kAccBlock, XN, YN, ZN);
        PointClass pc[4];
        float QX[4], QY[4], QZ[4], RR[4];
        int SM[4];
        bool nr[4];
        // ---- phase A: angular classification of the 4 points.  Straight-line code (bitwise | and &, no clamps: the
        // LUTs carry one spare cell and the map T+1 spare entries, so even a NaN or pa == 4 indexes inside LDS) so
        // that the four look-up chains overlap. ----
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float a = X[j] + tx, b = Y[j] + ty, c = Z[j] + tz;
            const float qx = a * R00 + b * R10 + c * R20;
            const float qy = a * R01 + b * R11 + c * R21;
            const float qz = a * R02 + b * R12 + c * R22;
            QX[j] = qx; QY[j] = qy; QZ[j] = qz;
            const float r2 = qx * qx + qy * qy + qz * qz;
            const float rs = __builtin_amdgcn_rsqf(r2);
            RR[j] = r2 * rs;                                             // |q|
            const float w = -qz * rs;                                    // -cos(phi)
            const float q1 = qy * __builtin_amdgcn_rcpf(fabsf(qx) + fabsf(qy));            // y / (|x| + |y|) in [-1, 1]
            const float pa = (qx >= 0.f) ? ((qy >= 0.f) ? q1 : 4.f + q1) : 2.f - q1;      // diamond angle in [0, 4]
            const LutCell et = lut_t[static_cast<int>(pa * cell_t)];     // NaN converts to 0; pa in [0,4] -> cell in [0, Mt]
            const LutCell ep = lut_p[static_cast<int>((w + 1.f) * cell_p)];
            const int bt = et.idx - ((pa < et.edge) ? 1 : 0);            // in [0, T]
            const int row = ep.idx - ((w < ep.edge) ? T : 0);            // T * polar bin, polar bin in [0, P] (the table stores T * edge index)
            SM[j] = map[row + bt];
            nr[j] = (force_exact != 0) | !(fabsf(pa - et.edge) >= guard_t) | !(fabsf(w - ep.edge) >= guard_p);
        }
        // ---- phase A2: only waves that touch an active voxel look at the hot records (radial test, d = q - mu1) ----
        const bool lane_has = ((SM[0] >= 0) & !nr[0]) | ((SM[1] >= 0) & !nr[1]) | ((SM[2] >= 0) & !nr[2]) | ((SM[3] >= 0) & !nr[3]);
        if (__ballot(lane_has) != 0ull) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int sm = SM[j];
                const bool has = (sm >= 0) & !nr[j];
                // slots beyond the LDS table (more active voxels than lds_slots) are classified by the literal path,
                // which reads its record from HBM: keeps every LDS access a ds_* instruction.
                const float* h = hot + min(max(sm, 0), lds_slots - 1) * 5;
                const float inner = h[0], outer = h[1];
                const float r = RR[j];
                const float gr = 1e-6f * r;
                nr[j] = nr[j] | (has & ((sm >= nl) | !(fabsf(r - inner) >= gr) | !(fabsf(r - outer) >= gr)));
                pc[j].s = nr[j] ? -1 : sm;
                pc[j].inb = has & (r >= inner) & (r <= outer);
                pc[j].dx = QX[j] - h[2]; pc[j].dy = QY[j] - h[3]; pc[j].dz = QZ[j] - h[4];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) { pc[j].s = -1; pc[j].inb = false; pc[j].dx = pc[j].dy = pc[j].dz = 0.f; }
        }
        // ---- phase B (rare): points within a guard band of a voxel edge are re-done with the literal formulas ----
        if (__ballot(nr[0] | nr[1] | nr[2] | nr[3]) != 0ull) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (nr[j] & (i0 + j < end)) classify_exact(QX[j], QY[j], QZ[j], map, thr, T, P, hs, pc[j]);
        }
        const bool any_slot = (pc[0].s >= 0) | (pc[1].s >= 0) | (pc[2].s >= 0) | (pc[3].s >= 0);
        if (__ballot(any_slot | (cur >= 0)) == 0ull) continue;           // wave-uniform: nothing here lands in an active voxel and no run is open
        // ---- phase C: run-length accumulation over the lane's consecutive points.  A finished run is parked in a
        // register stash instead of being flushed at once, so a lane converts to fixed point and touches LDS about
        // twice per trip of kAccPts points (a third run inside one trip, rare, flushes the stash early). ----
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int s = pc[j].s;
            if (s != cur) {
                if (cur >= 0) {
                    // only a lane whose OWN stash is occupied (a third run inside its 4 points: rare) flushes here; a wave-wide
                    // "any lane has a stash" test would run the flush at almost every j, for lanes that could have waited
                    if (bs >= 0) flush(bs, braw, bin, B0, B1, B2, B3, B4, B5, B6, B7, B8);
                    bs = cur; braw = nraw; bin = nin; B0 = S0; B1 = S1; B2 = S2; B3 = S3; B4 = S4; B5 = S5; B6 = S6; B7 = S7; B8 = S8;
                }
                cur = s; nraw = 0; nin = 0; S0 = S1 = S2 = S3 = S4 = S5 = S6 = S7 = S8 = 0.f;
            }
            if (s >= 0) {
                nraw++;
                if (pc[j].inb) {
                    const float dx = pc[j].dx, dy = pc[j].dy, dz = pc[j].dz;
                    nin++;
                    S0 += dx; S1 += dy; S2 += dz;
                    S3 += dx * dx; S4 += dx * dy; S5 += dx * dz; S6 += dy * dy; S7 += dy * dz; S8 += dz * dz;
                }
            }
        }
      }   // sub-groups of 4
      if (__ballot((cur >= 0) | (bs >= 0)) != 0ull) {
          flush(cur, nraw, nin, S0, S1, S2, S3, S4, S5, S6, S7, S8);
          if (__ballot(bs >= 0) != 0ull) flush(bs, braw, bin, B0, B1, B2, B3, B4, B5, B6, B7, B8);
      }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nl; s += kAccBlock) {
        const unsigned long long* L = lacc + s * 10;
        const unsigned long long cnt = L[0];
        if ((uint32_t)cnt == 0u) continue;                                  // no point of this chunk reached the voxel
        unsigned long long* G = reinterpret_cast<unsigned long long*>(gacc + (size_t)s * kAccWords);
        atomicAdd(&G[0], cnt);
        if ((uint32_t)(cnt >> 32)) {
#pragma unroll
            for (int k = 1; k < 10; k++) atomicAdd(&G[k], L[k]);
        }
    }
}

// fitCells2's per-voxel algebra + reduction + the 6x6 solve.  One block per pair.
__global__ __launch_bounds__(kBlock) void k_gn_solve(const int32_t* __restrict__ n_slots, const SlotFit* __restrict__ fitS, uint32_t* __restrict__ acc,
                                                     float* __restrict__ X_all, float* __restrict__ xf_all, float* __restrict__ out, AuxDev aux,
                                                     int V, int n, int iter, int runlen) {
    __shared__ float J[27];
    __shared__ float red[kBlock / 64][27];
    const int pair = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* X = X_all + pair * 6;
    if (threadIdx.x < 27) J[threadIdx.x] = xf_all[pair * kXf + 16 + threadIdx.x];      // written by the previous update (write_xf)
    __syncthreads();
    const int ns = n_slots[pair];
    float S[27];
#pragma unroll
    for (int k = 0; k < 27; k++) S[k] = 0.f;
    for (int s = threadIdx.x; s < ns; s += kBlock) {
        uint32_t* A = acc + ((size_t)pair * V + s) * kAccWords;
        const uint32_t n2 = A[0], m = A[1];
        float sd[3], sdd[6];
        const long long* AF = reinterpret_cast<const long long*>(A + 2);
#pragma unroll
        for (int k = 0; k < 3; k++) sd[k] = (float)((double)AF[k] * kFixInv);
#pragma unroll
        for (int k = 0; k < 6; k++) sdd[k] = (float)((double)AF[3 + k] * kFixInv);
#pragma unroll
        for (int k = 0; k < kAccWords; k++) A[k] = 0u;           // ready for the next iteration
        const SlotFit f = fitS[(size_t)pair * V + s];
        if (aux.n2_raw) aux.n2_raw[((size_t)pair * runlen + iter) * V + f.v] = (int)n2;
        if (aux.n2_in) aux.n2_in[((size_t)pair * runlen + iter) * V + f.v] = (int)m;
        if (!((int)n2 > n && (int)m > n)) continue;               // src/icet.cpp:290 (scan-2 half), :302
        const float fm = (float)m;
        const float db[3] = {sd[0] / fm, sd[1] / fm, sd[2] / fm};   // mean - mu1
        const float mu2[3] = {f.mu[0] + db[0], f.mu[1] + db[1], f.mu[2] + db[2]};
        const float den = (float)(m - 1), d2 = (float)(n2 - 1);
        // R_noise = sigma1/(|idx1|-1) + cov2/(|idx2|-1)                         src/icet.cpp:315
        float Rn[6];
        Rn[0] = f.s1n[0] + ((sdd[0] - fm * db[0] * db[0]) / den) / d2;
        Rn[1] = f.s1n[1] + ((sdd[1] - fm * db[0] * db[1]) / den) / d2;
        Rn[2] = f.s1n[2] + ((sdd[2] - fm * db[0] * db[2]) / den) / d2;
        Rn[3] = f.s1n[3] + ((sdd[3] - fm * db[1] * db[1]) / den) / d2;
        Rn[4] = f.s1n[4] + ((sdd[4] - fm * db[1] * db[2]) / den) / d2;
        Rn[5] = f.s1n[5] + ((sdd[5] - fm * db[2] * db[2]) / den) / d2;
        // Rp = M Rn M^T  (M = L U^T)                                             src/icet.cpp:317
        const float* M = f.M;
        float MR[9];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            MR[3 * i + 0] = M[3 * i] * Rn[0] + M[3 * i + 1] * Rn[1] + M[3 * i + 2] * Rn[2];
            MR[3 * i + 1] = M[3 * i] * Rn[1] + M[3 * i + 1] * Rn[3] + M[3 * i + 2] * Rn[4];
            MR[3 * i + 2] = M[3 * i] * Rn[2] + M[3 * i + 1] * Rn[4] + M[3 * i + 2] * Rn[5];
        }
        float Rp[6];
        Rp[0] = MR[0] * M[0] + MR[1] * M[1] + MR[2] * M[2];
        Rp[1] = MR[0] * M[3] + MR[1] * M[4] + MR[2] * M[5];
        Rp[2] = MR[0] * M[6] + MR[1] * M[7] + MR[2] * M[8];
        Rp[3] = MR[3] * M[3] + MR[4] * M[4] + MR[5] * M[5];
        Rp[4] = MR[3] * M[6] + MR[4] * M[7] + MR[5] * M[8];
        Rp[5] = MR[6] * M[6] + MR[7] * M[7] + MR[8] * M[8];
        float W[6];
        icetdev::pinv3_sym(Rp, 3.0f * FLT_EPSILON, W);                        // src/icet.cpp:320-321
        // H_z = M * [-I | Jx mu | Jy mu | Jz mu]                                  src/icet.cpp:324-329
        float Hj[9];      // columns 3..5 of H_j, row-major 3x3
#pragma unroll
        for (int i = 0; i < 3; i++) {
            Hj[3 * i + 0] = J[3 * i] * mu2[0] + J[3 * i + 1] * mu2[1] + J[3 * i + 2] * mu2[2];
            Hj[3 * i + 1] = J[9 + 3 * i] * mu2[0] + J[9 + 3 * i + 1] * mu2[1] + J[9 + 3 * i + 2] * mu2[2];
            Hj[3 * i + 2] = J[18 + 3 * i] * mu2[0] + J[18 + 3 * i + 1] * mu2[1] + J[18 + 3 * i + 2] * mu2[2];
        }
        float Hz[18];     // 3 x 6 row-major
#pragma unroll
        for (int i = 0; i < 3; i++) {
            Hz[6 * i + 0] = -M[3 * i]; Hz[6 * i + 1] = -M[3 * i + 1]; Hz[6 * i + 2] = -M[3 * i + 2];
#pragma unroll
            for (int j = 0; j < 3; j++) Hz[6 * i + 3 + j] = M[3 * i] * Hj[j] + M[3 * i + 1] * Hj[3 + j] + M[3 * i + 2] * Hj[6 + j];
        }
        float WH[18];     // W * Hz
#pragma unroll
        for (int j = 0; j < 6; j++) {
            WH[j]      = W[0] * Hz[j] + W[1] * Hz[6 + j] + W[2] * Hz[12 + j];
            WH[6 + j]  = W[1] * Hz[j] + W[3] * Hz[6 + j] + W[4] * Hz[12 + j];
            WH[12 + j] = W[2] * Hz[j] + W[4] * Hz[6 + j] + W[5] * Hz[12 + j];
        }
        // dz = M (mu2 - mu1)                                                       src/icet.cpp:335-337
        float dz[3];
#pragma unroll
        for (int i = 0; i < 3; i++) dz[i] = M[3 * i] * db[0] + M[3 * i + 1] * db[1] + M[3 * i + 2] * db[2];
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = a; b < 6; b++) { S[q] += Hz[a] * WH[b] + Hz[6 + a] * WH[6 + b] + Hz[12 + a] * WH[12 + b]; q++; }
        }
#pragma unroll
        for (int a = 0; a < 6; a++) S[21 + a] += WH[a] * dz[0] + WH[6 + a] * dz[1] + WH[12 + a] * dz[2];
    }
#pragma unroll
    for (int k = 0; k < 27; k++) { float t = wave_sum(S[k]); if (lane == 0) red[wave][k] = t; }
    __syncthreads();
    if (threadIdx.x != 0) return;

    float Hm[36], g[6];
    {
        int q = 0;
        for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) {
            float t = 0.f; for (int w = 0; w < kBlock / 64; w++) t += red[w][q];
            Hm[a * 6 + b] = t; Hm[b * 6 + a] = t; q++;
        }
        for (int a = 0; a < 6; a++) { float t = 0.f; for (int w = 0; w < kBlock / 64; w++) t += red[w][21 + a]; g[a] = t; }
    }
    // Normal case first: HTWH positive definite with condition number <= 1e6.  Then nothing is pruned
    // (checkCondition's cutoff, src/icet.cpp:453,469), every eigenvalue is above the pseudo-inverse's rank threshold
    // (eps * 6 < 1e-6), pinv(HTWH) is the inverse and dx = HTWH^-1 HTWdz -- a Cholesky factorisation gives both.
    // cond_2 <= |A|_F |A^-1|_F, so that product <= 1e6 PROVES the case without an eigen-solve.  When the bound is
    // inconclusive the eigenvalues decide, exactly as the reference does, and only a genuinely ill-conditioned or
    // rank-deficient HTWH takes the eigenvector route with its pruning.
    float ev[6];
    float cov[36], ps[6], dx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool plain = icetdev::chol6_inverse(Hm, cov);
    if (plain) {
        float fa = 0.f, fi = 0.f;
        for (int k = 0; k < 36; k++) { fa += Hm[k] * Hm[k]; fi += cov[k] * cov[k]; }
        if (!(fa * fi <= 1e12f)) {                                  // bound inconclusive (or NaN): ask the eigenvalues
            icetdev::eig6_sym<false>(Hm, ev, nullptr);
            float emax = 0.f;
            for (int k = 0; k < 6; k++) emax = fmaxf(emax, fabsf(ev[k]));
            plain = !(fabsf(ev[5] / ev[0]) > 1e6f);
            for (int k = 0; k < 6; k++) plain = plain && (fabsf(ev[k]) > 6.0f * FLT_EPSILON * emax);
        }
    }
    if (plain) {
        for (int k = 0; k < 6; k++) ps[k] = sqrtf(fabsf(cov[k * 6 + k]));         // src/icet.cpp:412-417
        for (int a = 0; a < 6; a++) { float t = 0.f; for (int b = 0; b < 6; b++) t += cov[a * 6 + b] * g[b]; dx[a] = t; }
    } else {
        float Q[36];
        icetdev::eig6_sym<true>(Hm, ev, Q);
        float emax = 0.f;
        for (int k = 0; k < 6; k++) emax = fmaxf(emax, fabsf(ev[k]));
        const float rthr = 6.0f * FLT_EPSILON * emax;              // rank rule eps*6 relative to the largest eigenvalue
        // noise_mat = pinv(HTWH) (src/icet.cpp:410-411)
        float inv[6];
        for (int k = 0; k < 6; k++) inv[k] = (fabsf(ev[k]) > rthr) ? 1.f / ev[k] : 0.f;
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
            float t = 0.f; for (int k = 0; k < 6; k++) t += Q[a * 6 + k] * inv[k] * Q[b * 6 + k];
            cov[a * 6 + b] = t;
        }
        for (int k = 0; k < 6; k++) ps[k] = sqrtf(fabsf(cov[k * 6 + k]));         // src/icet.cpp:412-417
        // checkCondition (src/icet.cpp:443-492)
        int k0 = 0;
        {
            float condition = ev[5] / ev[0];
            int eyecount = 1;
            while (fabsf(condition) > 1e6f && eyecount < 6) {
                for (int k = 0; k < 6; k++) ps[k] += Q[k * 6 + eyecount - 1];       // src/icet.cpp:479
                k0++;
                condition = ev[5] / ev[eyecount];
                eyecount++;
            }
        }
        // dx = pinv(L2 lam U2^T) L2 U2^T HTWdz  = sum_{k >= k0} q_k (q_k . g) / lam_k     src/icet.cpp:427-430
        for (int k = k0; k < 6; k++) {
            if (inv[k] == 0.f) continue;
            float pj = 0.f; for (int a = 0; a < 6; a++) pj += Q[a * 6 + k] * g[a];
            pj *= inv[k];
            for (int a = 0; a < 6; a++) dx[a] += Q[a * 6 + k] * pj;
        }
    }
    float Xn[6];
    for (int k = 0; k < 6; k++) { Xn[k] = X[k] + dx[k]; X[k] = Xn[k]; }
    write_xf(xf_all + pair * kXf, Xn);
    float* o = out + (size_t)pair * 48;
    for (int k = 0; k < 6; k++) { o[k] = Xn[k]; o[6 + k] = ps[k]; }
    for (int k = 0; k < 36; k++) o[12 + k] = cov[k];
    if (aux.x_hist) for (int k = 0; k < 6; k++) aux.x_hist[((size_t)pair * runlen + iter) * 6 + k] = Xn[k];
    if (aux.htwh) for (int k = 0; k < 36; k++) aux.htwh[((size_t)pair * runlen + iter) * 36 + k] = Hm[k];
    if (aux.htwdz) for (int k = 0; k < 6; k++) aux.htwdz[((size_t)pair * runlen + iter) * 6 + k] = g[k];
}

inline int chunks_for(int n_pairs, int max_n, int per_block_min, int target_blocks) {
    int by_work = (max_n + per_block_min - 1) / per_block_min;
    int want = (target_blocks + n_pairs - 1) / n_pairs;
    int c = want < by_work ? want : by_work;
    return c < 1 ? 1 : c;
}

}  // namespace

#define ICET_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t launch_keyframe(const Workspace& w, const LaunchCfg& c, const AuxDev* auxp, hipStream_t st) {
    AuxDev aux{}; if (auxp) aux = *auxp;
    // many small chunks per pair: with the XCD-aware mapping only ~4 pairs are then in flight per XCD (see decode_block)
    const int chunks = c.kf_chunks;
    const int groups = c.n_pairs >= 8 ? (c.n_pairs + 7) / 8 * 8 : c.n_pairs;
    dim3 grid(groups * chunks), blk(kBlock);
    const int np = c.n_pairs;
    hipError_t e;
    int pbits = 0; while ((1 << pbits) < c.n_pairs) pbits++;
    int vbits = 1; while ((1 << vbits) < c.V) vbits++;
    const bool batch = c.n_pairs > 1;
    if (!c.use_library_sort) { e = launch_rank_sort_splitters(w, c, st); if (e != hipSuccess) return e; }
    k_scan1_spherical<<<grid, blk, 0, st>>>(w.desc, w.r1, w.th1, w.ph1, (batch && c.use_library_sort) ? w.key64A : nullptr, c.use_library_sort ? w.keyA : nullptr, w.valA, w.bin16, c.T, c.P, np, chunks,
                                            c.use_library_sort ? nullptr : w.splitters, w.bkt, w.counts);
    ICET_LAUNCH_CHECK();
    if (c.stage_event && c.stage_at == 4) { e = hipEventRecord(c.stage_event, st); if (e != hipSuccess) return e; }
    if (c.use_library_sort) {
        if (batch) e = sort_pairs_u64(w.sort_tmp, w.sort_tmp_bytes, w.key64A, w.key64B, w.valA, w.valB, c.total_n1, 32 + pbits, st);
        else e = sort_pairs_u32(w.sort_tmp, w.sort_tmp_bytes, w.keyA, w.keyB, w.valA, w.valB, c.total_n1, 32, st);
        if (e != hipSuccess) return e;
        // valB = s : original index of the row with rank i
        k_inverse_perm<<<grid, blk, 0, st>>>(w.desc, w.valB, w.pred, np, chunks);
        ICET_LAUNCH_CHECK();
    } else {
        e = launch_rank_sort(w, c, st);        // valB = s, pred = s^-1  (icet_ranksort.hip)
        if (e != hipSuccess) return e;
    }
    if (c.stage_event && c.stage_at == 1) { e = hipEventRecord(c.stage_event, st); if (e != hipSuccess) return e; }
    e = hipMemsetAsync(w.flags, 0, sizeof(int32_t) * c.n_pairs, st); if (e != hipSuccess) return e;
    if (c.true_sort) {
        // non-parity extension: the rows stay in sorted order (what the reference's comment says the loop is meant to do)
        e = hipMemcpyAsync(w.src, w.valB, sizeof(int32_t) * (size_t)c.total_n1, hipMemcpyDeviceToDevice, st); if (e != hipSuccess) return e;
        k_bin_hist<<<grid, blk, (size_t)c.V * 4, st>>>(w.desc, w.src, w.bin16, w.binpos, w.counts, w.flags, c.V, np, chunks, 1);
        ICET_LAUNCH_CHECK();
    } else {
    const int max_walk = 4096;
    k_exec_flags<<<grid, blk, 0, st>>>(w.desc, w.valB, w.pred, w.bin16, w.flags, max_walk, np, chunks);
    ICET_LAUNCH_CHECK();
    k_scramble_src<<<grid, blk, (size_t)c.V * 4, st>>>(w.desc, w.valB, w.pred, w.src, w.flags, max_walk, w.bin16, w.binpos, w.counts, c.V, np, chunks);
    ICET_LAUNCH_CHECK();
    k_scramble_serial<<<c.n_pairs, 64, 0, st>>>(w.desc, w.valB, w.pred /* reused as scratch */, w.src, w.flags);
    ICET_LAUNCH_CHECK();
    if (c.stage_event && c.stage_at == 2) { e = hipEventRecord(c.stage_event, st); if (e != hipSuccess) return e; }
    k_bin_hist<<<grid, blk, (size_t)c.V * 4, st>>>(w.desc, w.src, w.bin16, w.binpos, w.counts, w.flags, c.V, np, chunks, 0);
    ICET_LAUNCH_CHECK();
    }
    e = launch_class_scan(w.counts, w.tile_base, w.bin_start, c.V, chunks, c.n_pairs, st);
    if (e != hipSuccess) return e;
    k_bin_scatter<<<grid, blk, (size_t)c.V * 16, st>>>(w.desc, w.src, w.binpos, w.tile_base, w.bin_start, w.valA, c.V, vbits, np, chunks);
    ICET_LAUNCH_CHECK();
    if (c.stage_event && c.stage_at == 3) { e = hipEventRecord(c.stage_event, st); if (e != hipSuccess) return e; }
    dim3 gfit((c.V + kBlock / 64 - 1) / (kBlock / 64), c.n_pairs);
    k_fit_scan1<<<gfit, blk, 0, st>>>(w.desc, w.bin_start, w.valA, w.r1, w.th1, w.ph1, w.midD, c.T, c.P, c.n, c.thresh, c.buff);
    ICET_LAUNCH_CHECK();
    k_fit_finish<<<dim3((c.V + kBlock - 1) / kBlock, c.n_pairs), blk, 0, st>>>(w.midD, w.hotD, w.fitD, w.activeD, aux, c.T, c.P, c.n);
    ICET_LAUNCH_CHECK();
    k_compact_slots<<<c.n_pairs, blk, 0, st>>>(w.hotD, w.fitD, w.activeD, w.hotS, w.fitS, w.slot_of_voxel, w.n_slots, w.acc, c.V);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_class_scan(const uint32_t* counts, uint32_t* tile_base, int32_t* class_start, int n_classes, int chunks, int n_pairs, hipStream_t st) {
    k_bin_tiles<<<dim3((n_classes + kBlock - 1) / kBlock, n_pairs), kBlock, 0, st>>>(counts, tile_base, class_start, n_classes, chunks);
    ICET_LAUNCH_CHECK();
    k_bin_scan<<<n_pairs, kBlock, 0, st>>>(class_start, n_classes);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_init_state(const Workspace& w, const LaunchCfg& c, const float* d_x0, hipStream_t st) {
    k_init_state<<<(c.n_pairs + 63) / 64, 64, 0, st>>>(d_x0, w.X, w.xf, c.n_pairs);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_gn_accumulate(const Workspace& w, const LaunchCfg& c, hipStream_t st) {
    // LDS rows for active voxels: a throughput batch keeps 320 rows (measured optimum on 64-channel scans: fewer rows
    // spill busy voxels to HBM atomics, more rows cost occupancy); a small batch has
    // CUs to spare, so a block may take most of a CU's LDS and keep every active voxel of a fine grid (150 x 48: often
    // > 1000) out of the slow HBM-atomic path.
    const size_t fixed = (size_t)(w.lut_Mt + w.lut_Mp + 2) * sizeof(LutCell) + (size_t)((c.V + c.T + 4) / 2) * 4 + 16;
    const size_t row = (5 + kAccLds) * 4;
    const size_t budget = (c.n_pairs >= 32) ? fixed + 320 * row : 144 * 1024;   // 320 rows: ~46 KB/block for 75 x 24, three blocks per CU
    int lds_slots = c.lds_slots > 0 ? c.lds_slots : (int)((budget > fixed ? budget - fixed : 0) / row);
    lds_slots = lds_slots < 32 ? 32 : lds_slots;
    if (lds_slots > c.V) lds_slots = c.V;
    const int chunks = chunks_for(c.n_pairs, c.max_n2, kAccBlock * c.acc_min_pts_per_thread, c.acc_target_blocks);
    const size_t lds = fixed + (size_t)lds_slots * (5 + kAccLds) * 4;
    static bool attr_done[64] = {};          // per device: a process may hold contexts on several GPUs
    int dev_ = 0; (void)hipGetDevice(&dev_);
    bool& attr_set = attr_done[dev_ & 63];
    if (!attr_set) {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gn_accumulate<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea == hipSuccess) ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gn_accumulate<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea != hipSuccess) return ea;
        attr_set = true;
    }
    const int groups = c.n_pairs >= 8 ? (c.n_pairs + 7) / 8 * 8 : c.n_pairs;
    dim3 grid(groups * chunks), blk(kAccBlock);
    const LutCell* lut = reinterpret_cast<const LutCell*>(w.lut);
    if (c.vec4_ok)
        k_gn_accumulate<true><<<grid, blk, lds, st>>>(w.desc, w.xf, w.slot_of_voxel, w.n_slots, w.hotS, w.acc, w.thr, lut, c.T, c.P, w.lut_Mt, w.lut_Mp,
                                                     w.guard_t, w.guard_p, lds_slots, chunks, c.n_pairs, c.force_exact);
    else
        k_gn_accumulate<false><<<grid, blk, lds, st>>>(w.desc, w.xf, w.slot_of_voxel, w.n_slots, w.hotS, w.acc, w.thr, lut, c.T, c.P, w.lut_Mt, w.lut_Mp,
                                                      w.guard_t, w.guard_p, lds_slots, chunks, c.n_pairs, c.force_exact);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_gn_solve(const Workspace& w, const LaunchCfg& c, int iter, float* d_out, const AuxDev* auxp, hipStream_t st) {
    AuxDev aux{}; if (auxp) aux = *auxp;
    k_gn_solve<<<c.n_pairs, kBlock, 0, st>>>(w.n_slots, w.fitS, w.acc, w.X, w.xf, d_out, aux, c.V, c.n, iter, c.runlen);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace icet
