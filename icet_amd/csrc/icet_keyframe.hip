// icet_amd/csrc/icet_keyframe.hip -- keyframe build: everything the reference does ONCE per pair on scan 1.
//
// Reference path (all under /root/reference): ICET::fitScan1 src/icet.cpp:68-107 and fitCells1 :109-252.  The device
// formulation is NOT a translation:
//     k_scan1_spherical   utils::cartesianToSpherical (src/utils.cpp:93-119): r of every row (bit-exact), its voxel
//                         (sortSphericalCoordinates, src/icet.cpp:534-554) through the fast classification with the literal
//                         fallback, its rank-sort bucket and the tile's bucket histogram
//     [rank sort, icet_ranksort.hip]                      std::sort by r (src/icet.cpp:72-77)
//     k_exec_flags / k_scramble_src (+ k_scramble_replay) the reference's one-step swap loop (src/icet.cpp:78-83) in
//                                                          parallel closed form
//     k_bin_tiles + k_bin_scan + k_bin_scatter             stable multi-split of positions by voxel = the order in which
//                                                          sortSphericalCoordinates appends rows to each voxel
//     k_fit_cluster / k_fit_roundtrip / k_fit_moments   fitCells1: findCluster (:557-607), bounds filter (:609-652), the
//                         spherical -> Cartesian round trip of the surviving rows (:159), their mean / covariance (:160-162)
//     k_fit_finish        per-bin tail, one lane per bin: 3x3 eigen-decomposition (:181-184), sigma points (:187-232) -> L,
//                         records of the active voxels
//     k_compact_slots     dense voxel table -> compact "slots" of active voxels
// theta / phi of a row are never stored: only decisions (which voxel, inside the bounds) and the Gaussians need them, and
// those evaluate them where needed under the shared arithmetic rule (icet_device_common.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include "icet_internal.h"
#include "icet_device_common.h"
#include "icet_device_math.h"
#include <algorithm>

namespace icet {
namespace {

#define ICET_FOR_CHUNK_OF_SCAN1(i)                                                         \
    int pair, chunk;                                                                       \
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;                               \
    const PairDesc d = desc[pair];                                                         \
    int cs_ = (d.n1 + chunks - 1) / chunks; cs_ = (cs_ + kBlock - 1) / kBlock * kBlock;     \
    const int lo_ = chunk * cs_, hi_ = min(d.n1, lo_ + cs_);                               \
    for (int i = lo_ + threadIdx.x; i < hi_; i += kBlock)

// ------------------------------------------------------------------------------------------------
// cartesianToSpherical + sortSphericalCoordinates of scan 1, decisions only
// ------------------------------------------------------------------------------------------------
// The literal formulas for a row within a guard band of a voxel edge (~0.02 % of the rows, plus zero / NaN rows).  Out of line:
// the double-precision atan2 / acos would otherwise set the register budget of the streaming loop.
__device__ __noinline__ int voxel_literal(float px, float py, float pz, float r_raw, int T, int P) {
    return voxel_of(theta_cr(py, px), phi_cr(pz, r_raw), T, P);
}

__device__ __forceinline__ int zero_voxel_of(int patt, const int4& zv) { return patt == 0 ? zv.x : (patt == 1 ? zv.y : (patt == 2 ? zv.z : zv.w)); }

__global__ __launch_bounds__(kBlock) void k_scan1_spherical(const PairDesc* __restrict__ desc, float* __restrict__ r1,
                                                            unsigned long long* __restrict__ key64, uint32_t* __restrict__ key32,
                                                            uint32_t* __restrict__ val, uint16_t* __restrict__ bin16, int T, int P, int n_pairs, int chunks,
                                                            const uint32_t* __restrict__ splitters, uint8_t* __restrict__ bkt, uint32_t* __restrict__ counts,
                                                            const LutCell* __restrict__ lut, int Mt, int Mp, float guard_t, float guard_p, int32_t* __restrict__ tile_vr, int4 zv, int32_t* __restrict__ zero_rows) {
    // First step of the rank sort fused in (icet_ranksort.hip): the pair's splitters are already known (k_rs_splitters
    // samples the radii straight from the Cartesian rows), so each row's bucket and this tile's bucket histogram cost no
    // extra pass over r1[].  splitters == nullptr: library-sort diagnostic path, nothing of this is needed.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s1[];
    LutCell* lut_t = reinterpret_cast<LutCell*>(smem_s1);               // Mt + 1 cells (the spare one catches pa == 4)
    LutCell* lut_p = lut_t + (Mt + 1);                                  // Mp + 1 cells (w == 1)
    __shared__ uint32_t sp[kRankSortMaxBuckets];
    __shared__ uint32_t lh[kRankSortMaxBuckets];
    __shared__ uint32_t s_zc[4];                                        // this tile's exact-zero rows per sign pattern of (y, x)
    {
        if (threadIdx.x < 4) s_zc[threadIdx.x] = 0u;
        const uint2* gl = reinterpret_cast<const uint2*>(lut);
        uint2* ll = reinterpret_cast<uint2*>(lut_t);
        for (int i = threadIdx.x; i < Mt + Mp + 2; i += kBlock) ll[i] = gl[i];
        int pair_, chunk_;
        if (splitters && decode_block(n_pairs, chunks, pair_, chunk_))
            for (int j = threadIdx.x; j < kRankSortMaxBuckets; j += kBlock) { sp[j] = splitters[(size_t)pair_ * kRankSortMaxBuckets + j]; lh[j] = 0u; }
        __syncthreads();
    }
    const float cell_t = (float)Mt * 0.25f, cell_p = (float)Mp * 0.5f;
    // the voxel of an exact-zero row: r = 0, phi = acos(NaN) -> 1000, theta = atan2(+-0, +-0) -- a function of the two sign bits, tabulated by the host (zv)
    // (x = y = +-0 exactly -- not a tiny vector whose squares underflow: its theta is a real angle; z does not matter: z / 0 is NaN or +-inf, acos of either NaN).  A real scan holds
    // thousands of such rows (invalid returns; 18 % of the reference's sample_pc scans): sent through voxel_literal one by one they put the
    // double-precision atan2 / acos into nearly every wave (205 us per 256 real pairs against 123 on synthetic scans)
    int vlo = 0x7FFFFFFF, vhi = -1;                                   // voxel ids this thread has seen
    ICET_FOR_CHUNK_OF_SCAN1(i) {
        const float* x = d.s1; const float* y = d.s1 + d.ld1; const float* z = d.s1 + 2 * (size_t)d.ld1;
        const float px = x[i], py = y[i], pz = z[i];
        const float rr = radius_raw(px, py, pz);
        // which voxel: fast classification, literal formulas within a guard band of an edge (and for anything that is not an
        // ordinary number: zero rows, NaN, inf, magnitudes whose square leaves the float range)
        const float r2 = px * px + py * py + pz * pz;
        const bool ordinary = (r2 >= kR2Min) & (r2 <= kR2Max);
        int bt, prow; bool near;
        classify_angular_fast(ordinary ? px : 1.f, ordinary ? py : 0.f, ordinary ? pz : 0.f, ordinary ? __builtin_amdgcn_rsqf(r2) : 1.f,
                              lut_t, lut_p, cell_t, cell_p, T, guard_t, guard_p, bt, prow, near);
        near = near | !ordinary;
        int v = prow + bt;
        if (near) {
            const bool zrow = (rr == 0.f) & (px == 0.f) & (py == 0.f);
            const int patt = (__builtin_signbit(py) ? 2 : 0) | (__builtin_signbit(px) ? 1 : 0);
            v = zrow ? zero_voxel_of(patt, zv) : voxel_literal(px, py, pz, rr, T, P);
            if (zrow) atomicAdd(&s_zc[patt], 1u);                      // (an LDS counter, not a register carried through the loop: the kernel sits at 64 VGPRs = 8 waves per SIMD)
        }
        const float r = (rr != rr) ? 1000.0f : rr;                      // src/utils.cpp:116
        size_t o = (size_t)d.off1 + i;
        r1[o] = r;
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
        bin16[o] = (uint16_t)v | (near ? kRowNearBit : (uint16_t)0) | (r == 0.f ? kRowZeroBit : (uint16_t)0);
        vlo = min(vlo, v); vhi = max(vhi, v);
    }
    {   // the tile's populated voxel range (a 64-channel scan fills ~1/3 of the id range of the 75 x 24 grid); no atomics: thousands of
        // waves per pair on one address cost this kernel 150 us per 256 pairs -- the rank sort's per-pair scan reduces the tiles' values
        __shared__ int s_lo[kBlock / 64], s_hi[kBlock / 64];
        vlo = wave_reduce_min_i(vlo); vhi = wave_reduce_max_i(vhi);
        if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = vlo; s_hi[threadIdx.x >> 6] = vhi; }

        __syncthreads();
        if (zero_rows && threadIdx.x >= 64 && threadIdx.x < 68 && s_zc[threadIdx.x - 64] != 0u) atomicAdd(&zero_rows[4 * pair + (threadIdx.x - 64)], (int)s_zc[threadIdx.x - 64]);
        if (threadIdx.x == 0) {
            int lo = s_lo[0], hi = s_hi[0];
            for (int k = 1; k < kBlock / 64; k++) { lo = min(lo, s_lo[k]); hi = max(hi, s_hi[k]); }
            tile_vr[((size_t)pair * chunks + chunk) * 2] = lo; tile_vr[((size_t)pair * chunks + chunk) * 2 + 1] = hi;
        }
    }
    if (splitters) {
        __syncthreads();
        for (int j = threadIdx.x; j < kRankSortMaxBuckets; j += kBlock) counts[((size_t)pair * chunks + chunk) * kRankSortMaxBuckets + j] = lh[j];
    }
}

#ifdef ICET_DIAG_LIBSORT
// pred[s[i]] = i : rank of every original row (the library sort's inverse-permutation pass; the rank sort writes pred[] itself).
__global__ __launch_bounds__(kBlock) void k_inverse_perm(const PairDesc* __restrict__ desc, const uint32_t* __restrict__ s, int32_t* __restrict__ pred,
                                                         int n_pairs, int chunks) {
    ICET_FOR_CHUNK_OF_SCAN1(i) pred[(size_t)d.off1 + s[(size_t)d.off1 + i]] = i;
}
#endif

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

// The flags go into a bit table, 64 rows per word: 116 k rows = 15 KB, which k_scramble_src keeps in LDS, so that "did step u
// execute" costs its walks no memory access (round 2a kept the flag in the row's voxel word: one more random read per chain step).
// A pair's words start at exec_word_base; word ranges of different pairs never overlap:
// floor((off1 + n) / 64) + 1 >= floor(off1 / 64) + ceil(n / 64).
constexpr uint16_t kBinMask = kRowBinMask;
__device__ __host__ __forceinline__ size_t exec_word_base(int32_t off1, int pair) { return (size_t)(off1 >> 6) + (size_t)pair; }

__global__ __launch_bounds__(kBlock) void k_exec_flags(const PairDesc* __restrict__ desc, const int32_t* __restrict__ pred,
                                                       unsigned long long* __restrict__ execbits, int32_t* __restrict__ flags, int max_walk, int n_pairs, int chunks) {
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;
    const PairDesc d = desc[pair];
    int cs_ = (d.n1 + chunks - 1) / chunks; cs_ = (cs_ + kBlock - 1) / kBlock * kBlock;
    const int lo_ = chunk * cs_, hi_ = min(d.n1, lo_ + cs_);
    const size_t o = d.off1;
    unsigned long long* bits = execbits + exec_word_base(d.off1, pair);
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
            // a wave holds 64 consecutive rows starting at a multiple of 64 (lo_ is a multiple of kBlock): one ballot = one word
            const int v = base + k * kBlock;
            const unsigned long long m = __ballot(v < hi_ && moved[k] && !(len[k] & 1));
            if ((threadIdx.x & 63) == 0 && v < hi_) bits[v >> 6] = m;
        }
    }
}

// The same bits from the RECURRENCE instead of the chain walks: step v executes iff pred(v) > v, or pred(v) < v and step pred(v) did
// not execute (the chain parity of k_exec_flags, one link at a time; pred(v) < v is never a fixed point, so its bit is its parity).
// Every row then needs ONE earlier bit instead of a walk through pred[] -- 0.7 random 4-byte reads per row, with the long chains
// walked by a few lanes per wave -- if the rows are taken in index order, and a pair's whole bit table is 15 KB: one block per pair
// keeps it in LDS and goes through the pair's rows in chunks of 1024 x kR, pred[] streamed (the next chunk in flight), the bit of
// pred(v) read from LDS.  A row whose pred(v) lies inside its own chunk (a few % of the rows) waits for that row's result: a state
// byte per row of the chunk and a few rounds behind a barrier, until no row of the block is pending.  No global random access at
// all; a pair takes ~60 chunks x a few barriers whatever the batch, so this is the kernel of THROUGHPUT batches (one block per CU
// and pair), while a small batch keeps k_exec_flags and its hundreds of independent tiles.  Measured per 256 pairs: 144 -> see DESIGN.
#ifndef ICET_EXEC_PAIR_ROWS
#define ICET_EXEC_PAIR_ROWS 8
#endif
#ifndef ICET_EXEC_PAIR_THREADS
#define ICET_EXEC_PAIR_THREADS 1024
#endif
constexpr int kExecPairThreads = ICET_EXEC_PAIR_THREADS;
template <int kR>
__global__ __launch_bounds__(kExecPairThreads) void k_exec_flags_pair(const PairDesc* __restrict__ desc, const int32_t* __restrict__ pred,
                                                                      unsigned long long* __restrict__ execbits) {
    extern __shared__ __attribute__((aligned(8))) unsigned char sm_exec[];
    constexpr int kT = kExecPairThreads, kC = kT * kR;
    unsigned char* st = sm_exec;                                          // per row of the chunk: 0 pending, 2 | bit resolved
    unsigned long long* lbits = reinterpret_cast<unsigned long long*>(sm_exec + kC);
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const PairDesc d = desc[pair];
    const int n = d.n1;
    const int32_t* pp = pred + (size_t)d.off1;
    int pn[kR];
#pragma unroll
    for (int r = 0; r < kR; r++) { const int v = r * kT + tid; pn[r] = (v < n) ? pp[v] : 0; }
    // "does any row of the block still wait": one flag per wave, ONE barrier, two sets of flags used in turn (a wave can only write set A again
    // after a barrier that every wave reaches after its reads of set A).  __syncthreads_or (ockl's work-group reduction) cost 0.9 us per
    // round -- the rounds were half of this kernel.
    __shared__ uint32_t s_any[2][kT / 64];
    int or_set = 0;
    auto block_or = [&](bool x) -> bool {
        const bool wave_any = __ballot(x) != 0ull;
        if (lane == 0) s_any[or_set][tid >> 6] = wave_any ? 1u : 0u;
        __syncthreads();
        uint32_t o = 0u;
#pragma unroll
        for (int k = 0; k < kT / 64; k++) o |= s_any[or_set][k];
        or_set ^= 1;
        return o != 0u;
    };
    for (int c0 = 0; c0 < n; c0 += kC) {
        int p[kR];
#pragma unroll
        for (int r = 0; r < kR; r++) { p[r] = pn[r]; const int v = c0 + kC + r * kT + tid; pn[r] = (v < n) ? pp[v] : 0; }      // the next chunk is in flight
        // straight-line (selects, no branch per row): the eight table reads leave together; behind `if (earlier chunk)` each one waited inside
        // its row's branch.  pv is a row of the pair (0 past the end), so the read is always inside the table; its word is only USED for an earlier chunk.
        unsigned long long w[kR];
#pragma unroll
        for (int r = 0; r < kR; r++) w[r] = lbits[p[r] >> 6];
        uint32_t pm = 0u, em = 0u;                                         // per row of the thread: still waiting / step executed
#pragma unroll
        for (int r = 0; r < kR; r++) {
            const int v = c0 + r * kT + tid, pv = p[r];
            const bool valid = v < n, later = pv >= v, early = pv < c0;
            const bool bit = (w[r] >> (pv & 63)) & 1ull;
            // moved to a later position: executes (a fixed point does not); an earlier chunk: its bit is in the table; else a row of this chunk
            const bool ex = valid & (later ? (pv != v) : (early & !bit));
            const bool pd = valid & !later & !early;
            em |= (ex ? 1u : 0u) << r; pm |= (pd ? 1u : 0u) << r;
            st[r * kT + tid] = pd ? (unsigned char)0 : (unsigned char)(2 | (ex ? 1 : 0));
        }
        // rounds: a pending row takes its result as soon as pred(v) has one; every thread looks after its own rows (on lidar data ~3 % of a
        // chunk is pending and three rounds settle it; a chain of k rows inside one chunk -- adversarial input -- takes k rounds).  Two
        // alternatives measured slower: polling the state bytes without barriers (the waves serialise on their rows), and one wave
        // resolving a compacted list of the pending rows (its dependent LDS reads cost more than the barriers of the parallel rounds).
        // A round is paid in VALU issue by all 16 waves of the block (4 per SIMD), so it only touches the rows that wait: a thread takes its
        // waiting rows one at a time (most threads have none, few more than one) instead of walking all eight.
        while (block_or(pm != 0u)) {                                       // (the barrier also publishes the state bytes)
            uint32_t todo = pm;
            while (__ballot(todo != 0u) != 0ull) {                         // wave-uniform
                const bool have = todo != 0u;
                const int r = have ? __builtin_ctz(todo) : 0;
                todo &= todo - 1u;
                int pr = p[0];
#pragma unroll
                for (int q = 1; q < kR; q++) pr = (r == q) ? p[q] : pr;
                const unsigned char sb = st[have ? pr - c0 : 0];           // may be written in this very round: then the row resolves now or in the next round
                const bool got = have & ((sb & 2) != 0);
                if (got) {
                    const bool ex = (sb & 1) == 0;
                    em |= (ex ? 1u : 0u) << r; pm &= ~(1u << r);
                    st[r * kT + tid] = (unsigned char)(2 | (ex ? 1 : 0));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kR; r++) {                                    // a wave holds 64 consecutive rows starting at a multiple of 64: one ballot = one word
            const int v = c0 + r * kT + tid;
            const unsigned long long m = __ballot((em >> r) & 1u);
            if (lane == 0 && v < n) lbits[v >> 6] = m;
        }
        __syncthreads();                                                   // the table is complete up to this chunk; the state bytes are free again
    }
    unsigned long long* gbits = execbits + exec_word_base(d.off1, pair);
    const int nw = (n + 63) >> 6;
    for (int i = tid; i < nw; i += kT) gbits[i] = lbits[i];
}

// src[v] = original row that ends at position v after the swap loop.  Position v receives row
// pred(v), except at the head of a run of executed steps, where the row arrives from the end of
// the forward chain v -> s[v] -> s[s[v]] ... while the steps executed.
// Fused with the first step of the voxel multi-split (k_bin_hist): the row that lands on a position is known here, so
// its voxel id and this tile's voxel histogram cost no extra pass over src[].
#ifndef ICET_SCR_WAVES
#define ICET_SCR_WAVES 8      /* <= 64 VGPRs: 8 waves per SIMD for a latency-bound walk (measured per 256 pairs: 6 waves 269 us, 7: 268, 8: 256) */
#endif
// kBitsInLds: the pair's exec table sits in LDS behind the voxel histogram (any scan below ~0.75 M rows); otherwise its words are
// read from memory.
template <bool kBitsInLds>
__global__ __launch_bounds__(kBlock, ICET_SCR_WAVES) void k_scramble_src(const PairDesc* __restrict__ desc, const uint32_t* __restrict__ s, const int32_t* __restrict__ pred,
                                                         const unsigned long long* __restrict__ execbits, int32_t* __restrict__ src, int32_t* __restrict__ flags, int max_walk,
                                                         const uint16_t* __restrict__ bin16, uint16_t* __restrict__ binpos, uint32_t* __restrict__ counts, int V,
                                                         int n_pairs, int chunks, const int32_t* __restrict__ vrange) {
    extern __shared__ __attribute__((aligned(8))) uint32_t lh[];
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;
    const PairDesc d = desc[pair];
    int cs_ = (d.n1 + chunks - 1) / chunks; cs_ = (cs_ + kBlock - 1) / kBlock * kBlock;
    const int lo_ = chunk * cs_, hi_ = min(d.n1, lo_ + cs_);
    const size_t o = d.off1;
    const unsigned long long* gbits = execbits + exec_word_base(d.off1, pair);
    // the tile's voxel histogram as 16-bit counters, two per word (a tile holds <= 4096 rows: no carry into the neighbour), then the
    // pair's exec bit table: 3.6 + 15 KB for a 116 k-row scan on 75 x 24 -- eight blocks per CU
    const int hist_words = ((V + 1) / 2 + 1) & ~1;                     // even: the bit table behind it is 8-byte aligned
    unsigned long long* lbits = reinterpret_cast<unsigned long long*>(lh + hist_words);
    for (int b = threadIdx.x; b < hist_words; b += kBlock) lh[b] = 0u;
    if (kBitsInLds) {
        // copied through registers, all reads of a thread in flight together: written as a plain loop the copy is compiled to
        // load - wait - store, one memory round trip per 256 words, i.e. 7 in a row for a 116 k-row scan -- in each of its 61 blocks
        const int nw = (d.n1 + 63) >> 6;
        constexpr int kCopy = 8;
        for (int i0 = 0; i0 < nw; i0 += kCopy * kBlock) {
            unsigned long long t[kCopy];
#pragma unroll
            for (int k = 0; k < kCopy; k++) { const int i = i0 + k * kBlock + (int)threadIdx.x; t[k] = (i < nw) ? gbits[i] : 0ull; }
#pragma unroll
            for (int k = 0; k < kCopy; k++) { const int i = i0 + k * kBlock + (int)threadIdx.x; if (i < nw) lbits[i] = t[k]; }
        }
    }
    __syncthreads();
    // the bit table as 32-bit words (bit u of the 64-bit words = bit u & 31 of word u >> 5): one read + one v_bfe per look-up
    const uint32_t* bits32 = kBitsInLds ? reinterpret_cast<const uint32_t*>(lbits) : reinterpret_cast<const uint32_t*>(gbits);
    auto exec = [&](int u) -> bool { return (bits32[u >> 5] >> (u & 31)) & 1u; };
    for (int base = lo_ + threadIdx.x; base < hi_; base += kWalk * kBlock) {
        int f[kWalk]; uint32_t act[kWalk];                      // act: 0 / ~0 -- masks, not bools: a bool that lives across the rounds costs a v_cndmask 0/1 and a v_cmp per use
        int pv[kWalk];
#pragma unroll
        for (int k = 0; k < kWalk; k++) {
            const int v = base + k * kBlock;
            pv[k] = pred[o + (v < hi_ ? v : lo_)];
        }
        uint32_t any = 0u;
#pragma unroll
        for (int k = 0; k < kWalk; k++) {
            const int v = base + k * kBlock;
            const bool valid = v < hi_;
            const bool moved = valid & (pv[k] != v);            // s[v] != v  <=>  pred[v] != v (fixed points of a permutation)
            const bool head = moved & exec(valid ? v : lo_) & !exec(pv[k]);   // head of a run of executed steps (no short circuit: both table reads leave at once)
            // f: the row that lands on v -- pred(v), v itself for a fixed point; a head walks it forward from v
            f[k] = head ? v : (moved ? pv[k] : v);
            act[k] = head ? 0xFFFFFFFFu : 0u;
            any |= act[k];
        }
        // The walk is paid in VALU issue by all eight waves of a SIMD (timing builds: it is 115 of this kernel's 245 us, and a round
        // of the loop was ~160 instructions): a chain is its walking pointer f and one mask, the length is the thread's round count
        // (all of a thread's chains start together), and after the loads a chain costs a v_bfe_i32 (the bit, as a mask), a v_bfi (the
        // pointer), an and and an or -- the eight table reads leave together.
        bool over = false;
        int rounds = 0;
        while (any) {
            int nu[kWalk];
#pragma unroll
            for (int k = 0; k < kWalk; k++) nu[k] = act[k] ? (int)s[o + f[k]] : 0;
            uint32_t w[kWalk];
#pragma unroll
            for (int k = 0; k < kWalk; k++) w[k] = kBitsInLds ? bits32[nu[k] >> 5] : (act[k] ? bits32[nu[k] >> 5] : 0u);     // (an idle chain reads word 0 of the LDS table)
            rounds++;
            any = 0u;
#pragma unroll
            for (int k = 0; k < kWalk; k++) {
                const uint32_t exm = (uint32_t)__builtin_amdgcn_sbfe((int)w[k], nu[k] & 31, 1);       // step nu executed: ~0
                f[k] = (int)(((uint32_t)nu[k] & act[k]) | ((uint32_t)f[k] & ~act[k]));
                act[k] &= exm;
                any |= act[k];
            }
            if (rounds > max_walk) { over = any != 0u; any = 0u; }   // (uniform) chains still running: the serial replay takes the pair
#ifdef ICET_TIMING_SCR_CAP
            if (rounds >= ICET_TIMING_SCR_CAP) any = 0u;            // timing build only (wrong src, valid rows)
#endif
        }
        if (over) atomicOr(&flags[pair], 1);
        uint16_t fb[kWalk];                                      // packed word of the row that lands here (its voxel id is what the histogram needs)
#pragma unroll
        for (int k = 0; k < kWalk; k++) { const int v = base + k * kBlock; fb[k] = (v < hi_) ? bin16[o + f[k]] : (uint16_t)0; }
#pragma unroll
        for (int k = 0; k < kWalk; k++) {
            const int v = base + k * kBlock;
            if (v < hi_) {
                src[o + v] = f[k];
                const uint16_t b = fb[k] & kBinMask;
                binpos[o + v] = fb[k];                              // voxel id + the row's near-edge flag
                atomicAdd(&lh[b >> 1], 1u << (16u * (b & 1u)));
            }
        }
    }
    __syncthreads();
    // only the pair's populated voxel range: the tables of the multi-split are (tiles x V) words -- 109 MB per 256-pair batch each, written
    // here, scanned by k_bin_tiles, read again by k_bin_scatter -- and nothing outside the range is ever read
    uint32_t* out = counts + ((size_t)pair * chunks + chunk) * V;
    const int v0 = min(max(vrange[2 * pair], 0), V), v1 = min(vrange[2 * pair + 1], V - 1);  // (an empty scan: lo = INT_MAX, hi = -1 -> v0 = V > v1, and no overflow in v0 + threadIdx.x)
    for (int b = v0 + threadIdx.x; b <= v1; b += kBlock) out[b] = (lh[b >> 1] >> (16u * (b & 1u))) & 0xFFFFu;
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
        const uint16_t wd = bin16[o + src[o + v]];
        binpos[o + v] = wd;
        atomicAdd(&lh[wd & kBinMask], 1u);
    }
    __syncthreads();
    uint32_t* out = counts + ((size_t)pair * chunks + chunk) * V;
    for (int b = threadIdx.x; b < V; b += kBlock) out[b] = lh[b];
}

// Adversarial permutations (a walk longer than max_walk set the pair's flag; never on lidar data: observed depth <= 14): ONE launch
// replays the literal swap loop on one lane (k_scramble_serial's body) and then redoes, tile by tile, what k_scramble_src fused in --
// the voxel word of every position and the tiles' voxel histograms.  One block per pair; a pair whose flag is clear costs the launch
// and nothing else (two separate always-launched kernels cost a single pair ~5 us each for nothing).
__global__ __launch_bounds__(kBlock) void k_scramble_replay(const PairDesc* __restrict__ desc, const uint32_t* __restrict__ s, int32_t* __restrict__ idx_tmp,
                                                            int32_t* __restrict__ src, const int32_t* __restrict__ flags, const uint16_t* __restrict__ bin16,
                                                            uint16_t* __restrict__ binpos, uint32_t* __restrict__ counts, int V, int chunks) {
    extern __shared__ uint32_t lh[];
    const int pair = blockIdx.x;
    if (!(flags[pair] & 1)) return;
    const PairDesc d = desc[pair];
    const size_t o = d.off1;
    if (threadIdx.x == 0) {
        for (int i = 0; i < d.n1; i++) { idx_tmp[o + i] = (int)s[o + i]; src[o + i] = i; }
        for (int i = 0; i < d.n1; i++) {
            int j = idx_tmp[o + i];
            if (j != i) {
                int t = src[o + i]; src[o + i] = src[o + j]; src[o + j] = t;
                idx_tmp[o + i] = idx_tmp[o + j]; idx_tmp[o + j] = j;
            }
        }
        __threadfence();
    }
    __syncthreads();
    int cs_ = (d.n1 + chunks - 1) / chunks; cs_ = (cs_ + kBlock - 1) / kBlock * kBlock;
    for (int chunk = 0; chunk < chunks; chunk++) {
        const int lo_ = chunk * cs_, hi_ = min(d.n1, lo_ + cs_);
        for (int b = threadIdx.x; b < V; b += kBlock) lh[b] = 0u;
        __syncthreads();
        for (int v = lo_ + threadIdx.x; v < hi_; v += kBlock) {
            const uint16_t wd = bin16[o + src[o + v]];
            binpos[o + v] = wd;
            atomicAdd(&lh[wd & kBinMask], 1u);
        }
        __syncthreads();
        uint32_t* out = counts + ((size_t)pair * chunks + chunk) * V;
        for (int b = threadIdx.x; b < V; b += kBlock) out[b] = lh[b];
        __syncthreads();
    }
}

// Exclusive scan over (class, tile) in class-major order, in two steps so that a single large pair (7200 voxels x 240 tiles)
// is not scanned by ONE block: k_bin_tiles (one thread per class, blocks over classes) turns each class's per-tile counts into
// per-tile offsets and leaves the class total in class_start[]; k_bin_scan (one block per pair) scans the totals in place.
// A block takes 64 classes; its four waves split the pair's tiles into quarters: wave g walks quarter g of every class (coalesced: a lane
// per class), keeps the running counts of its quarter in registers, and the quarters are joined through LDS -- one round of loads, one
// barrier, one round of stores per 4 x 16 tiles (a thread per class walking ALL tiles took four dependent rounds on a 116 k-row scan).
constexpr int kTileClasses = 64;
__global__ __launch_bounds__(kBlock) void k_bin_tiles(const uint32_t* __restrict__ counts, uint32_t* __restrict__ tile_base, int32_t* __restrict__ class_start,
                                                      int V, int chunks, const int32_t* __restrict__ class_range) {
    __shared__ int s_tot[kBlock / 64][kTileClasses];
    const int pair = blockIdx.y, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int b = blockIdx.x * kTileClasses + lane;
    int lo = 0, hi = V - 1;
    if (class_range) { lo = class_range[2 * pair]; hi = class_range[2 * pair + 1]; }
    const int b_first = blockIdx.x * kTileClasses, b_last = min(b_first + kTileClasses, V) - 1;
    if (b_last < lo || b_first > hi) {                            // no row of the pair has a class of this block (block-uniform)
        if (g == 0 && b < V) class_start[(size_t)pair * (V + 1) + b] = 0;
        return;
    }
    const bool live = b < V && b >= lo && b <= hi;
    constexpr int kQ = kBlock / 64, kTrip = 16;
    const int per = (chunks + kQ - 1) / kQ;                       // tiles per quarter
    const int t_lo = min(g * per, chunks), t_hi = min(t_lo + per, chunks);
    const uint32_t* c = counts + (size_t)pair * chunks * V + b;
    uint32_t* tb = tile_base + (size_t)pair * chunks * V + b;
    // pass 1: the quarter's total (for quarters of at most kTrip tiles -- scans up to ~130 k rows -- the counts stay in registers for pass 2)
    uint32_t x[kTrip];                                            // the quarter's FIRST trip stays in registers for pass 2
    int tot = 0;
#pragma unroll
    for (int k = 0; k < kTrip; k++) { x[k] = (live && t_lo + k < t_hi) ? c[(size_t)(t_lo + k) * V] : 0u; tot += (int)x[k]; }
    for (int t0 = t_lo + kTrip; t0 < t_hi; t0 += kTrip) {         // longer quarters (scans above ~130 k rows): further trips of 16 loads in flight
        uint32_t y[kTrip];
#pragma unroll
        for (int k = 0; k < kTrip; k++) y[k] = (live && t0 + k < t_hi) ? c[(size_t)(t0 + k) * V] : 0u;
#pragma unroll
        for (int k = 0; k < kTrip; k++) tot += (int)y[k];
    }
    s_tot[g][lane] = tot;
    __syncthreads();
    int run = 0;
    for (int q = 0; q < g; q++) run += s_tot[q][lane];
    // pass 2: where each tile's rows of the class start
#pragma unroll
    for (int k = 0; k < kTrip; k++) { if (live && t_lo + k < t_hi) tb[(size_t)(t_lo + k) * V] = (uint32_t)run; run += (int)x[k]; }
    for (int t0 = t_lo + kTrip; t0 < t_hi; t0 += kTrip) {
        uint32_t y[kTrip];
#pragma unroll
        for (int k = 0; k < kTrip; k++) y[k] = (live && t0 + k < t_hi) ? c[(size_t)(t0 + k) * V] : 0u;
#pragma unroll
        for (int k = 0; k < kTrip; k++) { if (live && t0 + k < t_hi) tb[(size_t)(t0 + k) * V] = (uint32_t)run; run += (int)y[k]; }
    }
    if (g == kQ - 1 && b < V) class_start[(size_t)pair * (V + 1) + b] = live ? run : 0;
}

// live (optional): the classes holding at least live_min rows, compacted in class order (16-byte records, pairs x V of them) -- the angular bins fitCells1 looks at at
// all (src/icet.cpp:115) -- so that the fit kernels walk ~1/4 of the grid instead of launching a wave per bin.
constexpr int kScanBlock = 1024;               // one block per pair scans its V classes: 2 rounds on 75 x 24, 8 on 150 x 48
__global__ __launch_bounds__(kScanBlock) void k_bin_scan(int32_t* __restrict__ class_start, int V, int32_t* __restrict__ live, int32_t* __restrict__ n_live, int live_min,
                                                     uint32_t* __restrict__ n_items, const int32_t* __restrict__ tile_vr, int chunks, int32_t* __restrict__ vrange_out) {
    __shared__ int wave_tot[kScanBlock / 64], wave_live[kScanBlock / 64];
    __shared__ int base, lbase;
    const int pair = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { base = 0; lbase = 0; }
    if (vrange_out && wave == kScanBlock / 64 - 1) {                  // (rank-sort call) the pair's voxel range from its tiles': the last wave, beside the scan
        int lo = 0x7FFFFFFF, hi = -1;
        for (int t = lane; t < chunks; t += 64) { lo = min(lo, tile_vr[((size_t)pair * chunks + t) * 2]); hi = max(hi, tile_vr[((size_t)pair * chunks + t) * 2 + 1]); }
        lo = wave_reduce_min_i(lo); hi = wave_reduce_max_i(hi);
        if (lane == 0) { vrange_out[2 * pair] = lo; vrange_out[2 * pair + 1] = hi; }
    }
    __syncthreads();
    for (int v0 = 0; v0 < V; v0 += kScanBlock) {
        const int b = v0 + threadIdx.x;
        const int tot = (b < V) ? class_start[(size_t)pair * (V + 1) + b] : 0;
        const int incl = wave_incl_sum(tot);
        const unsigned long long lm = __ballot(live != nullptr && b < V && tot >= live_min);
        if (lane == 63) wave_tot[wave] = incl;
        if (lane == 0) wave_live[wave] = __popcll(lm);
        __syncthreads();
        int woff = 0, loff = 0;
        for (int k = 0; k < wave; k++) { woff += wave_tot[k]; loff += wave_live[k]; }
        const int bb = base, lb = lbase;
        if (b < V) class_start[(size_t)pair * (V + 1) + b] = bb + woff + incl - tot;
        if ((lm >> lane) & 1ull)      // {class, first row, rows, (candidates: k_fit_cluster)}: what a fit wave needs about its bin in ONE 16-byte read
            reinterpret_cast<int4*>(live)[(size_t)pair * V + lb + loff + __popcll(lm & ((1ull << lane) - 1ull))] = make_int4(b, bb + woff + incl - tot, tot, 0);
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) { base = bb + woff + incl; lbase = lb + loff + __popcll(lm); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { class_start[(size_t)pair * (V + 1) + V] = base; if (n_live) n_live[pair] = lbase; if (n_items) n_items[pair] = 0u; }
}

// One block per tile (<= 2048 positions); wave w owns the w-th quarter (<= 8 rounds of 64 positions).  Everything
// global is loaded up front and stored at the end, so the only serial chain is 8 rounds of ballots + LDS.
// kLdsRank: the rank of a row among the rows of its wave's quarter that hold the same voxel is the value the LDS atomic of the counting
// step hands back -- gfx950 serves the lanes of one ds_add_rtn that hit the same address in ascending lane order, and a wave's LDS
// instructions in program order (scripts/hip/lds_atomic_order.hip: 10^10 atomics, no exception; checked again by every context at creation,
// lds_rank_selftest; the ballot form below stays as the fallback and as the tests' cross-check).  The ballot form costs ~10 VALU
// instructions per id bit and round -- 11 bits x 8 rounds -- and this kernel is bound by VALU issue (4 cycles per wave64 instruction).
constexpr int kScatterRounds = kKfMaxPtsPerThread;   // a tile is at most 4 waves x this many rounds x 64 positions
template <bool kLdsRank>
__global__ __launch_bounds__(kBlock) void k_bin_scatter(const PairDesc* __restrict__ desc, const int32_t* __restrict__ src, const uint16_t* __restrict__ binpos,
                                                        const uint32_t* __restrict__ tile_base, const int32_t* __restrict__ bin_start,
                                                        uint32_t* __restrict__ sorted_row, int V, int vbits, int n_pairs, int chunks, const int32_t* __restrict__ vrange) {
    // LDS: gb[V] (u32: where the tile's rows of a voxel start in the pair's table) | lc[4][V] (u16: per-wave counts, then the running
    // offset of each wave inside the tile's rows of the voxel).  12 bytes per voxel -- 21.6 KB for 75 x 24, seven blocks per CU; with
    // four u32 arrays (28.8 KB, five blocks) the kernel took 260 us per 256 pairs: it wants occupancy (capped at 4 / 3 blocks: 281 / 334).
    extern __shared__ uint32_t lb[];
    uint32_t* gb = lb;
    uint16_t* lc = reinterpret_cast<uint16_t*>(lb + V);
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
    for (int i = threadIdx.x; i < 2 * V; i += kBlock) lb[V + i] = 0u;  // the 4 x V u16 counters, two per word
    __syncthreads();
    const size_t o = d.off1;
    uint32_t bb[kScatterRounds]; uint32_t row[kScatterRounds]; bool ok[kScatterRounds];
    uint16_t* mine = lc + wave * V;
    // Two loops: all of the thread's loads first (unconditional, at a clamped index), then the LDS atomics.  Written as one loop with
    // `ok ? load : 0` every round sits in a branch of its own -- load, wait, atomic -- and the rounds are eight dependent memory round trips.
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++) {
        const int v = wlo + 64 * k + lane;
        ok[k] = (k < rounds) & (v < whi);
        const size_t vv = o + (size_t)(ok[k] ? v : lo_);
        bb[k] = binpos[vv]; row[k] = (uint32_t)src[vv];
    }
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++) {
        const uint32_t wd = ok[k] ? bb[k] : 0u;
        bb[k] = wd & kRowBinMask;
        row[k] = (ok[k] ? row[k] : 0u) | ((wd & kRowNearBit) ? kSortedNearBit : 0u) | ((wd & kRowZeroBit) ? kSortedZeroBit : 0u);   // the flags travel with the row
        if (ok[k]) {                                                   // 16-bit counters, 32-bit atomics: a quarter tile holds <= 512 rows, no carry into the neighbour
            const uint32_t e = (uint32_t)(wave * V) + bb[k];
            const uint32_t old = atomicAdd(&lb[V + (e >> 1)], 1u << (16u * (e & 1u)));
            if (kLdsRank) bb[k] |= ((old >> (16u * (e & 1u))) & 0xFFFFu) << 16;     // earlier rows of this wave with the same voxel (<= 511), kept above the 14 id bits
        }
    }
    __syncthreads();
    {
        const uint32_t* tb = tile_base + ((size_t)pair * chunks + chunk) * V;
        const int32_t* bst = bin_start + (size_t)pair * (V + 1);
        const int v0 = min(max(vrange[2 * pair], 0), V), v1 = min(vrange[2 * pair + 1], V - 1);   // no row of the pair lies outside: nothing there is looked up
        for (int b = v0 + threadIdx.x; b <= v1; b += kBlock) {
            const uint32_t c0 = lc[b], c1 = lc[V + b], c2 = lc[2 * V + b];
            gb[b] = (uint32_t)bst[b] + tb[b];
            lc[b] = 0; lc[V + b] = (uint16_t)c0; lc[2 * V + b] = (uint16_t)(c0 + c1); lc[3 * V + b] = (uint16_t)(c0 + c1 + c2);
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t dest[kScatterRounds];
    if (kLdsRank) {
#pragma unroll
        for (int k = 0; k < kScatterRounds; k++) { const uint32_t b = bb[k] & 0xFFFFu; dest[k] = ok[k] ? gb[b] + (uint32_t)mine[b] + (bb[k] >> 16) : 0u; }
    } else
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
            const uint32_t run = mine[bb[k]];
            dest[k] = gb[bb[k]] + run + (uint32_t)rank;
            if (rank == 0) mine[bb[k]] = (uint16_t)(run + (uint32_t)__popcll(peers));  // one leader per distinct voxel in this round
        }
    }
    // one scattered 4-byte store per row; k_fit_cluster gathers the coordinates through it (element-wise scattered stores of
    // the three coordinate arrays cost 4x more than gathering them, and a separate gather pass 0.15 ms more than gathering
    // inside the fit, whose independent loads hide the latency)
#pragma unroll
    for (int k = 0; k < kScatterRounds; k++)
        if (ok[k]) sorted_row[o + dest[k]] = row[k];
}
// fitCells1 (src/icet.cpp:109-252) in four kernels, each shaped for what bounds it:
//   k_fit_cluster    one wave per angular bin: findCluster (:557-607) on the bin's rows in stored (scrambled) order, then the rows
//                    inside the radial range are COMPACTED to the front of the bin's segment of cand[] / cand_r[] and announced as
//                    work items of <= 64 rows.  Pointer chasing through the sorted-row table: latency bound, light on registers,
//                    8 waves per SIMD.
//   k_fit_roundtrip  one wave per work item, full lanes: the angular half of filterPointsInsideCluster (:609-652) where it is not
//                    implied by the row's classification, and the reference's spherical -> Cartesian round trip (:159) under the
//                    shared arithmetic rule -- double-precision atan2 / acos, ~350 DP instructions per row: issue bound, 128 VGPRs.
//   k_fit_moments    one wave per bin: mean and covariance of the surviving rows (:160-162), exact sums -- a short coalesced read.
//   k_fit_finish     one LANE per bin: 3x3 eigen-decomposition, sigma points, slot records (scalar work).
// One fused kernel (one wave per bin for everything, round 2's first version) took 437 us per 256 pairs: the double-precision code
// set its register budget (4 waves per SIMD, which the latency-bound walk wanted to be 8) and ran beside half-empty waves.
// Blocks per launch of the fit kernels (a fixed number of blocks per pair walks the pair's live bins / work items).  Measured on 256 pairs:
// k_fit_cluster 512 blocks 213 us, 1024: 133, 1536: 110, 2048: 111, 4096: 152, 8192: 195, 16384: 220 -- more waves in flight make its
// gathers and returning atomics slower, not faster; k_fit_moments (a short coalesced read per bin) the other way: 2048: 70, 8192: 61;
// k_fit_roundtrip 1024: 115, 2048: 104, 4096: 103.
#ifndef ICET_FIT_BLOCKS
#define ICET_FIT_BLOCKS 2048
#endif
#ifndef ICET_MOM_BLOCKS
#define ICET_MOM_BLOCKS 8192
#endif
#ifndef ICET_RT_BLOCKS
#define ICET_RT_BLOCKS 2048
#endif
// k_fit_cluster pipelines the front of its per-bin chain (record -> rows -> their r -> walk) as far as 64 VGPRs allow: the rows of bin
// k + 1 and the record of bin k + 2 are in flight while bin k is walked (ICET_CLUSTER_PIPE 2, 62 VGPRs); the r gather of the current bin
// stays exposed.  The deeper pipeline of k_fit_roundtrip / k_fit_moments (1: also the r of bin k + 1) needs 72 VGPRs, and the kernel is
// bound by the walk's own chain of DPP scans, ballots and LDS shuffles, which wants 8 waves per SIMD -- measured per 256-pair keyframe:
// 0 (no look-ahead) 1.482 ms, 2: 1.467, 1 at 6 waves: +0.02, 1 at 8 waves (10 spills): +0.04.  k_fit_moments keeps its pipeline at
// 8 waves (2 spilled words, outside the loop).
#ifndef ICET_CLUSTER_WAVES
#define ICET_CLUSTER_WAVES 8
#endif
#ifndef ICET_MOM_WAVES
#define ICET_MOM_WAVES 8
#endif
#ifndef ICET_CLUSTER_PIPE
#define ICET_CLUSTER_PIPE 2
#endif
struct FitItem { int32_t base, v, k0, nb; };      // rows k0 .. k0 + nb of bin v's compacted candidates; base = the bin's start + k0 (relative to the pair's segment)
// A pair owns the item slots [item_base, item_base + n1 / 64 + V): at most one partial batch per bin plus the full ones.
__device__ __host__ __forceinline__ size_t item_base(int32_t off1, int pair, int V) { return (size_t)(off1 / 64) + (size_t)pair * (size_t)(V + 1); }

// kTail: chunks of 64 rows requested together past a bin's first 256 rows.  A real scan's near field puts tens of thousands of rows into a few
// bins (the reference's sample_pc pair: one wave walked 237 us while every other wave had long finished -- two dependent memory round trips
// per 64 rows).  Throughput batches keep 8 waves per SIMD (kTail 2 fits their 64 registers); small batches, whose waves are few anyway, take
// kTail 16 at 4 waves per SIMD: two round trips per 1024 rows.
template <int kTail>
__global__ __launch_bounds__(kBlock, kTail > 2 ? 4 : ICET_CLUSTER_WAVES) void k_fit_cluster(const PairDesc* __restrict__ desc, const int32_t* __restrict__ bin_start,
                                                      const uint32_t* __restrict__ sorted_row, const float* __restrict__ r1,
                                                      uint32_t* __restrict__ cand, float* __restrict__ cand_r, FitItem* __restrict__ items, uint32_t* __restrict__ n_items,
                                                      int32_t* __restrict__ live, const int32_t* __restrict__ n_live,
                                                      FitMid* __restrict__ midD, int T, int P, int n, float thresh, float buff, int n_pairs, int chunks, int half_gap,
                                                      const int32_t* __restrict__ zero_rows, int4 zv) {
    __shared__ float stage[kBlock / 64][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int V = T * P;
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;          // all bins of a pair on one XCD: its tables stay in that L2
    const PairDesc d = desc[pair];
    const int nl = n_live[pair];
    // a fixed number of waves per pair walks the pair's live bins (those with >= n rows, src/icet.cpp:115): a wave per bin of the
    // grid would launch 4 x as many waves as have work (measured: ~90 us of dispatching empty blocks per 256-pair launch).
    // A bin is a chain of dependent reads -- bin record ({bin, first row, rows}: one 16-byte read) -> the bin's rows -> their r (a
    // gather) -- before the walk can start; see ICET_CLUSTER_PIPE above for how much of it runs ahead.
    const int stride = chunks * (kBlock / 64), j0 = chunk * (kBlock / 64) + wave;
    if (j0 >= nl) return;                                             // wave-uniform; no block-wide barrier below
    // A bin that holds NOTHING BUT exact-zero rows (the invalid returns of a real scan: 5 k - 24 k rows in one voxel) has no cluster: the walk finds no break
    // behind its first row, first.r == 0 gives (0, 0) (src/icet.cpp:592-604), nothing is a candidate.  Walked and filtered by one wave it was two memory round trips per
    // 128 rows: 190 us per 256 real pairs for this kernel against 100 on synthetic scans.  k_scan1_spherical has counted the pair's exact-zero rows per sign pattern
    // (= per voxel they land in): a bin whose row count equals that number is skipped with the result the walk would give.
    int zr0 = 0, zr1 = 0, zr2 = 0, zr3 = 0;
    if (zero_rows) { zr0 = zero_rows[4 * pair]; zr1 = zero_rows[4 * pair + 1]; zr2 = zero_rows[4 * pair + 2]; zr3 = zero_rows[4 * pair + 3]; }
    const size_t po = (size_t)d.off1;
    constexpr uint32_t kRowMask = kSortedRowMask;
    // r of a row of the sorted-row table: rows flagged `r == 0` (the invalid returns of a real scan: one voxel holds thousands of them) are not gathered
    auto r_of = [&](uint32_t rw, bool valid) { return (valid && !(rw & kSortedZeroBit)) ? r1[po + (rw & kRowMask)] : 0.f; };
    constexpr int kCache = 4;                                         // a bin's first 4 x 64 rows travel in registers (most bins hold ~100-400 rows)
    const int4* lv = reinterpret_cast<const int4*>(live) + (size_t)pair * V;
    auto info = [&](int j) { int4 q = lv[min(j, nl - 1)]; if (j >= nl) q.z = 0; return q; };
    auto rows4 = [&](const int4& q, uint32_t (&rw)[kCache]) {
#pragma unroll
        for (int k = 0; k < kCache; k++) rw[k] = (lane + 64 * k < q.z) ? sorted_row[po + q.y + lane + 64 * k] : 0u;
    };
    auto radii4 = [&](const int4& q, const uint32_t (&rw)[kCache], float (&rr)[kCache]) {
#pragma unroll
        for (int k = 0; k < kCache; k++) rr[k] = r_of(rw[k], lane + 64 * k < q.z);
    };
#if ICET_CLUSTER_PIPE == 1
    int4 q0 = info(j0), q1 = info(j0 + stride), q2 = info(j0 + 2 * stride);
    uint32_t prw[kCache], prw1[kCache];
    rows4(q0, prw); rows4(q1, prw1);
    float pr[kCache];
    radii4(q0, prw, pr);
#elif ICET_CLUSTER_PIPE == 2
    int4 q0 = info(j0), q1 = info(j0 + stride);
    uint32_t prw[kCache];
    rows4(q0, prw);
#endif
    for (int j = j0; j < nl; j += stride) {
#if ICET_CLUSTER_PIPE == 1
    const int4 q3 = info(j + 3 * stride);
    uint32_t prw2[kCache];
    rows4(q2, prw2);
    float pr1[kCache];
    radii4(q1, prw1, pr1);
#elif ICET_CLUSTER_PIPE == 2
    const int4 q2 = info(j + 2 * stride);
    uint32_t prw1[kCache];
    rows4(q1, prw1);
    float pr[kCache];
    radii4(q0, prw, pr);
#else
    const int4 q0 = info(j);
    uint32_t prw[kCache]; float pr[kCache];
    rows4(q0, prw); radii4(q0, prw, pr);
#endif
    const int v = q0.x, bs = q0.y, cnt = q0.z;
    const size_t base = po + bs;
    // rows of this bin in (scrambled) position order: sorted_row[base + i] is the row of the pair's input-order tables
    // (top bit: the row lies within a guard band of a voxel edge)
    auto RS = [&](int i) { return r_of(sorted_row[base + i], true); };

    float inner = 0.f, outer = 0.f;
    int m_cand = 0;                                                   // rows inside the radial range (wave-uniform)
    const int all_zero = (v == zv.x ? zr0 : 0) + (v == zv.y ? zr1 : 0) + (v == zv.z ? zr2 : 0) + (v == zv.w ? zr3 : 0);

    if (all_zero != cnt) {
        // ---- findCluster (src/icet.cpp:557-607): first run of >= n consecutive points whose
        // successive |dr| <= thresh, walking the bin in stored (scrambled) order.
        int run_start = 0; float front = 0.f; float carry_prev = 0.f; bool found = false;
        float front_before = 0.f;                                     // the point walked just before `front` (extension ICET_FLAG_HALF_GAP_BOUNDS)
        // the half-gap rule of the Python variant (python/utils.py:92-119): a bound reaches half way to the nearest point outside the
        // cluster, at most buff; buff where there is no such point
        auto in_buff = [&](float fr, int start, float before) { return (half_gap && start > 0) ? fminf(buff, 0.5f * fabsf(fr - before)) : buff; };
        // A break is a point that does not continue the current run; a run is reported at the first break that closes
        // >= n points.  In the scrambled order most points are breaks, so instead of visiting the breaks of a 64-point
        // chunk one after another, every break lane looks up the break before it with a prefix-max scan and the first
        // lane whose run is long enough is picked with a ballot.
        auto walk = [&](int c0, float r) {
            const int i = c0 + lane; const bool valid = i < cnt;
            const float prev = wave_shr1(r, carry_prev);
            const bool brk = valid && (i == 0 || !(fabsf(prev - r) <= thresh));
            // a chunk without a single break only carries the run on (the thousands of exact-zero rows of a real scan share one bin and one
            // r: 360 chunks of it cost 160 us at ~100 instructions each before this test)
            if (__ballot(brk) == 0ull) { carry_prev = wave_read(r, 63); return; }
            const int pm = wave_incl_max(brk ? i : -1);              // inclusive prefix max of break positions (DPP: no LDS crossbar)
            int prevb = wave_shr1(pm, -1);                           // last break strictly before this lane ...
            prevb = max(prevb, run_start);                           // ... or the run carried in from earlier chunks
            const unsigned long long hit = __ballot(brk && (i - prevb >= n));
            if (hit) {
                const int b = __ffsll((long long)hit) - 1;
                // (every lane index below is wave-uniform: v_readlane, not a shuffle through LDS)
                const int rs0 = wave_read(prevb, b);                 // start of the run that this break closes
                const float back = (b > 0) ? wave_read(r, b - 1) : carry_prev;
                const float fr = (rs0 >= c0) ? wave_read(r, rs0 - c0) : front;
                const float before = (rs0 > c0) ? wave_read(r, max(rs0 - 1 - c0, 0)) : (rs0 == c0 ? carry_prev : front_before);
                const float after = wave_read(r, b);                 // the point that ended the run
                inner = fr - in_buff(fr, rs0, before);
                outer = back + (half_gap ? fminf(buff, 0.5f * fabsf(after - back)) : buff); found = true;
            } else {
                const int last = wave_read(pm, 63);                  // last break of this chunk, if any
                if (last >= 0) { run_start = last; front = wave_read(r, last - c0); front_before = (last > c0) ? wave_read(r, max(last - 1 - c0, 0)) : carry_prev; }
            }
            carry_prev = wave_read(r, 63);
        };
#pragma unroll
        for (int k = 0; k < kCache; k++) if (64 * k < cnt && !found) walk(64 * k, pr[k]);
        // rows past the cached ones: kTail chunks' rows, then their radii, are requested together
        for (int c0 = 64 * kCache; c0 < cnt && !found; c0 += 64 * kTail) {
            uint32_t tw[kTail]; float tr[kTail];
#pragma unroll
            for (int k = 0; k < kTail; k++) tw[k] = (c0 + 64 * k + lane < cnt) ? sorted_row[base + c0 + 64 * k + lane] : 0u;
#pragma unroll
            for (int k = 0; k < kTail; k++) tr[k] = r_of(tw[k], c0 + 64 * k + lane < cnt);
#pragma unroll
            for (int k = 0; k < kTail; k++) if (c0 + 64 * k < cnt && !found) walk(c0 + 64 * k, tr[k]);
        }
        if (!found && cnt - run_start >= n) {
            if (front != 0.f) { const float back = RS(cnt - 1); inner = front - in_buff(front, run_start, front_before); outer = back + buff; }
            else { inner = 0.f; outer = 0.f; }
        }
        // ---- the radial half of filterPointsInsideCluster (src/icet.cpp:609-652): rows with r in [inner, outer], compacted.  Nothing
        // is fitted unless outerDistance > 0.1 (:158, float against a double literal), so bins without a cluster stop here.
#if !(defined(ICET_EXP_FIT) && ICET_EXP_FIT == 2)
        if ((double)outer > 0.1) {
            const unsigned long long lt = (1ull << lane) - 1ull;
            auto keep = [&](int c0, uint32_t rw, float r) {
                const int i = c0 + lane;
                // (a row with r == 0 -- the invalid returns of a real scan, thousands of them in ONE bin whose "cluster" is [-buff, buff] -- has phi = 1000 and can
                // never pass the polar bounds (src/icet.cpp:632-633): not a candidate.  As candidates they became ~370 work items of k_fit_roundtrip per real scan,
                // every row of which came back as "did not survive")
                const bool in = (i < cnt) && !(rw & kSortedZeroBit) && (r >= inner) && (r <= outer);
                const unsigned long long m = __ballot(in);
                if (in) { const size_t pos = base + m_cand + __popcll(m & lt); cand[pos] = rw; cand_r[pos] = r; }
                m_cand += __popcll(m);
            };
#pragma unroll
            for (int k = 0; k < kCache; k++) if (64 * k < cnt) keep(64 * k, prw[k], pr[k]);      // the cached chunks
            // rows past the cached ones, kFilter chunks requested together: the bin that holds a real scan's exact-zero rows (5 k - 24 k rows, in ONE wave)
            // was two dependent memory round trips per 64 rows here -- 190 us per 256 real pairs for this kernel against 100 on synthetic scans
            constexpr int kFilter = kTail > 4 ? 8 : 4;
            for (int c0 = 64 * kCache; c0 < cnt; c0 += 64 * kFilter) {
                uint32_t tw[kFilter]; float tr[kFilter];
#pragma unroll
                for (int k = 0; k < kFilter; k++) tw[k] = (c0 + 64 * k + lane < cnt) ? sorted_row[base + c0 + 64 * k + lane] : 0u;
#pragma unroll
                for (int k = 0; k < kFilter; k++) tr[k] = r_of(tw[k], c0 + 64 * k + lane < cnt);
#pragma unroll
                for (int k = 0; k < kFilter; k++) if (c0 + 64 * k < cnt) keep(c0 + 64 * k, tw[k], tr[k]);
            }
            // announce the batches: one list and one counter per pair (a single counter for the whole launch serialises ~50 k
            // returning atomics on one word: measured 0.4 ms)
            const int nbatch = (m_cand + 63) >> 6;
            uint32_t slot = 0;
            if (lane == 0 && nbatch > 0) slot = atomicAdd(&n_items[pair], (uint32_t)nbatch);
            slot = (uint32_t)wave_read((int)slot, 0);
            FitItem* mine = items + item_base(d.off1, pair, V);
            for (int bI = lane; bI < nbatch; bI += 64) { FitItem it; it.base = bs + 64 * bI; it.v = v; it.k0 = 64 * bI; it.nb = min(64, m_cand - 64 * bI); mine[slot + bI] = it; }
        }
#endif
    }
    // one 64-byte record per live bin (the Gaussian is filled in by k_fit_moments; the records of the other bins were zeroed),
    // staged through LDS so that 16 lanes store it with one coalesced instruction
    if (lane == 0) {
        float* g = stage[wave];
#pragma unroll
        for (int k = 0; k < 9; k++) g[k] = 0.f;
        g[9] = inner; g[10] = outer; g[11] = __int_as_float(cnt); g[12] = __int_as_float(0); g[13] = __int_as_float(m_cand); g[14] = g[15] = 0.f;
    }
    // same wave wrote and reads: LDS operations of one wave complete in order
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    if (lane < 16) reinterpret_cast<float*>(midD + (size_t)pair * V + v)[lane] = stage[wave][lane];
    if (lane == 0) live[((size_t)pair * V + j) * 4 + 3] = m_cand;     // for k_fit_moments: the whole of its bin in the one record
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#if ICET_CLUSTER_PIPE == 1
    q0 = q1; q1 = q2; q2 = q3;
#pragma unroll
    for (int k = 0; k < kCache; k++) { prw[k] = prw1[k]; prw1[k] = prw2[k]; pr[k] = pr1[k]; }
#elif ICET_CLUSTER_PIPE == 2
    q0 = q1; q1 = q2;
#pragma unroll
    for (int k = 0; k < kCache; k++) prw[k] = prw1[k];
#endif
    }   // live bins of this wave
}

#ifndef ICET_FIT_WAVES
#define ICET_FIT_WAVES 4      /* register budget of the double-precision round trip (128 VGPRs) */
#endif
__global__ __launch_bounds__(kBlock, ICET_FIT_WAVES) void k_fit_roundtrip(const PairDesc* __restrict__ desc, const int32_t* __restrict__ bin_start,
                                                      const uint32_t* __restrict__ cand, const float* __restrict__ cand_r,
                                                      const FitItem* __restrict__ items, const uint32_t* __restrict__ n_items,
                                                      float* __restrict__ cart1, size_t cart_stride, int T, int P, int n_pairs, int chunks) {
    const int lane = threadIdx.x & 63;
    const int V = T * P;
    constexpr uint32_t kRowMask = kSortedRowMask;
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;          // the pair's blocks share an XCD: its scan and tables sit in that L2
    const PairDesc d = desc[pair];
    const uint32_t n_it = n_items[pair];
    const FitItem* mine = items + item_base(d.off1, pair, V);
    const uint32_t stride = chunks * (kBlock / 64), w0 = chunk * (kBlock / 64) + (threadIdx.x >> 6);
    if (w0 >= n_it) return;                                            // wave-uniform
    typedef __attribute__((address_space(1))) const float gfloat;      // the scan lives in HBM: global_load (vmcnt only), not flat
    gfloat* sx = (gfloat*)d.s1;
    // An item is a chain of dependent reads -- item record -> candidate row -> the row's coordinates (a gather) -- in front of ~350
    // double-precision instructions; walked one item at a time a wave spent three memory round trips per item waiting (76 % of its
    // cycles).  The chain is therefore software-pipelined over the wave's items: while item k is computed the coordinates of item
    // k + 1, the candidate rows of item k + 2 and the record of item k + 3 are in flight.  Look-ahead past the wave's last item
    // re-reads the pair's last record with nb = 0 (valid addresses, nothing stored); lanes past an item's nb read its last row.
    struct Cand { uint32_t rw; float r; };
    auto load_item = [&](uint32_t w) { FitItem it = mine[min(w, n_it - 1u)]; if (w >= n_it) it.nb = 0; return it; };
    auto load_cand = [&](const FitItem& it) { const size_t idx = (size_t)d.off1 + it.base + min(lane, max(it.nb - 1, 0)); Cand c; c.rw = cand[idx]; c.r = cand_r[idx]; return c; };
    auto gather = [&](const Cand& c, float& x, float& y, float& z) { const uint32_t row = c.rw & kRowMask; x = sx[row]; y = sx[d.ld1 + row]; z = sx[2 * (size_t)d.ld1 + row]; };
    FitItem i0 = load_item(w0), i1 = load_item(w0 + stride), i2 = load_item(w0 + 2 * stride);
    Cand c0 = load_cand(i0), c1 = load_cand(i1);
    float x0, y0, z0;
    gather(c0, x0, y0, z0);
    for (uint32_t w = w0; w < n_it; w += stride) {
        const FitItem i3 = load_item(w + 3 * stride);
        const Cand c2 = load_cand(i2);
        float x1, y1, z1;
        gather(c1, x1, y1, z1);
        float th, ph, X, Y, Z;
#if defined(ICET_EXP_FIT) && ICET_EXP_FIT == 1
        X = x0; Y = y0; Z = z0; th = 0.f; ph = 0.f;
#else
        roundtrip_cr(x0, y0, z0, c0.r, th, ph, X, Y, Z);
#endif
        // The angular half of the filter holds by construction for a row classified away from every edge (the same argument as
        // for the bin itself); a row flagged near-edge is tested with the literal formulas.  NaN angles (r = 0, NaN rows) fail the
        // test, like the reference's 1000 sentinel.
        if (c0.rw & kSortedNearBit) {
            float az0, az1, el0, el1;
            voxel_limits(i0.v % T, i0.v / T, T, P, az0, az1, el0, el1);
            if (!(th >= az0 && th <= az1 && ph >= el0 && ph <= el1)) X = __builtin_nanf("");     // marks "did not survive"
        }
        if (lane < i0.nb) {
            const size_t idx = (size_t)d.off1 + i0.base + lane;
            cart1[idx] = X; cart1[cart_stride + idx] = Y; cart1[2 * cart_stride + idx] = Z;
        }
        i0 = i1; i1 = i2; i2 = i3; c0 = c1; c1 = c2; x0 = x1; y0 = y1; z0 = z1;
    }
}

__global__ __launch_bounds__(kBlock, ICET_MOM_WAVES) void k_fit_moments(const PairDesc* __restrict__ desc, const int32_t* __restrict__ bin_start, const float* __restrict__ cart1,
                                                      size_t cart_stride, const int32_t* __restrict__ live, const int32_t* __restrict__ n_live,
                                                      FitMid* __restrict__ midD, int V, int n, int n_pairs, int chunks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;
    const int nl = n_live[pair];
    const int stride = chunks * (kBlock / 64), j0 = chunk * (kBlock / 64) + wave;
    if (j0 >= nl) return;
    // Software-pipelined like the two kernels before it: while bin k is summed, the first 256 round-tripped rows of bin k + 1 and the
    // record of bin k + 2 ({class, first row, rows, candidates}: k_bin_scan / k_fit_cluster) are in flight.
    constexpr int kKeep = 4;                                          // the first 256 rows stay in registers for the centred pass
    const size_t po = (size_t)desc[pair].off1;
    const int4* lv = reinterpret_cast<const int4*>(live) + (size_t)pair * V;
    auto info = [&](int j) { int4 q = lv[min(j, nl - 1)]; if (j >= nl) q.w = 0; return q; };
    auto fetch = [&](const int4& q, float (&fx)[kKeep], float (&fy)[kKeep], float (&fz)[kKeep]) {
        const float* qx = cart1 + po + q.y; const float* qy = qx + cart_stride; const float* qz = qy + cart_stride;
#pragma unroll
        for (int k = 0; k < kKeep; k++) {
            const int i = lane + 64 * k;
            fx[k] = (i < q.w) ? qx[i] : __builtin_nanf(""); fy[k] = (i < q.w) ? qy[i] : 0.f; fz[k] = (i < q.w) ? qz[i] : 0.f;
        }
    };
    int4 q0 = info(j0), q1 = info(j0 + stride);
    float cx[kKeep], cy[kKeep], cz[kKeep];
    fetch(q0, cx, cy, cz);
    for (int j = j0; j < nl; j += stride) {
    const int4 q2 = info(j + 2 * stride);
    float nx[kKeep], ny[kKeep], nz[kKeep];
    fetch(q1, nx, ny, nz);
    const int4 qc = q0;
    float ux[kKeep], uy[kKeep], uz[kKeep];
#pragma unroll
    for (int k = 0; k < kKeep; k++) { ux[k] = cx[k]; uy[k] = cy[k]; uz[k] = cz[k]; cx[k] = nx[k]; cy[k] = ny[k]; cz[k] = nz[k]; }
    q0 = q1; q1 = q2;
    const int v = qc.x;
    const int m = qc.w;                                               // candidates of this bin (k_fit_cluster)
    if (m <= 0) continue;
    FitMid* mid = midD + (size_t)pair * V + v;
    const size_t base = po + qc.y;
    const float* qx = cart1 + base; const float* qy = qx + cart_stride; const float* qz = qy + cart_stride;
    // Sums in double over float addends: exact (or within 2^-53), so the order of the lanes does not matter -- the shared rule.
    double sumx = 0.0, sumy = 0.0, sumz = 0.0; int rows = 0;
#pragma unroll
    for (int k = 0; k < kKeep; k++) {
        if (ux[k] == ux[k]) { sumx += (double)ux[k]; sumy += (double)uy[k]; sumz += (double)uz[k]; rows++; }
    }
    for (int i = lane + 64 * kKeep; i < m; i += 64) {
        const float X = qx[i];
        if (X == X) { sumx += (double)X; sumy += (double)qy[i]; sumz += (double)qz[i]; rows++; }
    }
    {   // the four totals in one reduce-scatter (the row count as a double: exact)
        double t[4] = {sumx, sumy, sumz, (double)rows};
        const double mine = wave_total_scatter<4>(t);
        sumx = wave_total_scatter_get<4>(mine, 0); sumy = wave_total_scatter_get<4>(mine, 1); sumz = wave_total_scatter_get<4>(mine, 2);
        rows = (int)wave_total_scatter_get<4>(mine, 3);
    }
    if (rows * 3 < n) continue;                                       // src/icet.cpp:158 (size() counts coefficients); has_fit stays 0
    float mean[3], cov[6];
    {
#pragma clang fp contract(off)
        mean[0] = (float)sumx / (float)rows; mean[1] = (float)sumy / (float)rows; mean[2] = (float)sumz / (float)rows;
        double c[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < kKeep; k++) {
            if (ux[k] == ux[k]) {
                const float dx = ux[k] - mean[0], dy = uy[k] - mean[1], dz = uz[k] - mean[2];
                c[0] += (double)(dx * dx); c[1] += (double)(dx * dy); c[2] += (double)(dx * dz);
                c[3] += (double)(dy * dy); c[4] += (double)(dy * dz); c[5] += (double)(dz * dz);
            }
        }
        for (int i = lane + 64 * kKeep; i < m; i += 64) {
            const float X = qx[i];
            if (X == X) {
                const float dx = X - mean[0], dy = qy[i] - mean[1], dz = qz[i] - mean[2];
                c[0] += (double)(dx * dx); c[1] += (double)(dx * dy); c[2] += (double)(dx * dz);
                c[3] += (double)(dy * dy); c[4] += (double)(dy * dz); c[5] += (double)(dz * dz);
            }
        }
        const float den = (float)(rows - 1);
        double t[8] = {c[0], c[1], c[2], c[3], c[4], c[5], 0.0, 0.0};
        const double mine = wave_total_scatter<8>(t);
#pragma unroll
        for (int k = 0; k < 6; k++) cov[k] = (float)wave_total_scatter_get<8>(mine, k) / den;
    }
    if (lane == 0) {
        mid->mean[0] = mean[0]; mid->mean[1] = mean[1]; mid->mean[2] = mean[2];
#pragma unroll
        for (int k = 0; k < 6; k++) mid->cov[k] = cov[k];
        mid->has_fit = 1;
    }
    }   // live bins of this wave
}

// fitCells1's per-bin tail (src/icet.cpp:181-252), one lane per angular bin: eigen-decomposition, U = eigenvectors^T, the six
// sigma points and their inside test -> L, the scan-1 half of the gate at :290, and the records of the active voxels.
// kFinishBins bins per block: 512 for a throughput batch (75 x 24 is then 4 blocks per pair, 1024 per 256-pair launch: ONE resident round at the kernel's
// 127 VGPRs instead of two), 256 for small batches (more blocks for a single pair's CUs).
#ifndef ICET_FINISH_TRIP
#define ICET_FINISH_TRIP 256
#endif
constexpr int kFinishTrip = ICET_FINISH_TRIP;    // fitted bins per trip of k_fit_finish's heavy stage (<= kBlock; smaller values only to exercise the loop: scripts/cmp_libs.py)
static_assert(kFinishTrip <= kBlock, "one lane per fitted bin of a trip");
template <int kFinishBins>
__global__ __launch_bounds__(kBlock) void k_fit_finish(const FitMid* __restrict__ midD, const int32_t* __restrict__ bin_start, SlotHot* __restrict__ hotD, SlotFit* __restrict__ fitD,
                                                       int32_t* __restrict__ activeD, AuxDev aux, int T, int P, int n) {
    // Two stages.  (A) one lane per bin: the cheap per-bin outputs, and the FITTED bins of the block (~12 % of its 256) compacted into an
    // LDS list; (B) one lane per fitted bin: the 3x3 eigen-decomposition, the six sigma points with their double-precision
    // cartesianToSpherical, the slot records.  With the heavy part on the bin's own lane (round 2) every wave ran it for its handful of
    // fitted lanes; compacted, one wave per block does (k_fit_finish 53 -> see DESIGN.md section 4).
    __shared__ int s_list[kFinishBins];
    __shared__ int s_count;
    __shared__ float s_pt[kBlock][6][3];                          // (B2) sigma points of the block's fitted bins, by list position
    __shared__ float s_lim[kBlock][6];                            // az0, az1, el0, el1, inner, outer
    __shared__ unsigned char s_in[kBlock][6], s_far[kBlock][6];   // inside the bounds / r > outer, per point
    const int V = T * P;
    const int pair = blockIdx.y;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < kFinishBins / kBlock; kb++) {
        const int v = blockIdx.x * kFinishBins + kb * kBlock + (int)threadIdx.x;
        if (v < V) {
            const size_t o = (size_t)pair * V + v;
            const int cnt_v = bin_start[(size_t)pair * (V + 1) + v + 1] - bin_start[(size_t)pair * (V + 1) + v];
            FitMid m{};                                               // a bin below n rows was never visited by the fit kernels: nothing to read
            if (cnt_v >= n) m = midD[o];
            if (m.has_fit) s_list[atomicAdd(&s_count, 1)] = v;        // (the order of the list is irrelevant: every entry writes its own rows)
            else {
                activeD[o] = 0;
                if (aux.bounds) {
                    float az0, az1, el0, el1;
                    voxel_limits(v % T, v / T, T, P, az0, az1, el0, el1);
                    float* b = aux.bounds + o * 6; b[0] = az0; b[1] = az1; b[2] = el0; b[3] = el1; b[4] = m.inner; b[5] = m.outer;
                }
                if (aux.n1_raw) aux.n1_raw[o] = cnt_v;
                if (aux.has_fit) aux.has_fit[o] = 0;
                if (aux.mu1) { aux.mu1[o * 3] = m.mean[0]; aux.mu1[o * 3 + 1] = m.mean[1]; aux.mu1[o * 3 + 2] = m.mean[2]; }
                if (aux.sigma1) { float* sg = aux.sigma1 + o * 9; sg[0] = m.cov[0]; sg[1] = m.cov[1]; sg[2] = m.cov[2]; sg[3] = m.cov[1]; sg[4] = m.cov[3]; sg[5] = m.cov[4]; sg[6] = m.cov[2]; sg[7] = m.cov[4]; sg[8] = m.cov[5]; }
                if (aux.evecs1) for (int k = 0; k < 9; k++) aux.evecs1[o * 9 + k] = 0.f;
                if (aux.l_diag) { aux.l_diag[o * 3] = 0.f; aux.l_diag[o * 3 + 1] = 0.f; aux.l_diag[o * 3 + 2] = 0.f; }
                if (aux.test_points) for (int k = 0; k < 18; k++) aux.test_points[o * 18 + k] = 0.f;
            }
        }
    }
    __syncthreads();
    const int n_list = s_count;
    for (int base = 0; base < n_list; base += kFinishTrip) {           // one trip on ordinary grids (a block's 512 bins hold ~60 fitted ones); block-uniform
    const int n_here = min(kFinishTrip, n_list - base);
    const bool mine = (int)threadIdx.x < n_here;                  // this lane owns list entry base + threadIdx.x
    const int v = mine ? s_list[base + threadIdx.x] : 0;
    const size_t o = (size_t)pair * V + v;
    FitMid m{};
    if (mine) { m = midD[o]; m.cnt = bin_start[(size_t)pair * (V + 1) + v + 1] - bin_start[(size_t)pair * (V + 1) + v]; }
    const int theta = v % T, phi = v / T;
    float az0, az1, el0, el1;
    voxel_limits(theta, phi, T, P, az0, az1, el0, el1);
    const float inner = m.inner, outer = m.outer;
    float ev[3] = {0.f, 0.f, 0.f}, Vm[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float Ld[3] = {0.f, 0.f, 0.f};
    int active = 0;
    if (m.has_fit) {
        icetdev::eig3_sym(m.cov[0], m.cov[1], m.cov[3], m.cov[2], m.cov[4], m.cov[5], ev, Vm);
        // sigma points mu +- 2 sqrt(lambda_k) * (row k of V): rotated = diag(2 sqrt(lambda)) * U^T = diag(.) * V (src/icet.cpp:187-202).
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int k = j >> 1;
            float px, py, pz;
            {
#pragma clang fp contract(off)
                // rot = axislen * V (row k), rounded, THEN added to / subtracted from the mean (src/icet.cpp:193-202)
                const float al = 2.0f * sqrtf(ev[k]);
                const float r0 = al * Vm[3 * k], r1_ = al * Vm[3 * k + 1], r2_ = al * Vm[3 * k + 2];
                px = (j & 1) ? m.mean[0] - r0 : m.mean[0] + r0;
                py = (j & 1) ? m.mean[1] - r1_ : m.mean[1] + r1_;
                pz = (j & 1) ? m.mean[2] - r2_ : m.mean[2] + r2_;
            }
            s_pt[threadIdx.x][j][0] = px; s_pt[threadIdx.x][j][1] = py; s_pt[threadIdx.x][j][2] = pz;
        }
        s_lim[threadIdx.x][0] = az0; s_lim[threadIdx.x][1] = az1; s_lim[threadIdx.x][2] = el0; s_lim[threadIdx.x][3] = el1;
        s_lim[threadIdx.x][4] = inner; s_lim[threadIdx.x][5] = outer;
    }
    // (B2) cartesianToSpherical of the 6 x count sigma points (double-precision atan2 / acos each), one per lane
    __syncthreads();
    for (int item = threadIdx.x; item < 6 * n_here; item += kBlock) {
        const int q = item / 6, j = item - 6 * q;
        float r, az, el; c2s_cr(s_pt[q][j][0], s_pt[q][j][1], s_pt[q][j][2], r, az, el);
        s_in[q][j] = inside_bounds(r, az, el, s_lim[q][0], s_lim[q][1], s_lim[q][2], s_lim[q][3], s_lim[q][4], s_lim[q][5]) ? 1 : 0;
        s_far[q][j] = (r > s_lim[q][5]) ? 1 : 0;
    }
    __syncthreads();
    if (mine) {
    if (m.has_fit) {
        // testSigmaPoints walks j = 0..5 and leaves the loop AFTER testing the first point with r > outer (:669-686): a point behind
        // such a point was never tested, i.e. counts as outside.
        bool inside[6] = {false, false, false, false, false, false};
        bool done = false;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            if (!done) { inside[j] = s_in[threadIdx.x][j] != 0; done = s_far[threadIdx.x][j] != 0; }
        }
        if (aux.test_points) {          // `testPoints` (src/icet.cpp:213-231): the sigma points of every axis that is pruned
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const bool pruned = !(inside[j & ~1] || inside[j | 1]);
                float* tp = aux.test_points + (o * 6 + j) * 3;
                tp[0] = pruned ? s_pt[threadIdx.x][j][0] : 0.f; tp[1] = pruned ? s_pt[threadIdx.x][j][1] : 0.f; tp[2] = pruned ? s_pt[threadIdx.x][j][2] : 0.f;
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
    }   // mine
    __syncthreads();                                              // the staging arrays are reused by the next trip
    }   // trips over the block's fitted bins
}

// Dense per-voxel records -> compact slots in voxel order (phi-major, theta inner: the reference's
// accumulation order, src/icet.cpp:391-404).
constexpr int kCompactBlock = 1024;            // one block per pair walks its V voxels: 2 rounds of 3 barriers on 75 x 24 instead of 8
__global__ __launch_bounds__(kCompactBlock) void k_compact_slots(const SlotHot* __restrict__ hotD, const SlotFit* __restrict__ fitD, const int32_t* __restrict__ activeD,
                                                          SlotHot* __restrict__ hotS, SlotFit* __restrict__ fitS, int16_t* __restrict__ slot_of_voxel,
                                                          int32_t* __restrict__ n_slots, uint32_t* __restrict__ acc, uint32_t* __restrict__ near_over_count, int V) {
    __shared__ int wave_tot[kCompactBlock / 64];
    __shared__ int base;
    const int pair = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int v0 = 0; v0 < V; v0 += kCompactBlock) {
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
        if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < kCompactBlock / 64; k++) t += wave_tot[k]; base = b + t; }
        __syncthreads();
    }
    const int ns = base;
    if (threadIdx.x == 0) { n_slots[pair] = ns; near_over_count[pair] = 0u; }
    for (int i = threadIdx.x; i < ns * kAccWords; i += kCompactBlock) acc[(size_t)pair * V * kAccWords + i] = 0u;
}


// The property kLdsRank rests on, checked on THIS device: every wave adds 1 to counters its lanes pick (few / many distinct ones, runs of
// equal neighbours, packed 16-bit halves) and compares the value handed back with the ballot-computed number of earlier occurrences.
constexpr int kOrderClasses = 512;
__global__ __launch_bounds__(kBlock) void k_lds_order_selftest(int rounds, int32_t* __restrict__ bad) {
    __shared__ uint32_t cnt[kBlock / 64][kOrderClasses / 2];
    __shared__ uint16_t ref[kBlock / 64][kOrderClasses];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (kBlock / 64) * kOrderClasses / 2; i += kBlock) (&cnt[0][0])[i] = 0u;
    for (int i = threadIdx.x; i < (kBlock / 64) * kOrderClasses; i += kBlock) (&ref[0][0])[i] = 0;
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nbad = 0;
    for (int r = 0; r < rounds; r++) {
        uint32_t h = (uint32_t)(blockIdx.x * 7919 + r * 104729 + wave * 31) * 0x9E3779B9u; h ^= h >> 15;
        const int distinct = 1 << (h % 10u);                              // 1 .. 512 classes in play
        uint32_t x = (h + (uint32_t)((r & 1) ? (lane >> (h >> 8 & 3)) : lane)) * 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
        const uint32_t c = x % (uint32_t)distinct;
        unsigned long long peers = ~0ull;
#pragma unroll
        for (int q = 0; q < 9; q++) { const bool bit = (c >> q) & 1u; const unsigned long long m = __ballot(bit); peers &= bit ? m : ~m; }
        const uint32_t before = ref[wave][c];
        const uint32_t old = atomicAdd(&cnt[wave][c >> 1], 1u << (16u * (c & 1u)));
        if (((old >> (16u * (c & 1u))) & 0xFFFFu) != before + (uint32_t)__popcll(peers & lt)) nbad++;
        if ((peers & lt) == 0ull) ref[wave][c] = (uint16_t)(before + (uint32_t)__popcll(peers));
    }
    if (nbad) atomicAdd(bad, nbad);
}

}  // namespace

#define ICET_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t lds_rank_selftest(int32_t* d_scratch, hipStream_t st, int* ok) {
    *ok = 0;
    hipError_t e = hipMemsetAsync(d_scratch, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    k_lds_order_selftest<<<1024, kBlock, 0, st>>>(48, d_scratch);          // every CU four times over; <= 48 x 64 per counter: no carry out of a 16-bit half
    ICET_LAUNCH_CHECK();
    int32_t bad = -1;
    e = hipMemcpyAsync(&bad, d_scratch, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) *ok = (bad == 0);
    return e;
}

// dynamic LDS of the keyframe kernels, bytes: the voxel-indexed ones grow with the grid (validated against the device in ensure_workspace)
// k_scramble_src keeps the pair's exec bit table in LDS while histogram + table stay below this (0.75 M rows on a 75 x 24 grid)
constexpr size_t kScrambleLdsMax = 100 * 1024;
static size_t scan1_lds_bytes(const Workspace& w) { return (size_t)(w.lut_Mt + w.lut_Mp + 2) * sizeof(LutCell); }

hipError_t init_keyframe_kernels() {
    const int cap = 160 * 1024 - 4096;        // static __shared__ of the kernels comes on top (k_scan1_spherical: 8 B per rank-sort bucket + 32)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bin_scatter<false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bin_scatter<true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_scramble_src<true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_scramble_src<false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bin_hist), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan1_spherical), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_exec_flags_pair<ICET_EXEC_PAIR_ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    return e;
}

hipError_t launch_keyframe(const Workspace& w, const LaunchCfg& c, const AuxDev* auxp, hipStream_t st, const int32_t* d_n1) {
    AuxDev aux{}; if (auxp) aux = *auxp;
    // many small chunks per pair: with the XCD-aware mapping only ~4 pairs are then in flight per XCD (see decode_block)
    const int chunks = c.kf_chunks;
    const int groups = grid_groups(c.n_pairs);
    dim3 grid(groups * chunks), blk(kBlock);
    const int np = c.n_pairs;
    hipError_t e;
    int pbits = 0; while ((1 << pbits) < c.n_pairs) pbits++;
    int vbits = 1; while ((1 << vbits) < c.V) vbits++;
    const bool batch = c.n_pairs > 1;
    if (!c.use_library_sort) { e = launch_rank_sort_splitters(w, c, st, d_n1); if (e != hipSuccess) return e; }
    else {   // diagnostic path: no rank sort, hence nobody reduces the tiles' voxel ranges: the full range (vmin = 0, vmax = "large", clipped to V - 1 by the readers)
        if (d_n1) { e = launch_patch_counts(w, c, d_n1, nullptr, st); if (e != hipSuccess) return e; }
        e = hipMemset2DAsync(w.vrange, 8, 0x00, 4, c.n_pairs, st); if (e != hipSuccess) return e;
        e = hipMemset2DAsync(w.vrange + 1, 8, 0x7F, 4, c.n_pairs, st); if (e != hipSuccess) return e;
    }
    k_scan1_spherical<<<grid, blk, scan1_lds_bytes(w), st>>>(w.desc, w.r1, (batch && c.use_library_sort) ? w.key64A : nullptr, c.use_library_sort ? w.keyA : nullptr, w.valA, w.bin16,
                                                              c.T, c.P, np, chunks, c.use_library_sort ? nullptr : w.splitters, w.bkt, w.counts,
                                                              reinterpret_cast<const LutCell*>(w.lut), w.lut_Mt, w.lut_Mp, w.guard_t, w.guard_p, w.tile_vr, make_int4(w.zero_voxel[0], w.zero_voxel[1], w.zero_voxel[2], w.zero_voxel[3]), c.use_library_sort ? nullptr : w.zero_rows);
    ICET_LAUNCH_CHECK();
    if (c.stage_event && c.stage_at == 4) { e = hipEventRecord(c.stage_event, st); if (e != hipSuccess) return e; }
    if (c.use_library_sort) {
#ifdef ICET_DIAG_LIBSORT
        if (batch) e = sort_pairs_u64(w.sort_tmp, w.sort_tmp_bytes, w.key64A, w.key64B, w.valA, w.valB, c.total_n1, 32 + pbits, st);
        else e = sort_pairs_u32(w.sort_tmp, w.sort_tmp_bytes, w.keyA, w.keyB, w.valA, w.valB, c.total_n1, 32, st);
        if (e != hipSuccess) return e;
        // valB = s : original index of the row with rank i
        k_inverse_perm<<<grid, blk, 0, st>>>(w.desc, w.valB, w.pred, np, chunks);
        ICET_LAUNCH_CHECK();
#else
        (void)pbits; return hipErrorNotSupported;        // (icet_set_option refuses "library_sort" in a build without the diagnostic backend)
#endif
    } else {
        e = launch_rank_sort(w, c, st);        // valB = s, pred = s^-1  (icet_ranksort.hip)
        if (e != hipSuccess) return e;
    }
    if (c.stage_event && c.stage_at == 1) { e = hipEventRecord(c.stage_event, st); if (e != hipSuccess) return e; }
    if (c.use_library_sort) { e = hipMemsetAsync(w.flags, 0, sizeof(int32_t) * c.n_pairs, st); if (e != hipSuccess) return e; }      // (otherwise k_rs_splitters cleared the flags)
    if (c.true_sort) {
        // non-parity extension: the rows stay in sorted order (what the reference's comment says the loop is meant to do)
        e = hipMemcpyAsync(w.src, w.valB, sizeof(int32_t) * (size_t)c.total_n1, hipMemcpyDeviceToDevice, st); if (e != hipSuccess) return e;
        k_bin_hist<<<grid, blk, (size_t)c.V * 4, st>>>(w.desc, w.src, w.bin16, w.binpos, w.counts, w.flags, c.V, np, chunks, 1);
        ICET_LAUNCH_CHECK();
    } else {
        const int max_walk = 4096;
        {
            constexpr int kR = ICET_EXEC_PAIR_ROWS;
            const size_t pair_lds = (size_t)kExecPairThreads * kR + (size_t)((c.max_n1 + 63) / 64) * 8;    // state bytes + the bit table
            const bool fits = pair_lds <= (size_t)150 * 1024;
            const bool pairwise = fits && (c.exec_pairwise > 0 || (c.exec_pairwise < 0 && c.n_pairs >= 64));
            if (pairwise) k_exec_flags_pair<kR><<<c.n_pairs, kExecPairThreads, pair_lds, st>>>(w.desc, w.pred, w.execbits);
            else k_exec_flags<<<grid, blk, 0, st>>>(w.desc, w.pred, w.execbits, w.flags, max_walk, np, chunks);
        }
        ICET_LAUNCH_CHECK();
        const size_t hist_bytes = (size_t)((((c.V + 1) / 2 + 1) & ~1)) * 4, bit_bytes = (size_t)((c.max_n1 + 63) / 64) * 8;
        if (c.exec_bits_lds && hist_bytes + bit_bytes <= kScrambleLdsMax)
            k_scramble_src<true><<<grid, blk, hist_bytes + bit_bytes, st>>>(w.desc, w.valB, w.pred, w.execbits, w.src, w.flags, max_walk, w.bin16, w.binpos, w.counts, c.V, np, chunks, w.vrange);
        else
            k_scramble_src<false><<<grid, blk, hist_bytes, st>>>(w.desc, w.valB, w.pred, w.execbits, w.src, w.flags, max_walk, w.bin16, w.binpos, w.counts, c.V, np, chunks, w.vrange);
        ICET_LAUNCH_CHECK();
        k_scramble_replay<<<c.n_pairs, blk, (size_t)c.V * 4, st>>>(w.desc, w.valB, w.pred /* reused as scratch */, w.src, w.flags, w.bin16, w.binpos, w.counts, c.V, chunks);
        ICET_LAUNCH_CHECK();
        if (c.stage_event && c.stage_at == 2) { e = hipEventRecord(c.stage_event, st); if (e != hipSuccess) return e; }
    }
    e = launch_class_scan(w.counts, w.tile_base, w.bin_start, c.V, chunks, c.n_pairs, st, w.live_bins, w.n_live, c.n, w.fit_n_items, w.vrange);
    if (e != hipSuccess) return e;
    if (c.lds_rank) k_bin_scatter<true><<<grid, blk, (size_t)c.V * 12 + 8, st>>>(w.desc, w.src, w.binpos, w.tile_base, w.bin_start, w.valA, c.V, vbits, np, chunks, w.vrange);
    else k_bin_scatter<false><<<grid, blk, (size_t)c.V * 12 + 8, st>>>(w.desc, w.src, w.binpos, w.tile_base, w.bin_start, w.valA, c.V, vbits, np, chunks, w.vrange);
    ICET_LAUNCH_CHECK();
    if (c.stage_event && c.stage_at == 3) { e = hipEventRecord(c.stage_event, st); if (e != hipSuccess) return e; }
    // keyA / keyB (bucket-grouped keys and the overflow scratch of the rank sort) are dead by now: candidate rows and their r
    // (k_bin_scan zeroed the pairs' item counters; k_fit_finish takes a bin below n rows for empty without reading its record)
    FitItem* items = reinterpret_cast<FitItem*>(w.fit_items);
    // a fixed number of blocks per pair walks the pair's live bins / work items (their numbers are only known on the device):
    // enough blocks to fill the chip whatever the batch size
    const int fit_chunks = std::max(1, std::min((c.V + kBlock / 64 - 1) / (kBlock / 64), (ICET_FIT_BLOCKS + c.n_pairs - 1) / c.n_pairs));
    if (c.n_pairs <= 16)
        k_fit_cluster<16><<<dim3(groups * fit_chunks), blk, 0, st>>>(w.desc, w.bin_start, w.valA, w.r1, w.keyA, reinterpret_cast<float*>(w.keyB), items, w.fit_n_items,
                                                                    w.live_bins, w.n_live, w.midD, c.T, c.P, c.n, c.thresh, c.buff, np, fit_chunks, c.half_gap, c.use_library_sort ? nullptr : w.zero_rows, make_int4(w.zero_voxel[0], w.zero_voxel[1], w.zero_voxel[2], w.zero_voxel[3]));
    else
        k_fit_cluster<2><<<dim3(groups * fit_chunks), blk, 0, st>>>(w.desc, w.bin_start, w.valA, w.r1, w.keyA, reinterpret_cast<float*>(w.keyB), items, w.fit_n_items,
                                                                    w.live_bins, w.n_live, w.midD, c.T, c.P, c.n, c.thresh, c.buff, np, fit_chunks, c.half_gap, c.use_library_sort ? nullptr : w.zero_rows, make_int4(w.zero_voxel[0], w.zero_voxel[1], w.zero_voxel[2], w.zero_voxel[3]));
    ICET_LAUNCH_CHECK();
    {
        const int rt_chunks = std::max(1, std::min(64, (ICET_RT_BLOCKS + c.n_pairs - 1) / c.n_pairs));
        k_fit_roundtrip<<<dim3(groups * rt_chunks), blk, 0, st>>>(w.desc, w.bin_start, w.keyA, reinterpret_cast<const float*>(w.keyB), items, w.fit_n_items, w.cart1, (size_t)w.cap_n1,
                                                                  c.T, c.P, np, rt_chunks);
        ICET_LAUNCH_CHECK();
    }
    const int mom_chunks = std::max(1, std::min((c.V + kBlock / 64 - 1) / (kBlock / 64), (ICET_MOM_BLOCKS + c.n_pairs - 1) / c.n_pairs));
    k_fit_moments<<<dim3(groups * mom_chunks), blk, 0, st>>>(w.desc, w.bin_start, w.cart1, (size_t)w.cap_n1, w.live_bins, w.n_live, w.midD, c.V, c.n, np, mom_chunks);
    ICET_LAUNCH_CHECK();
    if ((long long)c.n_pairs * ((c.V + kBlock - 1) / kBlock) >= 2048) k_fit_finish<2 * kBlock><<<dim3((c.V + 2 * kBlock - 1) / (2 * kBlock), c.n_pairs), blk, 0, st>>>(w.midD, w.bin_start, w.hotD, w.fitD, w.activeD, aux, c.T, c.P, c.n);
    else k_fit_finish<kBlock><<<dim3((c.V + kBlock - 1) / kBlock, c.n_pairs), blk, 0, st>>>(w.midD, w.bin_start, w.hotD, w.fitD, w.activeD, aux, c.T, c.P, c.n);
    ICET_LAUNCH_CHECK();
    k_compact_slots<<<c.n_pairs, kCompactBlock, 0, st>>>(w.hotD, w.fitD, w.activeD, w.hotS, w.fitS, w.slot_of_voxel, w.n_slots, w.acc, w.near_over_count, c.V);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_class_scan(const uint32_t* counts, uint32_t* tile_base, int32_t* class_start, int n_classes, int chunks, int n_pairs, hipStream_t st,
                             int32_t* live, int32_t* n_live, int live_min, uint32_t* n_items, const int32_t* class_range, const int32_t* tile_vr, int32_t* vrange_out) {
    k_bin_tiles<<<dim3((n_classes + kTileClasses - 1) / kTileClasses, n_pairs), kBlock, 0, st>>>(counts, tile_base, class_start, n_classes, chunks, class_range);
    ICET_LAUNCH_CHECK();
    k_bin_scan<<<n_pairs, kScanBlock, 0, st>>>(class_start, n_classes, live, n_live, live_min, n_items, tile_vr, chunks, vrange_out);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace icet
