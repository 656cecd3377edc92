// icet_amd/csrc/icet_accumulate.hip -- the point pass of one Gauss-Newton iteration: the hot kernel of the path.
//
// ICET::fitScan2 (/root/reference/src/icet.cpp:372-436) up to the per-voxel sums: transform (:375-378), cartesianToSpherical
// (:387), sortSphericalCoordinates (:388), filterPointsInsideCluster (:299) and the sums behind mean / covariance (:303-306),
// 12 bytes of HBM traffic per scan-2 point per iteration and nothing N-sized written back.
#include <hip/hip_runtime.h>
#include <math.h>
#include "icet_internal.h"
#include "icet_device_common.h"
#include "icet_solve_body.h"
#include <algorithm>
#include <type_traits>

namespace icet {
namespace {

#ifndef ICET_ACC_BLOCK
#define ICET_ACC_BLOCK 512
#endif
#ifndef ICET_ACC_WAVES
#define ICET_ACC_WAVES 6
#endif
#ifndef ICET_ACC_PTS
#define ICET_ACC_PTS 4
#endif
static_assert(ICET_ACC_PTS == 4, "phase C of k_gn_accumulate is written for 4 consecutive points per lane");
constexpr int kAccPts = ICET_ACC_PTS;                      // consecutive points per lane per trip (two dwordx4 loads per coordinate)
constexpr int kAccBlock = ICET_ACC_BLOCK;        // k_gn_accumulate: the waves of a block share one copy of the pair's LDS tables
constexpr int kAccWavesPerSimd = ICET_ACC_WAVES; // register budget: 6 -> 84 VGPRs, no spills, three 512-thread blocks per CU (with 1536 blocks per 256-pair launch: 130 -> 121 us); 8 spills
constexpr int kHotWords = 8;                     // LDS record of an active voxel in k_gn_accumulate: inner, outer, mu1 (5 words) at a 32-byte stride: ds_read_b128 + ds_read_b32, address by shift

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
// Partial sums of a slot that has no LDS row go straight to HBM.  Kept out of line so that the LDS update in the hot loop
// stays a ds_add_* (a select between the two pointers would turn both into flat atomics).
__device__ __noinline__ void spill_flush(uint32_t* A, uint32_t nraw, uint32_t nin, float S0, float S1, float S2, float S3, float S4,
                                         float S5, float S6, float S7, float S8) {
    acc_add_hbm(A, nraw, nin, S0, S1, S2, S3, S4, S5, S6, S7, S8);
}

// A flush whose values may leave the range of the two-instruction conversion (to_fix_biased): exact wide conversion, same bias.
__device__ __noinline__ void flush_wide(unsigned long long* F, float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float a8) {
    atomicAdd(&F[1], to_fix_wide_biased(a0)); atomicAdd(&F[2], to_fix_wide_biased(a1)); atomicAdd(&F[3], to_fix_wide_biased(a2)); atomicAdd(&F[4], to_fix_wide_biased(a3));
    atomicAdd(&F[5], to_fix_wide_biased(a4)); atomicAdd(&F[6], to_fix_wide_biased(a5)); atomicAdd(&F[7], to_fix_wide_biased(a6)); atomicAdd(&F[8], to_fix_wide_biased(a7));
    atomicAdd(&F[9], to_fix_wide_biased(a8));
}

constexpr int kCntBits = 21;            // LDS count word: three 21-bit fields
constexpr unsigned long long kCntMask = (1ull << kCntBits) - 1ull;
constexpr uint32_t kNearCap = 512;      // undecided points a block of a THROUGHPUT batch parks in LDS (2 KB); the rest goes to the per-pair overflow list in HBM
constexpr uint32_t kNearCapSmall = 2048;  // small batches (a block has most of a CU's LDS): a whole 2048-point chunk -- the thousands of exact-zero rows of a real scan are
                                        // undecided points in the first iteration, and the overflow list is drained by ONE block of k_gn_solve (36 - 62 us on the reference's sample pairs)

typedef __attribute__((address_space(1))) const float gfloat;
typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const vfloat4 gfloat4;

// kRT2 (ICET_FLAG_ROUNDTRIP_SCAN2, a parity-study option): an in-bounds point enters the sums as sphericalToCartesian(cartesianToSpherical(q))
// (src/icet.cpp:303) -- a double-precision atan2 + acos inside the loop, in a kernel of its own so that the default kernel keeps its registers.
// kSmall: a small batch's block (one per CU, nearly all of its LDS) parks up to kNearCapSmall undecided points instead of kNearCap -- a template
// parameter, not an argument: one more live scalar in this loop costs SGPR spills and, through them, 25 spilled VGPRs (measured: +5 % per launch).
// What the kernel with the keep list (kKeep) takes on top: the masks it writes in a full pass, the list it walks otherwise, the per-pair state, the bin edges in
// the stand-in coordinates, and the margin constants of keep_margin below.
struct KeepDev { unsigned long long* mask; const uint32_t* list; const float* edges; const int32_t* modes; float c1, c2, capT, capP, invT; };
// Which pair the blocks of slot `s` take, and what they do with it (KeepArgs::modes_next).  Within each group of 256 slots: the pairs that walk their whole scan
// first, then the list pairs from the longest list to the shortest (eight classes of list length between the group's extremes, index order inside a class).  Blocks
// start in slot order and a launch is about two rounds of blocks, so what starts LAST should be SHORT: with the list pairs in index order the launch ended 15-20 us
// after its average block slot had run dry (list lengths differ by 2x between pairs).  Every wave works the order out for itself from the group's mode words --
// one coalesced read of 4 words per lane, then ballots on registers -- no LDS, no atomics; which block takes which pair never shows in the bits.
__device__ __forceinline__ int keep_pair_of_slot(const int32_t* __restrict__ modes, int n_pairs, int s, int& mode_word) {
    const int lane = (int)(threadIdx.x & 63u);
    const int gb = s & ~255, ng = min(256, n_pairs - gb), sl = s - gb;
    int m[4];
#pragma unroll
    for (int c = 0; c < 4; c++) m[c] = (c * 64 + lane < ng) ? modes[gb + c * 64 + lane] : -1;      // 0 whole scan, n_keep + 1 list, -1 no such pair
    int hi = 0, lo = 0x7FFFFFFF;
#pragma unroll
    for (int c = 0; c < 4; c++) { hi = max(hi, m[c]); lo = min(lo, m[c] > 0 ? m[c] : 0x7FFFFFFF); }
    hi = wave_reduce_max_i(hi); lo = wave_reduce_min_i(lo);
    const float scale = 7.999f / (float)max(hi - lo + 1, 1);
    int cls[4], cnt[9];
#pragma unroll
    for (int k = 0; k < 9; k++) cnt[k] = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        cls[c] = m[c] < 0 ? 9 : (m[c] == 0 ? 0 : 1 + (int)((float)(hi - m[c]) * scale));            // 1 = the longest lists ... 8 = the shortest
#pragma unroll
        for (int k = 0; k < 9; k++) cnt[k] += __popcll(__ballot(cls[c] == k));
    }
    int k_of = 0, r = sl;                                             // the slot's class and its rank inside it
#pragma unroll
    for (int k = 0; k < 8; k++) { const bool past = (k_of == k) & (r >= cnt[k]); r -= past ? cnt[k] : 0; k_of += past ? 1 : 0; }
    int found = -1, word = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const unsigned long long bm = __ballot(cls[c] == k_of);
        const int rk = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
        const unsigned long long hit = __ballot(cls[c] == k_of && rk == r);
        if (found < 0 && hit != 0ull) { const int l = __builtin_ctzll(hit); found = gb + c * 64 + l; word = __builtin_amdgcn_readlane(m[c], l); }
        r -= __popcll(bm);
    }
    mode_word = word;
    return found;
}
constexpr int kKeepOwn = 63;            // list pass: entries a wave-trip owns (lanes 1..63); lane 0 repeats the last entry of the previous trip (the "ghost")

template <bool kVec4, bool kRT2, bool kSmall, bool kKeep>
__device__ __forceinline__ void gn_accumulate_body(const PairDesc* __restrict__ desc, const float* __restrict__ xf_all,
                                                          const int16_t* __restrict__ slot_of_voxel, const int32_t* __restrict__ n_slots,
                                                          const SlotHot* __restrict__ hotS, uint32_t* __restrict__ acc,
                                                          const float* __restrict__ thr, const LutCell* __restrict__ lut,
                                                          int T, int P, int Mt, int Mp, float guard_t, float guard_p,
                                                          int lds_slots, int chunks, int n_pairs, int force_exact,
                                                          uint32_t* __restrict__ near_over, uint32_t* __restrict__ near_over_count, const KeepDev& kd) {
    constexpr uint32_t near_cap = kSmall ? kNearCapSmall : kNearCap;
    constexpr int kW = kAccBlock / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int V = T * P;
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;
    [[maybe_unused]] int mode_word = 0;
    if constexpr (kKeep) { pair = keep_pair_of_slot(kd.modes, n_pairs, pair, mode_word); if (pair < 0) return; }      // whole-scan pairs first
    const PairDesc d = desc[pair];
    int cs = (d.n2 + chunks - 1) / chunks;
    cs = (cs + kAccPts * 64 - 1) / (kAccPts * 64) * (kAccPts * 64);   // whole WAVE-trips (256 points): chunks of equal size whatever the block's trip; a block's last trip may leave waves idle
    const int begin = chunk * cs;
    // keep list (KeepState, icet_internal.h): a pair in list mode is walked in wave-trips of kKeepOwn list entries, trips [tau_begin, tau_end) for this block
    int list_mode = 0, nk = 0, tau_begin = 0, tau_end = 0;
    if constexpr (kKeep) { list_mode = mode_word > 0; nk = mode_word - 1; }
    if (kKeep && list_mode) {
        const int n_trips = nk / kKeepOwn + 1;                            // (the last trip may own nothing: it then only finishes the previous trip's last entry)
        int tpb = (n_trips + chunks - 1) / chunks; tpb = (tpb + kW - 1) / kW * kW;
        tau_begin = chunk * tpb; tau_end = min(n_trips, tau_begin + tpb);
        if (tau_begin >= n_trips) return;
    } else if (begin >= d.n2) return;
    const int end = (kKeep && list_mode) ? d.n2 : min(d.n2, begin + cs);
    const int lane = (int)(threadIdx.x & 63u);
    // list pass: lane l of trip tau holds list entry 63 tau - 1 + l.  The list is read UNCONDITIONALLY at a clamped index and the "no such entry" case is selected a
    // trip later, when the value is used: a load under a condition is followed by its select, i.e. by a wait for the load itself and for every point load requested
    // before it.  The first two trips' entries are requested HERE, in front of the block's table set-up, so that the first point loads can go out right behind it.
    const uint32_t* L = kd.list + ((size_t)d.off2 >> 2) + pair;
    const int e_max = max(nk - 1, 0);
    auto entry_raw = [&](int tau) -> int { return (int)L[min(max(kKeepOwn * tau - 1 + lane, 0), e_max)]; };
    auto entry_ok = [&](int tau) -> bool { return (tau < tau_end) & ((unsigned)(kKeepOwn * tau - 1 + lane) < (unsigned)nk); };
    [[maybe_unused]] int gc_first = -1, gn_first = -1;
    if (kKeep && list_mode) { gc_first = entry_raw(tau_begin + (int)(threadIdx.x >> 6)); gn_first = entry_raw(tau_begin + (int)(threadIdx.x >> 6) + kW); }

    LutCell* lut_t = reinterpret_cast<LutCell*>(smem);                    // Mt + 1 cells (the spare one catches pa == 4)
    LutCell* lut_p = lut_t + (Mt + 1);                                    // Mp + 1 cells (w == 1)
    // per slot 10 x u64: [raw | in << 21 | conversions << 42] (a block sees < 2^21 points: launch_gn_accumulate), then the 9
    // fixed-point sums with the conversion bias still in them (to_fix_biased) -- constant offsets inside a flush, one 64-bit add
    // for the counts, two VALU instructions per value; the bias comes out once per slot when the block hands its sums to HBM
    unsigned long long* lacc = reinterpret_cast<unsigned long long*>(lut_p + (Mp + 1));
    // lds_slots x 8 floats, 16-byte aligned (one ds_read_b128 + one ds_read_b96 per point): the four radial thresholds of radial_zones, mu1, pad
    float* hot = reinterpret_cast<float*>(smem + (((size_t)(Mt + Mp + 2) * sizeof(LutCell) + (size_t)80 * lds_slots + 15) & ~(size_t)15));
    int16_t* map = reinterpret_cast<int16_t*>(hot + lds_slots * kHotWords);
    const int map_words = (V + 1) / 2;
    uint32_t* nearq = reinterpret_cast<uint32_t*>(map) + (V + T + 4) / 2;   // near_cap point indices, then the fill counter
    // keep masks (full pass of a kKeep kernel): bin edges in pa / w units (T + 2 and P + 2 floats) and, per voxel, which of {itself, its lower-azimuth
    // neighbour, its higher-azimuth neighbour} is active (bits 0..2), padded by T zeros in front and 2 T behind so that the rows above / below index inside
    [[maybe_unused]] float* k_edge_t = reinterpret_cast<float*>(nearq + near_cap + 5);
    [[maybe_unused]] float* k_edge_p = k_edge_t + (T + 2);
    [[maybe_unused]] uint8_t* k_nb = reinterpret_cast<uint8_t*>(k_edge_p + (P + 2)) + T;
    const int ns = n_slots[pair];
    const int nl = min(ns, lds_slots);
    const SlotHot* hs = hotS + (size_t)pair * V;
    {
        // The block's tables come from L2 through registers with every read of a thread in flight at once: the first two trips of the
        // voxel map, the first three of the LUTs and the first hot record are requested before anything is stored (written as plain
        // copy loops they compile to load - wait - store per trip: six dependent round trips in front of the block's first point).
        const uint32_t* gm = reinterpret_cast<const uint32_t*>(slot_of_voxel + (size_t)pair * ((V + 1) & ~1));   // rows padded to even length
        uint32_t* lm = reinterpret_cast<uint32_t*>(map);
        const uint2* gl = reinterpret_cast<const uint2*>(lut);
        uint2* ll = reinterpret_cast<uint2*>(lut_t);
        const int n_lut = Mt + Mp + 2;
        constexpr int kMapAhead = 2, kLutAhead = 3;
        uint32_t mreg[kMapAhead]; uint2 lreg[kLutAhead];
#pragma unroll
        for (int k = 0; k < kMapAhead; k++) { const int i = (int)threadIdx.x + k * kAccBlock; mreg[k] = (i < map_words) ? gm[i] : 0u; }
#pragma unroll
        for (int k = 0; k < kLutAhead; k++) { const int i = (int)threadIdx.x + k * kAccBlock; lreg[k] = (i < n_lut) ? gl[i] : make_uint2(0u, 0u); }
        SlotHot g0{};
        if ((int)threadIdx.x < nl) g0 = hs[threadIdx.x];
#pragma unroll
        for (int k = 0; k < kMapAhead; k++) { const int i = (int)threadIdx.x + k * kAccBlock; if (i < map_words) lm[i] = mreg[k]; }
        for (int i = (int)threadIdx.x + kMapAhead * kAccBlock; i < map_words; i += kAccBlock) lm[i] = gm[i];
        for (int i = 2 * map_words + threadIdx.x; i < V + T + 2; i += kAccBlock) map[i] = (int16_t)-1;   // bt == T or bp == P land here
#pragma unroll
        for (int k = 0; k < kLutAhead; k++) { const int i = (int)threadIdx.x + k * kAccBlock; if (i < n_lut) ll[i] = lreg[k]; }
        for (int i = (int)threadIdx.x + kLutAhead * kAccBlock; i < n_lut; i += kAccBlock) ll[i] = gl[i];
        for (int i = threadIdx.x; i < nl; i += kAccBlock) {
            const SlotHot g = (i == (int)threadIdx.x) ? g0 : hs[i];
            float* h = hot + i * kHotWords;
            h[0] = g.inner; h[1] = g.outer; h[2] = g.mu[0]; h[3] = g.mu[1]; h[4] = g.mu[2];
        }
        for (int i = threadIdx.x; i < 10 * nl; i += kAccBlock) lacc[i] = 0ull;   // only the rows in use
        if (threadIdx.x < 5) nearq[near_cap + threadIdx.x] = 0u;      // the queue's fill counter, then the four counters of points that land exactly on the origin
    }
    const float* xf = xf_all + pair * kXf;
    const float tx = xf[0], ty = xf[1], tz = xf[2];
    const float R00 = xf[3], R01 = xf[4], R02 = xf[5], R10 = xf[6], R11 = xf[7], R12 = xf[8], R20 = xf[9], R21 = xf[10], R22 = xf[11];
    const float cell_t = (float)Mt * 0.25f, cell_p = (float)Mp * 0.5f;
    __syncthreads();
    if constexpr (kKeep) {
        if (!list_mode) {                                                   // block-uniform
            for (int i = threadIdx.x; i < T + P + 4; i += kAccBlock) {
                const int k = i < T + 2 ? i : i - (T + 2);
                k_edge_t[i] = (i < T + 2) ? kd.edges[min(k, T)] : kd.edges[T + 1 + min(k, P)];      // (k_edge_p follows k_edge_t)
            }
            for (int v = (int)threadIdx.x - T; v < V + 2 * T + 1; v += kAccBlock) {
                uint32_t code = 0u;
                if (v >= 0 && v < V) {
                    const int bt = v % T, r0 = v - bt;
                    code = (map[v] >= 0 ? 1u : 0u) | (map[r0 + (bt == 0 ? T - 1 : bt - 1)] >= 0 ? 2u : 0u) | (map[r0 + (bt == T - 1 ? 0 : bt + 1)] >= 0 ? 4u : 0u);
                }
                k_nb[v] = (uint8_t)code;
            }
            __syncthreads();
        }
    }

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
#ifndef ICET_ACC_PHASE
#define ICET_ACC_PHASE 9       // TIMING BUILDS ONLY (profiles/r03_acc_phases.txt; results are wrong below 9): 0 loads, 1 + classification, 2 + hot records, 3 + parked points, 4 + sums without conversion / LDS atomics
#endif
    [[maybe_unused]] float sink = 0.f;
    float XN[4], YN[4], ZN[4];
    // One wave-trip: the lane's group of 4 consecutive points starting at i0 (its raw coordinates are in XN / YN / ZN), the next group's loads issued
    // behind the transform.  kMode 0: the plain pass; 1: a full pass that also writes the keep mask of its 256 points; 2: a pass over the keep
    // list -- `grp` is the lane's group index in the scan (-1: none), lane 0 the ghost of the previous trip's last entry.
    auto trip = [&](auto mode_c, const int i0, const bool have_next, const int i0_next, [[maybe_unused]] const int grp) {
      constexpr int kMode = decltype(mode_c)::value;
      [[maybe_unused]] const bool ghost = kMode == 2 && lane == 0;
      auto flush = [&](int slot, uint32_t cr, uint32_t ci, float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float a8) {
          if (ICET_ACC_PHASE == 4) { sink += (float)(slot + (int)cr + (int)ci) + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8; return; }
          if (slot >= 0) {
              if (slot < nl) {
                  unsigned long long* F = lacc + slot * 10;
                  atomicAdd(&F[0], (unsigned long long)cr | ((unsigned long long)ci << kCntBits) | ((ci ? 1ull : 0ull) << (2 * kCntBits)));
                  if (ci) {
                      // the two-instruction conversion holds below 2^15 m^2; the three squared sums bound the other six values (kFixFastMax).
                      // Wave-uniform test: ordinary grids never leave the first branch (coarse grid x long range does: ADVICE r2)
                      if (__ballot(!(fmaxf(fmaxf(a3, a6), a8) < kFixFastMax)) == 0ull) {
                          atomicAdd(&F[1], to_fix_biased(a0)); atomicAdd(&F[2], to_fix_biased(a1)); atomicAdd(&F[3], to_fix_biased(a2)); atomicAdd(&F[4], to_fix_biased(a3));
                          atomicAdd(&F[5], to_fix_biased(a4)); atomicAdd(&F[6], to_fix_biased(a5)); atomicAdd(&F[7], to_fix_biased(a6)); atomicAdd(&F[8], to_fix_biased(a7));
                          atomicAdd(&F[9], to_fix_biased(a8));
                      } else {
                          flush_wide(F, a0, a1, a2, a3, a4, a5, a6, a7, a8);
                      }
                  }
              } else {
                  spill_flush(gacc + (size_t)slot * kAccWords, cr, ci, a0, a1, a2, a3, a4, a5, a6, a7, a8);
              }
          }
      };
      {
        PointClass pc[4];
        float QX[4], QY[4], QZ[4], RR[4];
        // == transform_point (icet_device_common.h), on the scalars already in registers: the deferred literal path must see the same bits.
        // Done for all 4 points BEFORE the next group's loads are issued into the same registers (no copy of the 12 prefetched words).
        {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 q[6];                                                      // (x, y, z) of points 0-1, then of points 2-3: one packed instruction per two points
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const f2 a = f2{XN[2 * h], XN[2 * h + 1]} + f2{tx, tx}, b = f2{YN[2 * h], YN[2 * h + 1]} + f2{ty, ty}, c = f2{ZN[2 * h], ZN[2 * h + 1]} + f2{tz, tz};
                q[3 * h + 0] = __builtin_elementwise_fma(c, f2{R20, R20}, __builtin_elementwise_fma(b, f2{R10, R10}, a * f2{R00, R00}));
                q[3 * h + 1] = __builtin_elementwise_fma(c, f2{R21, R21}, __builtin_elementwise_fma(b, f2{R11, R11}, a * f2{R01, R01}));
                q[3 * h + 2] = __builtin_elementwise_fma(c, f2{R22, R22}, __builtin_elementwise_fma(b, f2{R12, R12}, a * f2{R02, R02}));
            }
            // (the empty asm pins the order: the compiler would otherwise hoist the loads and copy the 12 registers first)
            asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]) : : "memory");
#pragma unroll
            for (int h = 0; h < 2; h++) { QX[2 * h] = q[3 * h].x; QX[2 * h + 1] = q[3 * h].y; QY[2 * h] = q[3 * h + 1].x; QY[2 * h + 1] = q[3 * h + 1].y; QZ[2 * h] = q[3 * h + 2].x; QZ[2 * h + 1] = q[3 * h + 2].y; }
        }
        if (have_next) load4(i0_next, XN, YN, ZN);
        if (ICET_ACC_PHASE == 0) { sink += (QX[0] + QX[1] + QX[2] + QX[3]) + (QY[0] + QY[1] + QY[2] + QY[3]) + (QZ[0] + QZ[1] + QZ[2] + QZ[3]); return; }
        int SM[4];
        bool nr[4];
        float R2[4];
        [[maybe_unused]] bool keep_any = false;
        // ---- phase A: angular classification of the 4 points.  Straight-line code (bitwise | and &, no clamps: the
        // LUTs carry one spare cell and the map T+1 spare entries, so even a NaN or pa == 4 indexes inside LDS) so
        // that the four look-up chains overlap. ----
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float qx = QX[j], qy = QY[j], qz = QZ[j];
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
            R2[j] = r2;
            if constexpr (kMode == 1) {
                // keep_margin.  Can this point reach the angular bin of an ACTIVE voxel while X stays within the budgets of the list (|t - t_ref| <= bt,
                // |R - R_ref|_F <= br)?  With q = R^T (p + t): |q' - q| <= br |q| + bt =: dmax, so its polar angle moves by at most asin(dmax / |q|) and its azimuth by at
                // most asin(dmax / rho), rho the horizontal range; for ratios <= 0.5, asin(x) <= 1.05 x: Mp = c1 + c2 / |q|, Mt = (c1 |q| + c2) / rho with
                // c1 = 1.05 br + 1e-4 (the slack covers every rounding in sight), c2 = 1.05 bt.  The stand-ins move LESS than the angles (|d pa / d theta| <= 1,
                // |d w / d phi| <= 1), so "farther than M from an edge in pa / w" implies "farther than M in angle".  The point is kept if its own bin, or a
                // neighbour it is within M of (3 x 3, azimuth wrapping), is active -- or if nothing can be said: M beyond a bin width or 0.5 rad (close to the
                // sensor or to the pole axis), a guard-band / out-of-range point, an ambiguous cell.
                const float irho = __builtin_amdgcn_rsqf(fmaf(qx, qx, qy * qy));
                const float Mpol = fmaf(kd.c2, rs, kd.c1), Maz = fmaf(kd.c1, RR[j], kd.c2) * irho;
                const float lo_t = k_edge_t[bt], hi_t = k_edge_t[bt + 1];
                const int bpi = static_cast<int>(fmaf((float)row, kd.invT, 0.5f));
                const float lo_p = k_edge_p[bpi], hi_p = k_edge_p[bpi + 1];
                const bool nlt = !((pa - lo_t) >= Maz), nht = !((hi_t - pa) >= Maz), nlp = !((w - lo_p) >= Mpol), nhp = !((hi_p - w) >= Mpol);
                const bool unsure = !(Maz <= kd.capT) | !(Mpol <= kd.capP) | (bt >= T) | (row >= V) | nr[j] | !((r2 >= kR2Min) & (r2 <= kR2Max));
                const int v = row + bt;
                const uint32_t c0 = k_nb[v], cm = k_nb[v - T], cp = k_nb[v + T];
                const uint32_t msk = 1u | (nlt ? 2u : 0u) | (nht ? 4u : 0u);
                const uint32_t hit = (c0 & msk) | (nlp ? (cm & msk) : 0u) | (nhp ? (cp & msk) : 0u);
                keep_any |= (i0 + j < end) & (unsure | (hit != 0u));
            }
        }
        if constexpr (kMode == 1) {
            const unsigned long long km = __ballot(keep_any);
            if (lane == 0 && i0 < end) kd.mask[((size_t)d.off2 >> 8) + pair + (i0 >> 8)] = km;      // lane 0's group starts an aligned block of 256 points
        }
        // |q|^2 outside [1e-30, 1e30] over- or underflows the stand-in coordinates while the literal formulas stay well defined
        // (absurd inputs, but the claim is "never decides differently"): literal path.  PER POINT since round 5: one min / max over the lane's 4 points sent
        // the three NEIGHBOURS of every exact-zero row through the literal path as well, and a real scan has 8 - 18 % of such rows, scattered: in a first
        // iteration with X0 = 0 a third of all points overflowed the queues (339 + 86 us for that iteration of a 256-pair batch of real scans)
#pragma unroll
        for (int j = 0; j < 4; j++) nr[j] |= !((R2[j] >= kR2Min) & (R2[j] <= kR2Max));
        // ---- phase A2: only waves that touch an active voxel look at the hot records (radial test, d = q - mu1) ----
        // E: the point's slot, -1 for "no active voxel" and for a point that waits for the literal formulas
        const int E0 = nr[0] ? -1 : SM[0], E1 = nr[1] ? -1 : SM[1], E2 = nr[2] ? -1 : SM[2], E3 = nr[3] ? -1 : SM[3];
        if (ICET_ACC_PHASE == 1) { sink += (float)(E0 + E1 + E2 + E3) + RR[0] + RR[1] + RR[2] + RR[3] + QX[0] + QY[1] + QZ[2]; return; }
        if (__ballot((E0 & E1 & E2 & E3) >= 0) != 0ull) {                  // some lane holds a point with E >= 0 (the sign bits do not all agree on "negative")
            // A slot beyond the LDS table (more active voxels than lds_slots; the launch sizes the table so that this is rare) keeps
            // its record in HBM.  Its points stay in the lane's runs like any other -- how the sums are grouped must not depend on a
            // launch-shape knob, or the knob would show in the result bits -- so the record is fetched here, in a wave-uniform branch
            // that ordinary waves skip (every LDS access stays a ds_* instruction, every HBM access a global_*).
            const bool beyond = __ballot(max(max(E0, E1), max(E2, E3)) >= nl) != 0ull;
            const int E[4] = {E0, E1, E2, E3};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int sm = E[j];
                const bool has = sm >= 0;
                const float* h = hot + min(max(sm, 0), lds_slots - 1) * kHotWords;
                float inner = h[0], outer = h[1], m0 = h[2], m1 = h[3], m2 = h[4];
                if (beyond) {
                    if (sm >= nl) { const SlotHot g = hs[sm]; inner = g.inner; outer = g.outer; m0 = g.mu[0]; m1 = g.mu[1]; m2 = g.mu[2]; }
                    // The record must have ARRIVED before this rare branch rejoins the common path: vmcnt counts loads in issue order, so a
                    // wait for these five words placed after the join (where the compiler puts it: at the first use) is `s_waitcnt vmcnt(0)`
                    // on EVERY trip -- a wait for the next trip's point loads as well.
                    asm volatile("" : "+v"(inner), "+v"(outer), "+v"(m0), "+v"(m1), "+v"(m2));
                }
                // |q| against the cluster's radial bounds (filterPointsInsideCluster, src/icet.cpp:299).  r = r2 * rsq(r2) is good to a few
                // ulps, so a point within 1e-6 r of a bound waits for the literal formulas.  With m = min(r - inner, outer - r), the signed
                // distance to the nearer bound (inner <= outer), "near a bound" is |m| < guard and "inside" is m >= 0.
                const float r = RR[j];
                const float gr = 1e-6f * r;
                const float m = fminf(r - inner, outer - r);
                const bool edge = has & !(fabsf(m) >= gr);
                nr[j] |= edge;
                pc[j].s = edge ? -1 : sm;
                pc[j].inb = has & (m >= 0.f);
                float ux = QX[j], uy = QY[j], uz = QZ[j];
                if (kRT2) { if (pc[j].inb & !edge) roundtrip_any(ux, uy, uz, ux, uy, uz); }
                pc[j].dx = ux - m0; pc[j].dy = uy - m1; pc[j].dz = uz - m2;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) { pc[j].s = -1; pc[j].inb = false; pc[j].dx = pc[j].dy = pc[j].dz = 0.f; }
        }
        // ---- phase B (rare): a point within a guard band of a voxel edge must be classified with the literal formulas (double-
        // precision atan2 / acos).  Not here: a call or that much code inside this loop costs every trip registers and scratch
        // traffic (measured: 119 -> 238 us per launch).  The point's INDEX is parked in the block's LDS queue and classified
        // after the loop, as a run of one; past kNearCap entries (adversarial input, or the force_exact diagnostic) it goes to
        // the pair's overflow list in HBM, which k_gn_solve drains.  Integer accumulation makes the order irrelevant. ----
        if (ICET_ACC_PHASE == 2) { for (int j = 0; j < 4; j++) sink += (float)pc[j].s + pc[j].dx + pc[j].dy + pc[j].dz + (pc[j].inb ? 1.f : 0.f) + (nr[j] ? 1.f : 0.f); return; }
        if (__ballot(nr[0] | nr[1] | nr[2] | nr[3]) != 0ull) {
            // A point that lands EXACTLY on the origin is not parked: r = 0, theta = atan2(+-0, +-0), phi = acos(NaN) -> 1000 (src/utils.cpp:103-116), so
            // its voxel is a function of the two sign bits and it can never pass the bounds test (phi = 1000): such points are only COUNTED, per sign
            // pattern, and the block adds the counts to the voxels' raw counts at its end.  Why it matters: the invalid returns of a real scan are
            // exact-zero rows (5 k - 24 k per scan of the reference's sample data), and in a first iteration with X0 = 0 every one of them lands on
            // the origin; parked, they filled the queue, the overflow list (one global atomic each on ONE counter) and k_gn_solve's one-block drain:
            // 600 + 284 us instead of 100 + 13 for that iteration of a 256-pair batch of real scans.
            if (__ballot((R2[0] == 0.f) | (R2[1] == 0.f) | (R2[2] == 0.f) | (R2[3] == 0.f)) != 0ull) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool zq = (R2[j] == 0.f) & (QX[j] == 0.f) & (QY[j] == 0.f) & (i0 + j < end);      // x = y = +-0 exactly (not a tiny vector whose squares underflow: its theta is a real angle)
                    const bool zc = zq & !ghost;                                                         // (the ghost's points are counted and parked by the lane that owns them)
                    const int patt = (__builtin_signbit(QY[j]) ? 2 : 0) | (__builtin_signbit(QX[j]) ? 1 : 0);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const unsigned long long m = __ballot(zc & (patt == k));
                        if (m != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&nearq[near_cap + 1 + k], (uint32_t)__popcll(m));
                    }
                    nr[j] &= !zq;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (nr[j] & (i0 + j < end) & !ghost) {
                    const uint32_t e = atomicAdd(&nearq[near_cap], 1u);
                    if (e < near_cap) nearq[e] = (uint32_t)(i0 + j);
                    else near_over[(size_t)d.off2 + atomicAdd(&near_over_count[pair], 1u)] = (uint32_t)(i0 + j);
                }
            }
        }
        if (ICET_ACC_PHASE == 3) { for (int j = 0; j < 4; j++) sink += (float)pc[j].s + pc[j].dx + pc[j].dy + pc[j].dz + (pc[j].inb ? 1.f : 0.f); return; }
        const int s0 = pc[0].s, s1 = pc[1].s, s2 = pc[2].s, s3 = pc[3].s;
        if (__ballot((s0 >= 0) | (s1 >= 0) | (s2 >= 0) | (s3 >= 0)) == 0ull) return;   // wave-uniform: nothing here lands in an active voxel
        // ---- phase C: run-length accumulation over the lane's 4 consecutive points.  Lidar storage order keeps neighbours in one
        // voxel (on the bench scans 92 % of the lanes see a single run, the rest two), so the sums are formed per RUN -- a maximal
        // group of consecutive points of one slot, summed in point order -- and a lane converts to fixed point and touches LDS once
        // or twice per trip.  No state machine: with 4 points a run is the run of point 0 (A), the run of point 3 when that is
        // another one (Z), or lies strictly between them (points 1 / 2: needs three runs in four points; its own rare branch).
        // Membership is a mask per point, so the two sets of sums are straight-line code. ----
        {
            const bool e1 = s1 == s0, e2 = s2 == s1, e3 = s3 == s2;
            const bool a1 = e1, a2 = a1 & e2, a3 = a2 & e3;                 // point j continues the run of point 0
            const bool z3 = !a3, z2 = z3 & e3, z1 = z2 & e2;                // point j belongs to the run of point 3, a different run
            const bool i0 = pc[0].inb, i1 = pc[1].inb, i2 = pc[2].inb, i3 = pc[3].inb;
            // The nine sums of a run as two scalars and three PAIRS -- (sum x, sum y), (sum xx, sum yy), (sum xz, sum yz) -- so that a
            // point costs 2 v_pk_fma_f32 + 1 v_pk_add_f32 + 2 v_fmac + 1 v_add instead of 6 + 3 scalar instructions; every half of a
            // packed fma is the same IEEE fma, so the bits are those of the scalar form.  (The OTHER packing -- run A with run Z in
            // the two halves -- needs 79 VGPRs and spills, and the Z sums can then no longer be sunk into the branch that flushes them.)
            typedef float f2 __attribute__((ext_vector_type(2)));
            // A lane's SECOND run (the run of point 3 when it is not the run of point 0: 8 % of the lanes, but 93 % of the wave-trips
            // have such a lane, so its sums and its flush used to be paid by nearly every trip) almost always continues in the next lane:
            // lane L's points 1-3 and lane L + 1's point 0 are neighbours in the stream.  When it does (`fw`), lane L hands its suffix
            // points to lane L + 1 (nine DPP moves), which adds them -- first, i.e. in stream order -- to its own run A: one flush per
            // lane-trip instead of two, no Z sums on the common path.  Lane 63 has no successor and a suffix run that ends exactly at
            // the lane boundary has nothing to join: those keep the Z path below, now a rare wave-uniform branch.  What is grouped
            // into one float partial sum stays a property of the data: a maximal run of one slot inside an aligned group of 4 points,
            // extended backwards over the suffix run of the previous group of the same aligned 256-point block (a wave's trip).
            const uint32_t zr = 1u + (z2 ? 1u : 0u) + (z1 ? 1u : 0u);
            const uint32_t zi = (i3 ? 1u : 0u) + ((z2 & i2) ? 1u : 0u) + ((z1 & i1) ? 1u : 0u);
            // (list pass: only where the next lane's group IS the next group of the same aligned 256-point block of the scan -- where the full pass hands on)
            bool fw = z3 & (s3 >= 0) & (wave_shl1(s0, -2) == s3);
            if constexpr (kMode == 2) fw &= (wave_shl1(grp, -7) == grp + 1) & (((grp + 1) & 63) != 0);
            const bool h1 = fw & z1 & i1, h2 = fw & z2 & i2, h3 = fw & i3;
            const float g1x = wave_shr1_zero(h1 ? pc[1].dx : 0.f), g1y = wave_shr1_zero(h1 ? pc[1].dy : 0.f), g1z = wave_shr1_zero(h1 ? pc[1].dz : 0.f);
            const float g2x = wave_shr1_zero(h2 ? pc[2].dx : 0.f), g2y = wave_shr1_zero(h2 ? pc[2].dy : 0.f), g2z = wave_shr1_zero(h2 ? pc[2].dz : 0.f);
            const float g3x = wave_shr1_zero(h3 ? pc[3].dx : 0.f), g3y = wave_shr1_zero(h3 ? pc[3].dy : 0.f), g3z = wave_shr1_zero(h3 ? pc[3].dz : 0.f);
            const uint32_t gc = (uint32_t)wave_shr1_zero((int)(fw ? (zr | (zi << 8)) : 0u));      // raw | in-bounds counts of what arrives
            f2 Axy, Asq, Acz; float A2, A4, A8;
            {   // the points handed over by the previous lane come first (zeros when there are none)
                Axy = f2{g1x, g1y}; A2 = g1z; Asq = Axy * Axy; Acz = f2{g1z, g1z} * Axy; A4 = g1x * g1y; A8 = g1z * g1z;
            }
#define ICET_ACC_ADD(S, vx_, vy_, vz_)                                                                                                       \
            {                                                                                                                                \
                const f2 vxy = f2{vx_, vy_}; const float vz = vz_;                                                                           \
                S##xy += vxy; S##2 += vz;                                                                                                    \
                S##sq = __builtin_elementwise_fma(vxy, vxy, S##sq); S##cz = __builtin_elementwise_fma(vxy, f2{vz, vz}, S##cz);               \
                S##4 = fmaf(vxy.x, vxy.y, S##4); S##8 = fmaf(vz, vz, S##8);                                                                  \
            }
            ICET_ACC_ADD(A, g2x, g2y, g2z)
            ICET_ACC_ADD(A, g3x, g3y, g3z)
            // then the lane's own run A; an out-of-bounds point only counts (its d may be anything: masked to 0, never multiplied in)
            { const bool m = i0;      ICET_ACC_ADD(A, m ? pc[0].dx : 0.f, m ? pc[0].dy : 0.f, m ? pc[0].dz : 0.f) }
            { const bool m = a1 & i1; ICET_ACC_ADD(A, m ? pc[1].dx : 0.f, m ? pc[1].dy : 0.f, m ? pc[1].dz : 0.f) }
            { const bool m = a2 & i2; ICET_ACC_ADD(A, m ? pc[2].dx : 0.f, m ? pc[2].dy : 0.f, m ? pc[2].dz : 0.f) }
            { const bool m = a3 & i3; ICET_ACC_ADD(A, m ? pc[3].dx : 0.f, m ? pc[3].dy : 0.f, m ? pc[3].dz : 0.f) }
            const float A0 = Axy.x, A1 = Axy.y, A3 = Asq.x, A6 = Asq.y, A5 = Acz.x, A7 = Acz.y;
            const uint32_t ar = 1u + (a1 ? 1u : 0u) + (a2 ? 1u : 0u) + (a3 ? 1u : 0u) + (gc & 0xFFu);
            const uint32_t ai = (i0 ? 1u : 0u) + ((a1 & i1) ? 1u : 0u) + ((a2 & i2) ? 1u : 0u) + ((a3 & i3) ? 1u : 0u) + (gc >> 8);
            flush(ghost ? -1 : s0, ar, ai, A0, A1, A2, A3, A4, A5, A6, A7, A8);                 // (the ghost is here for its suffix run alone)
            if (__ballot(z3 & (s3 >= 0) & !fw) != 0ull) {                   // a suffix run that could not be handed on
                f2 Zxy, Zsq, Zcz; float Z2, Z4, Z8;
                { const bool m = z1 & i1; const float x = m ? pc[1].dx : 0.f, y = m ? pc[1].dy : 0.f, z = m ? pc[1].dz : 0.f;
                  Zxy = f2{x, y}; Z2 = z; Zsq = Zxy * Zxy; Zcz = f2{z, z} * Zxy; Z4 = x * y; Z8 = z * z; }
                { const bool m = z2 & i2; ICET_ACC_ADD(Z, m ? pc[2].dx : 0.f, m ? pc[2].dy : 0.f, m ? pc[2].dz : 0.f) }
                { const bool m = i3;      ICET_ACC_ADD(Z, m ? pc[3].dx : 0.f, m ? pc[3].dy : 0.f, m ? pc[3].dz : 0.f) }
                // list pass: lane 63 cannot see its successor -- its suffix run is handed on or flushed by the next trip's ghost
                flush((z3 & !fw & !(kMode == 2 && lane == 63)) ? s3 : -1, zr, zi, Zxy.x, Zxy.y, Z2, Zsq.x, Z4, Zcz.x, Zsq.y, Zcz.y, Z8);
            }
#undef ICET_ACC_ADD
            const bool m1 = !a1 & !z1, m2 = !a2 & !z2;                      // points of a run strictly between A and Z
            if (__ballot((m1 & (s1 >= 0)) | (m2 & (s2 >= 0))) != 0ull) {
                const bool joint = m1 & m2 & e2;                            // points 1 and 2 form one run
                const float bx = i1 ? pc[1].dx : 0.f, by = i1 ? pc[1].dy : 0.f, bz = i1 ? pc[1].dz : 0.f;
                const float cx = i2 ? pc[2].dx : 0.f, cy = i2 ? pc[2].dy : 0.f, cz = i2 ? pc[2].dz : 0.f;
                if (m1 & !joint) flush(ghost ? -1 : s1, 1u, i1 ? 1u : 0u, bx, by, bz, bx * bx, bx * by, bx * bz, by * by, by * bz, bz * bz);
                if (m2) {
                    const float jx = joint ? bx : 0.f, jy = joint ? by : 0.f, jz = joint ? bz : 0.f;
                    flush(ghost ? -1 : s2, joint ? 2u : 1u, (i2 ? 1u : 0u) + ((joint & i1) ? 1u : 0u), jx + cx, jy + cy, jz + cz,
                          fmaf(cx, cx, jx * jx), fmaf(cx, cy, jx * jy), fmaf(cx, cz, jx * jz), fmaf(cy, cy, jy * jy), fmaf(cy, cz, jy * jz), fmaf(cz, cz, jz * jz));
                }
            }
        }
      }
    };
    if (kKeep && list_mode) {
        // Lane l of trip tau holds list entry 63 tau - 1 + l.  Lane 0 is the GHOST: it repeats the last entry of the previous trip (whose lane 63 cannot see its
        // successor) to finish that group's suffix run -- handed to lane 1 where the full pass hands it to the next lane, flushed otherwise -- and contributes
        // nothing else.  The entries of trip tau + 2 kW are requested while trip tau is classified: the point loads of the next trip need theirs in a register.
        int tau = tau_begin + (int)(threadIdx.x >> 6);
        int gc = gc_first, gn_raw = gn_first;
        gc = entry_ok(tau) ? gc : -1;
        load4(gc >= 0 ? 4 * gc : end, XN, YN, ZN);
        for (; tau < tau_end; tau += kW) {                                          // wave-uniform
            const int gnn_raw = entry_raw(tau + 2 * kW);                            // in flight while this trip is classified
            const int gn = entry_ok(tau + kW) ? gn_raw : -1;                        // (requested a trip ago)
            trip(std::integral_constant<int, 2>{}, gc >= 0 ? 4 * gc : end, tau + kW < tau_end, gn >= 0 ? 4 * gn : end, gc);
            gc = gn; gn_raw = gnn_raw;
        }
    } else {
        load4(begin + kAccPts * (int)threadIdx.x, XN, YN, ZN);
        for (int t0 = begin + kAccPts * threadIdx.x; t0 < begin + cs; t0 += kAccPts * kAccBlock)    // whole waves iterate together
            trip(std::integral_constant<int, kKeep ? 1 : 0>{}, t0, t0 + kAccPts * kAccBlock < begin + cs, t0 + kAccPts * kAccBlock, -1);
    }
    if (ICET_ACC_PHASE != 9 && sink == 1.2345e-30f) near_over_count[pair] = 1u;      // keeps the timing builds' work alive
    __syncthreads();
    {   // ---- the parked points: literal classification, each a run of one ----
        const uint32_t nq = min(nearq[near_cap], near_cap);
        for (uint32_t e = threadIdx.x; e < nq; e += kAccBlock) {
            const int i = (int)nearq[e];
            float qx, qy, qz;
            transform_point(px[i], py[i], pz[i], xf, qx, qy, qz);
            PointClass pc1;
            classify_literal(qx, qy, qz, map, thr, T, P, hs, pc1, kRT2);
            if (pc1.s >= 0) {
                const float dx = pc1.dx, dy = pc1.dy, dz = pc1.dz;
                unsigned long long* F = lacc + min(pc1.s, nl > 0 ? nl - 1 : 0) * 10;
                if (pc1.s < nl) {
                    atomicAdd(&F[0], 1ull | ((unsigned long long)(pc1.inb ? 1u : 0u) << kCntBits));   // unbiased values below: not counted as conversions
                    if (pc1.inb) {
                        atomicAdd(&F[1], to_fix(dx)); atomicAdd(&F[2], to_fix(dy)); atomicAdd(&F[3], to_fix(dz));
                        atomicAdd(&F[4], to_fix(dx * dx)); atomicAdd(&F[5], to_fix(dx * dy)); atomicAdd(&F[6], to_fix(dx * dz));
                        atomicAdd(&F[7], to_fix(dy * dy)); atomicAdd(&F[8], to_fix(dy * dz)); atomicAdd(&F[9], to_fix(dz * dz));
                    }
                } else {
                    acc_add_hbm(gacc + (size_t)pc1.s * kAccWords, 1u, pc1.inb ? 1u : 0u, dx, dy, dz, dx * dx, dx * dy, dx * dz, dy * dy, dy * dz, dz * dz);
                }
            }
        }
        if (threadIdx.x < 4 && nearq[near_cap + 1 + threadIdx.x] != 0u) {       // the points on the origin, by sign pattern: one literal classification for all of them
            PointClass pz;
            classify_literal((threadIdx.x & 1) ? -0.f : 0.f, (threadIdx.x & 2) ? -0.f : 0.f, 0.f, map, thr, T, P, hs, pz, false);
            const uint32_t cnt = nearq[near_cap + 1 + threadIdx.x];
            if (pz.s >= 0) {                                               // (never in bounds: phi = 1000)
                if (pz.s < nl) atomicAdd(&lacc[pz.s * 10], (unsigned long long)cnt);
                else atomicAdd(reinterpret_cast<unsigned long long*>(gacc + (size_t)pz.s * kAccWords), (unsigned long long)cnt);
            }
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nl; s += kAccBlock) {
        const unsigned long long* L = lacc + s * 10;
        const unsigned long long cnt = L[0];
        const unsigned long long raw = cnt & kCntMask, in = (cnt >> kCntBits) & kCntMask, conv = cnt >> (2 * kCntBits);
        if (raw == 0ull) continue;                                          // no point of this chunk reached the voxel
        unsigned long long* G = reinterpret_cast<unsigned long long*>(gacc + (size_t)s * kAccWords);
        atomicAdd(&G[0], raw | (in << 32));
        if (in) {
            const unsigned long long bias = conv * kFixBias;                // mod 2^64, like the sums
#pragma unroll
            for (int k = 1; k < 10; k++) atomicAdd(&G[k], L[k] - bias);
        }
    }
}

// The point pass as a kernel of its own (throughput batches, fine grids, timing): the solve follows in k_gn_solve.
template <bool kVec4, bool kRT2, bool kSmall>
__global__ __launch_bounds__(kAccBlock, kRT2 ? 1 : (kSmall ? 2 : kAccWavesPerSimd)) void k_gn_accumulate(const PairDesc* __restrict__ desc, const float* __restrict__ xf_all,
                                                          const int16_t* __restrict__ slot_of_voxel, const int32_t* __restrict__ n_slots,
                                                          const SlotHot* __restrict__ hotS, uint32_t* __restrict__ acc,
                                                          const float* __restrict__ thr, const LutCell* __restrict__ lut,
                                                          int T, int P, int Mt, int Mp, float guard_t, float guard_p,
                                                          int lds_slots, int chunks, int n_pairs, int force_exact,
                                                          uint32_t* __restrict__ near_over, uint32_t* __restrict__ near_over_count) {
    gn_accumulate_body<kVec4, kRT2, kSmall, false>(desc, xf_all, slot_of_voxel, n_slots, hotS, acc, thr, lut, T, P, Mt, Mp, guard_t, guard_p, lds_slots, chunks, n_pairs, force_exact, near_over, near_over_count, KeepDev{});
}

// The point pass with the keep list (throughput batches; KeepState in icet_internal.h): every block reads its pair's mode -- the list, or the whole scan plus
// the keep masks of its points.  Same sums, same bits as k_gn_accumulate.
template <bool kVec4>
__global__ __launch_bounds__(kAccBlock, kAccWavesPerSimd) void k_gn_accumulate_keep(const PairDesc* __restrict__ desc, const float* __restrict__ xf_all,
                                                          const int16_t* __restrict__ slot_of_voxel, const int32_t* __restrict__ n_slots,
                                                          const SlotHot* __restrict__ hotS, uint32_t* __restrict__ acc,
                                                          const float* __restrict__ thr, const LutCell* __restrict__ lut,
                                                          int T, int P, int Mt, int Mp, float guard_t, float guard_p,
                                                          int lds_slots, int chunks, int n_pairs, int force_exact,
                                                          uint32_t* __restrict__ near_over, uint32_t* __restrict__ near_over_count, KeepDev kd) {
    gn_accumulate_body<kVec4, false, false, true>(desc, xf_all, slot_of_voxel, n_slots, hotS, acc, thr, lut, T, P, Mt, Mp, guard_t, guard_p, lds_slots, chunks, n_pairs, force_exact, near_over, near_over_count, kd);
}

// Small batches (a sequential caller's single pair above all): the point pass and the solve of one iteration in ONE launch.  Every block of a pair takes a ticket when
// its sums are in HBM; the block that draws the last one -- every other block's atomics and overflow entries are then visible to it: release fence before the ticket,
// acquire fence after -- runs the pair's solve (gn_solve_body, the one-block form of grids up to 4096 voxels, on its first 256 threads) and clears the ticket counter
// for the next iteration.  No block waits for another: a pair's solve simply rides on whichever block happens to be last.  Same sums, same algebra, same bits as
// k_gn_accumulate + k_gn_solve; seven launches less per solve (each costs a sequential caller ~6 us on the device and ~7 us of host time in a graph replay).
// (The parameters carry no __restrict__ here: acc / the overflow list are written by the first half and read by the second.)
struct SolveFuse { const SlotFit* fitS; float* X; float* out; AuxDev aux; int n, iter, runlen, reject_moving; float cond_bound2; uint32_t* done; };
template <bool kVec4>
__global__ __launch_bounds__(kAccBlock, 2) void k_gn_accumulate_solve(const PairDesc* desc, float* xf_all, const int16_t* slot_of_voxel, const int32_t* n_slots,
                                                                      const SlotHot* hotS, uint32_t* acc, const float* thr, const LutCell* lut,
                                                                      int T, int P, int Mt, int Mp, float guard_t, float guard_p,
                                                                      int lds_slots, int chunks, int n_pairs, int force_exact,
                                                                      uint32_t* near_over, uint32_t* near_over_count, SolveFuse sf) {
    gn_accumulate_body<kVec4, false, true, false>(desc, xf_all, slot_of_voxel, n_slots, hotS, acc, thr, lut, T, P, Mt, Mp, guard_t, guard_p, lds_slots, chunks, n_pairs, force_exact, near_over, near_over_count, KeepDev{});
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;           // (a padding block of the grid: belongs to no pair)
    __shared__ uint32_t s_last;
    // release: this block's atomics and overflow entries before its ticket.  The device-wide half of the fence costs an L2 write-back on this part (eight L2s, one
    // per XCD), so ONE thread pays it: the block's waves order their writes before the barrier at work-group scope, thread 0's device-scope release behind the
    // barrier then covers them (cumulativity); the acquire side likewise.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t t = atomicAdd(&sf.done[pair], 1u);
        s_last = (t == (uint32_t)chunks - 1u) ? 1u : 0u;
        if (s_last) { sf.done[pair] = 0u; __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }     // every block of the pair has drawn: ready for the next iteration's launch
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const NearOverflow over{desc, slot_of_voxel, hotS, thr, near_over, near_over_count, T, P, 0};
    gn_solve_body<256, 0, kAccBlock, true>(n_slots, sf.fitS, acc, sf.X, xf_all, sf.out, sf.aux, T * P, sf.n, sf.iter, sf.runlen, over, sf.reject_moving, nullptr, 1, sf.cond_bound2, KeepArgs{}, pair);      // (the default W; ICET_FLAG_DOUBLE_W is not fused)
}

inline int chunks_for(int n_pairs, int max_n, int per_block_min, int target_blocks) {
    int by_work = (max_n + per_block_min - 1) / per_block_min;
    int want = (target_blocks + n_pairs - 1) / n_pairs;
    int c = want < by_work ? want : by_work;
    return c < 1 ? 1 : c;
}

}  // namespace

#define ICET_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t init_accumulate_kernels() {
    hipError_t e = hipSuccess;
#define ICET_ACC_ATTR(V4, RT, SM) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gn_accumulate<V4, RT, SM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    ICET_ACC_ATTR(true, false, false); ICET_ACC_ATTR(false, false, false); ICET_ACC_ATTR(true, true, false); ICET_ACC_ATTR(false, true, false);
    ICET_ACC_ATTR(true, false, true); ICET_ACC_ATTR(false, false, true); ICET_ACC_ATTR(true, true, true); ICET_ACC_ATTR(false, true, true);
#undef ICET_ACC_ATTR
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gn_accumulate_keep<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gn_accumulate_keep<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gn_accumulate_solve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gn_accumulate_solve<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    return e;
}

// points2_OG of prepScan2 (src/icet.cpp:263-275) without its permutation (which only reorders sums): every row of scan 2 through
// cartesianToSpherical -> sphericalToCartesian under the shared arithmetic rule, written to the round-tripped copy the loop then reads.
__global__ __launch_bounds__(kBlock) void k_rt2_prepare(const PairDesc* __restrict__ desc, const PairDesc* __restrict__ desc_rt, int n_pairs, int chunks) {
    int pair, chunk;
    if (!decode_block(n_pairs, chunks, pair, chunk)) return;
    const PairDesc d = desc[pair], o = desc_rt[pair];
    float* ox = const_cast<float*>(o.s2); float* oy = ox + o.ld2; float* oz = ox + 2 * (size_t)o.ld2;
    const float* px = d.s2; const float* py = px + d.ld2; const float* pz = px + 2 * (size_t)d.ld2;
    int cs = (d.n2 + chunks - 1) / chunks; cs = (cs + kBlock - 1) / kBlock * kBlock;
    const int lo = chunk * cs, hi = min(d.n2, lo + cs);
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) {
        float x, y, z;
        roundtrip_any(px[i], py[i], pz[i], x, y, z);
        ox[i] = x; oy[i] = y; oz[i] = z;
    }
}

hipError_t launch_rt2_prepare(const Workspace& w, const LaunchCfg& c, hipStream_t st) {
    const int chunks = std::max(1, std::min(64, (c.max_n2 + 4 * kBlock - 1) / (4 * kBlock)));
    k_rt2_prepare<<<grid_groups(c.n_pairs) * chunks, kBlock, 0, st>>>(w.desc, w.desc_rt, c.n_pairs, chunks);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

// LDS of one k_gn_accumulate block apart from its slot rows: both look-up tables, the voxel -> slot map, the queue of parked points; and the bytes of a slot row.
// The SMALLEST launch keeps 32 rows (launch_gn_accumulate): ensure_thresholds sizes the look-up tables so that it fits the device.
size_t acc_fixed_lds_bytes(int T, int P, int Mt, int Mp, bool small_batch, bool keep) {
    const uint32_t near_cap = small_batch ? kNearCapSmall : kNearCap;
    const size_t keep_bytes = keep ? (size_t)(T + P + 4) * 4 + (((size_t)T * P + 3 * (size_t)T + 1 + 3) & ~(size_t)3) : 0;      // bin edges + the voxels' neighbourhood codes (k_edge_t, k_nb)
    return (size_t)(Mt + Mp + 2) * sizeof(LutCell) + (size_t)(((size_t)T * P + T + 4) / 2) * 4 + (near_cap + 5) * 4 + 32 + keep_bytes;   // + alignment of the hot records
}
size_t acc_row_lds_bytes() { return (kHotWords + kAccLds) * 4; }

hipError_t launch_gn_accumulate(const Workspace& w, const LaunchCfg& c, hipStream_t st, const FuseArgs* fuse, bool* fused, int keep_pass) {
    if (fused) *fused = false;
    if (keep_pass && (c.n_pairs < 32 || c.rt2 || !w.keep_state || !w.keep_mask || !w.keep_list || !w.edges || !w.keep_modes)) return hipErrorInvalidValue;      // (enqueue_loop asks for it on throughput batches only)
    // LDS rows for active voxels: a throughput batch keeps 320 rows (measured optimum on 64-channel scans: fewer rows
    // spill busy voxels to HBM atomics, more rows cost occupancy); a small batch has
    // CUs to spare, so a block may take most of a CU's LDS and keep every active voxel of a fine grid (150 x 48: often
    // > 1000) out of the slow HBM-atomic path.
    const uint32_t near_cap = (c.n_pairs >= 32) ? kNearCap : kNearCapSmall;
    const size_t fixed = acc_fixed_lds_bytes(c.T, c.P, w.lut_Mt, w.lut_Mp, near_cap == kNearCapSmall, keep_pass != 0);
    const size_t row = acc_row_lds_bytes();
    // the fused form (k_gn_accumulate_solve): small batch, one-block solve (grids up to 4096 voxels: launch_gn_solve), no scan-2 round trip
    // (... and a grid whose tables leave the fused kernel -- 12 KB less dynamic LDS: its solve half has static tables -- fewer than the 32 slot rows every launch keeps)
    const bool fuse_it = fuse && c.fuse_solve && near_cap == kNearCapSmall && c.V <= 4096 && !c.rt2 && c.ref_w && w.gn_done() && fixed + 32 * row <= (size_t)148 * 1024;
    const size_t budget = (c.n_pairs >= 32) ? fixed + 320 * row : (fuse_it ? 144 : 156) * 1024;   // 320 rows: ~46 KB/block for 75 x 24, three blocks per CU; a small batch: one block per CU, nearly all of its LDS (the fused kernel's solve half has ~6 KB of static tables)
    int lds_slots = c.lds_slots > 0 ? c.lds_slots : (int)((budget > fixed ? budget - fixed : 0) / row);
    lds_slots = lds_slots < 32 ? 32 : lds_slots;
    const int fit = (int)(((fuse_it ? 148 : 160) * 1024 - fixed) / row);               // what a block can hold at all (option "lds_slots" is a wish, not a launch failure); >= 32: ensure_thresholds
    if (lds_slots > fit) lds_slots = fit;
    if (lds_slots > c.V) lds_slots = c.V;
    int chunks = chunks_for(c.n_pairs, c.max_n2, kAccBlock * c.acc_min_pts_per_thread, c.acc_target_blocks);
    chunks = std::max(chunks, (int)(((long long)c.max_n2 + (1 << 20) - 1) >> 20));          // a block's counts live in 21-bit fields: at most 2^20 (+ rounding) points per block
    const size_t lds = fixed + (size_t)lds_slots * row;
    dim3 grid(grid_groups(c.n_pairs) * chunks), blk(kAccBlock);
    const LutCell* lut = reinterpret_cast<const LutCell*>(w.lut);
#define ICET_ACC_LAUNCH(V4, RT, SM) k_gn_accumulate<V4, RT, SM><<<grid, blk, lds, st>>>(w.desc, w.xf, w.slot_of_voxel, w.n_slots, w.hotS, w.acc, w.thr, lut, c.T, c.P, w.lut_Mt, w.lut_Mp, \
                                                     w.guard_t, w.guard_p, lds_slots, chunks, c.n_pairs, c.force_exact, w.near_over, w.near_over_count)
    if (fuse_it) {
        AuxDev aux{}; if (fuse->aux) aux = *fuse->aux;
        aux.pair_user = c.pair_user; aux.done_flag = (fuse->iter == c.runlen - 1) ? c.done_flag : nullptr;      // (as launch_gn_solve sets them)
        const SolveFuse sf{w.fitS, w.X, fuse->d_out, aux, c.n, fuse->iter, c.runlen, c.reject_moving, c.gn_cond_bound2, w.gn_done()};
        if (c.vec4_ok) k_gn_accumulate_solve<true><<<grid, blk, lds, st>>>(w.desc, w.xf, w.slot_of_voxel, w.n_slots, w.hotS, w.acc, w.thr, lut, c.T, c.P, w.lut_Mt, w.lut_Mp, w.guard_t, w.guard_p, lds_slots, chunks, c.n_pairs, c.force_exact, w.near_over, w.near_over_count, sf);
        else k_gn_accumulate_solve<false><<<grid, blk, lds, st>>>(w.desc, w.xf, w.slot_of_voxel, w.n_slots, w.hotS, w.acc, w.thr, lut, c.T, c.P, w.lut_Mt, w.lut_Mp, w.guard_t, w.guard_p, lds_slots, chunks, c.n_pairs, c.force_exact, w.near_over, w.near_over_count, sf);
        ICET_LAUNCH_CHECK();
        if (fused) *fused = true;
        return hipSuccess;
    }
    if (keep_pass) {
        const double two_pi = 6.283185307179586476925286766559;
        const KeepDev kd{w.keep_mask, w.keep_list, w.edges, w.keep_modes + (size_t)(keep_pass >> 1) * c.n_pairs, 1.05f * c.keep_br + 1e-4f, 1.05f * c.keep_bt,
                         (float)std::min(two_pi / c.T, 0.5), (float)std::min(0.5 * two_pi / c.P, 0.5), 1.0f / (float)c.T};
        if (c.vec4_ok) k_gn_accumulate_keep<true><<<grid, blk, lds, st>>>(w.desc, w.xf, w.slot_of_voxel, w.n_slots, w.hotS, w.acc, w.thr, lut, c.T, c.P, w.lut_Mt, w.lut_Mp, w.guard_t, w.guard_p, lds_slots, chunks, c.n_pairs, c.force_exact, w.near_over, w.near_over_count, kd);
        else k_gn_accumulate_keep<false><<<grid, blk, lds, st>>>(w.desc, w.xf, w.slot_of_voxel, w.n_slots, w.hotS, w.acc, w.thr, lut, c.T, c.P, w.lut_Mt, w.lut_Mp, w.guard_t, w.guard_p, lds_slots, chunks, c.n_pairs, c.force_exact, w.near_over, w.near_over_count, kd);
        ICET_LAUNCH_CHECK();
        return hipSuccess;
    }
#define ICET_ACC_LAUNCH2(V4, RT) do { if (near_cap == kNearCapSmall) ICET_ACC_LAUNCH(V4, RT, true); else ICET_ACC_LAUNCH(V4, RT, false); } while (0)
    if (c.rt2) { if (c.vec4_ok) ICET_ACC_LAUNCH2(true, true); else ICET_ACC_LAUNCH2(false, true); }
    else { if (c.vec4_ok) ICET_ACC_LAUNCH2(true, false); else ICET_ACC_LAUNCH2(false, false); }
#undef ICET_ACC_LAUNCH2
#undef ICET_ACC_LAUNCH
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace icet
