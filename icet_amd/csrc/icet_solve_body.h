// icet_amd/csrc/icet_solve_body.h -- the per-voxel and 6 x 6 part of one Gauss-Newton iteration as a DEVICE FUNCTION (gn_solve_body), shared by
// k_gn_solve (icet_solve.hip: one launch per iteration behind the point pass) and by k_gn_accumulate_solve (icet_accumulate.hip: small batches --
// the block of a pair that finishes its share of the point pass LAST runs the pair's solve in the same launch, which halves the kernel count of
// the loop of a sequential caller).  ICET::fitScan2, /root/reference/src/icet.cpp:372-436 after the point pass.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "icet_internal.h"
#include "icet_device_common.h"
#include "icet_device_math.h"

namespace icet {
namespace {

// Per-pair transform record, kXf floats: t[3] | R[9] (utils::R, src/utils.cpp:144-152, row-major) | angles[3] | pad |
// J[27] (get_H's three derivative matrices, src/icet.cpp:507-529).  Written once per iteration by the lane that updates
// X, so the six sin/cos are evaluated once and serve both the next point pass (R) and the next voxel pass (J).
constexpr int kXf = 48;
constexpr int kTwoStageBlocks = 3;        // first-stage blocks per pair of the two-stage solve (3 x 512 slots: one round for up to 1536 active voxels)
constexpr int kTwoStageMaxPairs = 4;      // the two-stage form is for small batches on fine grids; a throughput batch has a block per CU anyway
constexpr int kMaxVirtualBlocks = (kMaxVoxels + 511) / 512;     // 20
static_assert(kTwoStageMaxPairs * kMaxVirtualBlocks * 27 <= kGnPartWords, "Workspace::gn_part");
__device__ __forceinline__ void write_xf(float* xf, const float X[6]) {
    const float phi = X[3], theta = X[4], psi = X[5];
    // the shared arithmetic rule: the correctly rounded float of the exact value -- evaluated in double, rounded once -- like
    // every other transcendental of the path.  With ocml's float sincosf (1 ulp) the matrix differed from the CPU restatement's in a last bit now and then, and the
    // FIRST iteration of a solve with X0 != 0 then put a point or two per 100 k into the neighbouring voxel (tests/param_sweep.py found it).  One lane per pair
    // and iteration: six double evaluations.
    double sd[3], cd[3];
    sincos((double)phi, &sd[0], &cd[0]); sincos((double)theta, &sd[1], &cd[1]); sincos((double)psi, &sd[2], &cd[2]);      // one argument reduction per angle: 0.4 us per solve against six separate calls, same bits
    const float sph = (float)sd[0], cph = (float)cd[0], sth = (float)sd[1], cth = (float)cd[1], sps = (float)sd[2], cps = (float)cd[2];
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

// The 6 x 6 tail of one iteration (src/icet.cpp:410-433) for one (HTWH, HTWdz).
// Normal case first: HTWH positive definite with a condition number that the Frobenius bound cond_2 <= |A|_F |A^-1|_F proves to be <= sqrt(bound2)
// (default 2.5e5: a factor 4 below checkCondition's cutoff of 1e6, src/icet.cpp:453,469 -- at cond ~ 1e6 a float inverse knows its own norm to a few per
// cent only, and with the bound AT the cutoff the device kept matrices on this route that the reference prunes; the 256 bench pairs reach 2.1e5).  Then nothing is pruned, every pivot is above the pseudo-inverse's rank threshold
// (6 eps < 1e-6), pinv(HTWH) is the inverse and dx = HTWH^-1 HTWdz: a Cholesky factorisation gives both (route 0).  Everything else -- bound
// inconclusive, Cholesky pivot not positive, NaN -- takes the literal restatement of the reference's statements (route 2, icet_device_math.h
// gn_tail_literal), which decides rank, pruning and eigenvector signs exactly as the reference's algorithms do on the same bits.
// `ws`: the literal route's workspace in LDS; `leader`: the one lane of the wave that fills in the matrix (the whole wave then walks the workspace and
// reads the results, so that the outputs are wave-uniform on either route).  Must be called by a whole wave, every lane with the same matrix.
// (is the Cholesky route enough?  `cov` = the inverse when it is)
__device__ __forceinline__ bool gn_tail_plain(const float* Hm, float bound2, float* cov) {
    bool plain = icetdev::chol6_inverse(Hm, cov);
    if (plain) {
        float fa = 0.f, fi = 0.f;
        for (int k = 0; k < 36; k++) { fa += Hm[k] * Hm[k]; fi += cov[k] * cov[k]; }
        plain = fa * fi <= bound2;                                  // (false for NaN)
    }
    return plain;
}
// pinv_done != nullptr: a second wave of the block runs gn_tail_helper on the same matrix beside this call (gn_solve_body)
__device__ __forceinline__ void gn_tail(const float* Hm, const float* g, float bound2, float* cov, float* ps, float* dx, float* ev, int& route, int& pruned,
                                        icetdev::GnTailWs& ws, bool leader, volatile int* pinv_done = nullptr) {
    bool plain = gn_tail_plain(Hm, bound2, cov);
    if (plain) {
        for (int k = 0; k < 6; k++) ps[k] = sqrtf(fabsf(cov[k * 6 + k]));         // src/icet.cpp:412-417
        for (int a = 0; a < 6; a++) { float t = 0.f; for (int b = 0; b < 6; b++) t += cov[a * 6 + b] * g[b]; dx[a] = t; }
        for (int k = 0; k < 6; k++) ev[k] = __builtin_nanf("");     // not computed on this route
        route = 0; pruned = 0;
    } else {
        if (leader) {
            for (int k = 0; k < 36; k++) ws.H[k] = Hm[k];
            for (int k = 0; k < 6; k++) ws.g[k] = g[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (pinv_done) icetdev::gn_tail_literal<true>(ws, pinv_done); else icetdev::gn_tail_literal<false>(ws);      // the whole wave works (`plain` is wave-uniform: every lane holds the same matrix)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        for (int k = 0; k < 36; k++) cov[k] = ws.cov[k];
        for (int k = 0; k < 6; k++) { ps[k] = ws.ps[k]; dx[k] = ws.dx[k]; ev[k] = ws.ev[k]; }
        pruned = ws.pruned;
        route = 2;
    }
}

// The second wave's share of the 6 x 6 tail: the same decision on the same matrix; on the literal route pinv(HTWH) while the first wave runs the eigen-decomposition.
__device__ __forceinline__ void gn_tail_helper(const float* Hm, float bound2, icetdev::GnTailWs& ws, volatile int* pinv_done) {
    float cov[36];
    if (gn_tail_plain(Hm, bound2, cov)) return;                     // the first wave takes the Cholesky route and waits for nobody
    icetdev::gn_tail_literal_helper(Hm, ws, pinv_done);
}

// Undecided scan-2 points that did not fit a block's LDS queue in k_gn_accumulate (see there): literal classification, each a
// run of one, straight into the HBM accumulators.  Empty on ordinary data; the whole list when the force_exact diagnostic is on.
struct NearOverflow { const PairDesc* desc; const int16_t* slot_of_voxel; const SlotHot* hotS; const float* thr; uint32_t* list; uint32_t* count; int T, P; int rt2; };

__device__ __noinline__ void drain_near_overflow(const NearOverflow& o, int pair, int V, const float* __restrict__ xf, uint32_t* __restrict__ acc_pair, uint32_t nov) {
    const PairDesc d = o.desc[pair];
    const float* px = d.s2; const float* py = px + d.ld2; const float* pz = px + 2 * (size_t)d.ld2;
    const int16_t* map = o.slot_of_voxel + (size_t)pair * ((V + 1) & ~1);
    const SlotHot* hs = o.hotS + (size_t)pair * V;
    // The exact-zero rows of a real scan (5 k - 24 k per scan) all land on ONE point once t != 0 -- the transformed origin -- and when an update is mostly vertical that
    // point lies next to the pole axis, where the polar look-up cells are ambiguous: every one of them is parked, thousands overflow the blocks' queues and arrive here
    // (27 us of literal classifications by this one block: k_gn_solve 40 us instead of 13 in iterations 1-2 of a 256-pair batch of real scans).  Points whose transformed
    // coordinates are bit for bit the transformed origin's are COUNTED; the point is classified once and its contribution added count times -- integer accumulation:
    // exactly what that many runs of one add.
    float o0, o1, o2;
    transform_point(0.f, 0.f, 0.f, xf, o0, o1, o2);
    __shared__ uint32_t s_org;
    if (threadIdx.x == 0) s_org = 0u;
    __syncthreads();
    constexpr int kU = 4;                                                 // entries per thread per round, their loads in flight together: a round is two dependent memory round trips (index -> row)
    for (uint32_t e0 = 0; e0 < nov; e0 += kU * blockDim.x) {              // block-uniform trip count: the ballots below need whole waves
        int idx[kU]; bool have[kU]; float x[kU], y[kU], z[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) { const uint32_t e = e0 + u * blockDim.x + threadIdx.x; have[u] = e < nov; idx[u] = have[u] ? (int)o.list[(size_t)d.off2 + e] : 0; }
#pragma unroll
        for (int u = 0; u < kU; u++) { x[u] = have[u] ? px[idx[u]] : 0.f; y[u] = have[u] ? py[idx[u]] : 0.f; z[u] = have[u] ? pz[idx[u]] : 0.f; }
#pragma unroll
        for (int u = 0; u < kU; u++) {
            float qx, qy, qz;
            transform_point(x[u], y[u], z[u], xf, qx, qy, qz);
            const bool org = have[u] & (__float_as_uint(qx) == __float_as_uint(o0)) & (__float_as_uint(qy) == __float_as_uint(o1)) & (__float_as_uint(qz) == __float_as_uint(o2));
            const unsigned long long m = __ballot(org);
            if (m != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&s_org, (uint32_t)__popcll(m));
            if (have[u] && !org) {
                PointClass pc;
                classify_literal(qx, qy, qz, map, o.thr, o.T, o.P, hs, pc, o.rt2 != 0);
                if (pc.s >= 0)
                    acc_add_hbm(acc_pair + (size_t)pc.s * kAccWords, 1u, pc.inb ? 1u : 0u, pc.dx, pc.dy, pc.dz, pc.dx * pc.dx, pc.dx * pc.dy, pc.dx * pc.dz,
                                pc.dy * pc.dy, pc.dy * pc.dz, pc.dz * pc.dz);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_org != 0u) {
        const unsigned long long cnt = s_org;
        PointClass pc;
        classify_literal(o0, o1, o2, map, o.thr, o.T, o.P, hs, pc, o.rt2 != 0);
        if (pc.s >= 0) {
            unsigned long long* G = reinterpret_cast<unsigned long long*>(acc_pair + (size_t)pc.s * kAccWords);
            atomicAdd(&G[0], cnt | ((pc.inb ? cnt : 0ull) << 32));
            if (pc.inb) {
                atomicAdd(&G[1], cnt * to_fix(pc.dx)); atomicAdd(&G[2], cnt * to_fix(pc.dy)); atomicAdd(&G[3], cnt * to_fix(pc.dz));
                atomicAdd(&G[4], cnt * to_fix(pc.dx * pc.dx)); atomicAdd(&G[5], cnt * to_fix(pc.dx * pc.dy)); atomicAdd(&G[6], cnt * to_fix(pc.dx * pc.dz));
                atomicAdd(&G[7], cnt * to_fix(pc.dy * pc.dy)); atomicAdd(&G[8], cnt * to_fix(pc.dy * pc.dz)); atomicAdd(&G[9], cnt * to_fix(pc.dz * pc.dz));
            }
        }
    }
}

// The keep list of the point pass (KeepState, icet_internal.h), maintained by the block that runs a pair's 6 x 6 part.  keep_list_build: a pair whose point pass walked
// the whole scan has left one 64-bit keep mask per 256 points; its set bits, in order, are the list (entry = group index in the scan).  In rounds of 256 mask
// words: one thread per word, the block's exclusive prefix of the popcounts places a word's entries, the thread writes them bit by bit into an LDS stage (16-bit,
// relative to the round), and the block copies the stage out in coalesced stores.  (Two earlier forms, measured on the 256-pair batch: one WAVE per word with lane l
// writing bit l at its rank -- ~120 words one after the other per wave, +12 us per solve; one thread per word storing straight to memory -- 4.6 M four-byte
// store transactions per launch, +15..25 us.)  The transform the masks were written under (the record BEFORE this solve's update) becomes the list's reference.
// Returns with s_ref = that reference (or the older one of a pair that walked its list) for keep_budget_check, behind a barrier.
constexpr int kKeepRound = 256;
template <int kThreads>
__device__ __forceinline__ int keep_list_build(const KeepArgs& k, int pair, const float* xf, float* s_ref, int* s_wtot) {
    static_assert(kThreads >= kKeepRound && kThreads % 64 == 0, "one thread per mask word of a round");
    __shared__ uint16_t s_ent[kKeepRound * 64];
    KeepState* st = k.state + pair;
    const int mode = st->mode;                                          // block-uniform
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int kW = kKeepRound / 64;
    if (mode != 0 || k.on == 2) {                                       // (on == 2: behind the last pass -- statistics only)
        if (threadIdx.x < 12) s_ref[threadIdx.x] = st->ref[threadIdx.x];
        if (threadIdx.x == 0 && mode != 0) st->list_passes += 1;
        const int nk = st->n_keep;
        __syncthreads();
        return nk;
    }
    const PairDesc d = k.desc[pair];
    const int nw = (d.n2 + 255) >> 8;
    const unsigned long long* M = k.mask + ((size_t)d.off2 >> 8) + pair;
    uint32_t* L = k.list + ((size_t)d.off2 >> 2) + pair;
    if (threadIdx.x < 12) { const float v = xf[threadIdx.x]; s_ref[threadIdx.x] = v; st->ref[threadIdx.x] = v; }
    int carry = 0;                                                      // entries of the words of earlier rounds
    for (int w0 = 0; w0 < nw; w0 += kKeepRound) {                       // block-uniform
        const int w = w0 + (int)threadIdx.x;
        unsigned long long m = ((int)threadIdx.x < kKeepRound && w < nw) ? M[w] : 0ull;
        const int c = __popcll(m);
        const int incl = wave_incl_sum(c);
        if (lane == 63 && wv < kW) s_wtot[wv] = incl;
        __syncthreads();
        int base = incl - c, total = 0;
#pragma unroll
        for (int q = 0; q < kW; q++) { const int t = s_wtot[q]; base += q < wv ? t : 0; total += t; }
        const uint32_t g0 = threadIdx.x * 64u;
        while (m != 0ull) {
            s_ent[base++] = (uint16_t)(g0 + (uint32_t)__builtin_ctzll(m));
            m &= m - 1ull;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < total; j += kThreads) L[carry + j] = (uint32_t)w0 * 64u + s_ent[j];
        carry += total;
        __syncthreads();                                                // (the stage and s_wtot are written again in the next round)
    }
    if (threadIdx.x == 0) { st->n_keep = carry; st->builds += 1; }
    return carry;
}
// The pair's next transform record against the list's reference: inside the budgets the next point pass may walk the list (no dropped point can have reached an
// active bin: keep_margin, icet_accumulate.hip), outside it walks the whole scan and the list is rebuilt behind it.  Called by the first wave.
__device__ __forceinline__ void keep_budget_check(const KeepArgs& k, int pair, const float* xf_new, const float* s_ref, int lane, int n_keep) {
    const float dv = (lane < 12) ? xf_new[lane] - s_ref[lane] : 0.f;
    const float dt2 = wave_total(lane < 3 ? dv * dv : 0.f), dr2 = wave_total(lane >= 3 ? dv * dv : 0.f);
    if (lane == 0) {
        const int mode = (dt2 <= k.bt2 && dr2 <= k.br2) ? 1 : 0;                        // (NaN: the whole scan)
        k.state[pair].mode = mode;
        k.modes_next[pair] = mode ? n_keep + 1 : 0;
    }
}

// kT threads per block: 256 for ordinary grids (a 64-channel scan on 75 x 24 has ~220 active voxels: one round), 512 -- the most that 248 VGPRs allow --
// for fine grids (150 x 48: > 1000 active voxels, three rounds of the per-voxel algebra instead of five)
// kStage 0: everything in ONE block per pair (coarse grids, batches).  The two-stage form of fine grids and small batches (a 150 x 48 grid has
// > 1000 active voxels: three rounds of the per-voxel algebra in one block): kStage 1 = `nblk` blocks per pair, each reduces the contributions of
// its share of the slots to 27 partial sums in HBM; kStage 2 = one block per pair adds the partials and runs the 6 x 6 part.  Undecided points
// waiting in the overflow list must be drained before any sum is read, which one block cannot do for the others: stage 1 then declines (every
// block sees the same count) and stage 2 runs the whole solve like stage 0.
// So that a pair's bits do not depend on which form solved it, fine grids (kT = 512) reduce in ONE canonical tree in every form: the slots in
// "virtual blocks" of 512 (slot s belongs to virtual block s / 512, lane s % 512), each virtual block to its 27 sums (DPP totals of its eight
// waves, added in wave order), the virtual blocks added in index order.  Coarse grids (kT = 256, one block per pair always) keep one reduction
// over whatever a thread accumulated.
#ifndef ICET_SOLVE_PHASE
#define ICET_SOLVE_PHASE 9      /* TIMING BUILDS ONLY (results are wrong below 9): 1 loads, 2 + per-slot algebra, 3 + reductions, 4 + the 6 x 6 tail */
#endif
// kBlockT: threads of the calling block (> kT when the caller is the point-pass kernel: its threads beyond kT own no slot and no row of the reduction table, but
// reach every barrier).  The pointers carry no __restrict__: in the fused kernel acc / the overflow list were written through other names a few lines earlier.
// kRefW (ICET_FLAG_REFERENCE_W): the per-voxel weight W as the reference computes it -- the float CompleteOrthogonalDecomposition of the full (not
// symmetrised) 3 x 3 L U^T R_noise U L^T -- instead of the double-precision pseudo-inverse of the default path.
template <int kT, int kStage, int kBlockT = kT, bool kRefW = false>
__device__ __forceinline__ void gn_solve_body(const int32_t* n_slots, const SlotFit* fitS, uint32_t* acc,
                                              float* X_all, float* xf_all, float* out, const AuxDev& aux,
                                              int V, int n, int iter, int runlen, const NearOverflow& over, int reject_moving, float* part, int nblk, float cond_bound2,
                                              const KeepArgs& keep = KeepArgs{}, int pair_of_block = -1) {
    // No contraction of a * b + c in this function: the bits of the per-voxel algebra must not depend on which instantiation the compiler is looking at (its choice of
    // what to fuse follows the surrounding code: after this body moved into a header the two-stage and the one-block form of one pair disagreed in last bits), and the
    // CPU restatement evaluates these expressions unfused as well.
#pragma clang fp contract(off)
    constexpr bool kCanon = kT == 512;
    static_assert(kStage == 0 || kCanon, "the two-stage form reduces in virtual blocks of 512 slots");
    static_assert(kBlockT == kT || (!kCanon && kStage == 0), "a guest block runs the one-block form of coarse grids only");
    __shared__ float J[27];
    __shared__ float red[kT / 64][27];
    __shared__ float vpart[kCanon ? kMaxVirtualBlocks : 1][27];     // the virtual blocks' sums when ONE block walks them all (stage 0, or stage 2 after a drain)
    const int pair = pair_of_block >= 0 ? pair_of_block : (kStage == 1 ? (int)blockIdx.x / nblk : (int)blockIdx.x), blk = kStage == 1 ? (int)blockIdx.x % nblk : 0;
    const bool own = kBlockT == kT || (int)threadIdx.x < kT;        // (a thread beyond kT: no slot, no row of the reduction table)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_first = blk * kT + (int)threadIdx.x, vb_stride = kStage == 1 ? nblk : 1;
    float* X = X_all + pair * 6;
    // The block is a chain of dependent latencies, so everything it will need is requested up front: the two counts, the Jacobian
    // table and -- speculatively, for slot threadIdx.x, before the number of slots is known (any slot < V is valid memory) -- the
    // first round's accumulator and fit records, as 16-byte loads.
    struct Rec { uint4 q[5]; };
    static_assert(sizeof(Rec) == kAccWords * 4 && sizeof(Rec) == sizeof(SlotFit), "80-byte records");
    const uint32_t nov = over.count[pair];                              // block-uniform
    const int ns = n_slots[pair];
    const float jmine = (threadIdx.x < 27) ? xf_all[pair * kXf + 16 + threadIdx.x] : 0.f;      // written by the previous update (write_xf)
    auto load_rec = [](const void* p) { Rec r; const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
        for (int k = 0; k < 5; k++) r.q[k] = q[k];
        return r; };
    if (kStage == 1 && nov) return;                                     // block-uniform (and the same in every block of the pair): stage 2 does the whole solve
    const bool from_partials = kStage == 2 && nov == 0u;                // stage 1 has reduced the slots already
    Rec accR{}, fitR{};
    if (!from_partials && own && s_first < V) { accR = load_rec(acc + ((size_t)pair * V + s_first) * kAccWords); fitR = load_rec(fitS + (size_t)pair * V + s_first); }
    if (nov) {
        drain_near_overflow(over, pair, V, xf_all + pair * kXf, acc + (size_t)pair * V * kAccWords, nov);
        __threadfence();                                                // this block reads the sums it has just added to
        __syncthreads();
        if (threadIdx.x == 0) over.count[pair] = 0u;
        if (own && s_first < V) accR = load_rec(acc + ((size_t)pair * V + s_first) * kAccWords);   // the speculative copy predates the drain
    }
    if (threadIdx.x < 27) J[threadIdx.x] = jmine;
    __syncthreads();
    __shared__ float s_keep_ref[12];
    __shared__ int s_keep_wtot[kT / 64];
    [[maybe_unused]] int keep_n = 0;
    if constexpr (kStage != 1 && kBlockT == kT) { if (keep.on) keep_n = keep_list_build<kT>(keep, pair, xf_all + pair * kXf, s_keep_ref, s_keep_wtot); }      // (block-uniform; ends with a barrier)
    if (ICET_SOLVE_PHASE == 1) { if (accR.q[0].x == 0x7FFFFFFFu && fitR.q[0].x == 0x7FFFFFFFu) out[0] = 1.f; return; }
    float S[27];
#pragma unroll
    for (int k = 0; k < 27; k++) S[k] = 0.f;
    const int nvb = from_partials ? 0 : (ns + kT - 1) / kT;             // rounds = virtual blocks of kT slots
    for (int vb = blk; vb < nvb; vb += vb_stride) {
      const int s = vb * kT + (int)threadIdx.x;
      if (own && s < ns) do {
        uint32_t* A = acc + ((size_t)pair * V + s) * kAccWords;
        if (s != s_first) { accR = load_rec(A); fitR = load_rec(fitS + (size_t)pair * V + s); }       // later rounds
        uint32_t aw[kAccWords];
        __builtin_memcpy(aw, &accR, sizeof(Rec));
        const uint32_t n2 = aw[0], m = aw[1];
        long long AF[9];
        __builtin_memcpy(AF, aw + 2, sizeof(AF));
        double sdD[3], sddD[6];
#pragma unroll
        for (int k = 0; k < 3; k++) sdD[k] = (double)AF[k] * kFixInv;
#pragma unroll
        for (int k = 0; k < 6; k++) sddD[k] = (double)AF[3 + k] * kFixInv;
        {
            uint4* z = reinterpret_cast<uint4*>(A);                  // ready for the next iteration
#pragma unroll
            for (int k = 0; k < 5; k++) z[k] = make_uint4(0u, 0u, 0u, 0u);
        }
        SlotFit f;
        __builtin_memcpy(&f, &fitR, sizeof(SlotFit));
        if (aux.n2_raw) aux.n2_raw[((size_t)pair * runlen + iter) * V + f.v] = (int)n2;
        if (aux.n2_in) aux.n2_in[((size_t)pair * runlen + iter) * V + f.v] = (int)m;
        if (!((int)n2 > n && (int)m > n)) break;                  // src/icet.cpp:290 (scan-2 half), :302  (`break` leaves the do { } while (0) of this slot)
        // mean and covariance of the m surviving points from the sums about mu1 (src/icet.cpp:303-306), in DOUBLE: the scatter in
        // a voxel's thin direction (1e-6 m^2 for a single-ring line) is what is left of sum(d d^T) ~ m |mu2 - mu1|^2 (1e-2) after the
        // subtraction -- in float that cancellation cost percents of the voxel's weight (round 2, scripts/diag_voxel.py)
        const double fmD = (double)m, rfmD = 1.0 / fmD;
        const double dbD[3] = {sdD[0] * rfmD, sdD[1] * rfmD, sdD[2] * rfmD};
        const float db[3] = {(float)dbD[0], (float)dbD[1], (float)dbD[2]};   // mean - mu1
        const float mu2[3] = {(float)((double)f.mu[0] + dbD[0]), (float)((double)f.mu[1] + dbD[1]), (float)((double)f.mu[2] + dbD[2])};
        const double denD = 1.0 / (double)(m - 1);
        const float d2 = (float)(n2 - 1);
        float cov2[6];
        cov2[0] = (float)((sddD[0] - fmD * dbD[0] * dbD[0]) * denD); cov2[1] = (float)((sddD[1] - fmD * dbD[0] * dbD[1]) * denD);
        cov2[2] = (float)((sddD[2] - fmD * dbD[0] * dbD[2]) * denD); cov2[3] = (float)((sddD[3] - fmD * dbD[1] * dbD[1]) * denD);
        cov2[4] = (float)((sddD[4] - fmD * dbD[1] * dbD[2]) * denD); cov2[5] = (float)((sddD[5] - fmD * dbD[2] * dbD[2]) * denD);
        // R_noise = sigma1/(|idx1|-1) + cov2/(|idx2|-1)                         src/icet.cpp:315
        float Rn[6];
#pragma unroll
        for (int k = 0; k < 6; k++) Rn[k] = f.s1n[k] + cov2[k] / d2;
        const float* M = f.M;
        // dz = M (mu2 - mu1)                                                       src/icet.cpp:335-337
        float dz[3];
#pragma unroll
        for (int i = 0; i < 3; i++) dz[i] = M[3 * i] * db[0] + M[3 * i + 1] * db[1] + M[3 * i + 2] * db[2];
        // extension (ICET_FLAG_REJECT_MOVING): a voxel whose compact residual is beyond the cutoff in a kept axis is a moving object
        if (reject_moving && iter >= kRejectMovingStartIter &&
            (fabsf(dz[0]) > kRejectMovingThresh || fabsf(dz[1]) > kRejectMovingThresh || fabsf(dz[2]) > kRejectMovingThresh)) break;
        // Rp = M Rn M^T  (M = L U^T)                                             src/icet.cpp:317
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
        float W9[9];
        if constexpr (kRefW) {
            // the matrix the reference hands to Eigen: ((L U^T R_n) U) L^T, all nine entries (the two triangles differ by roundings), then its float COD
            float Rp9[9];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) Rp9[3 * i + j] = MR[3 * i] * M[3 * j] + MR[3 * i + 1] * M[3 * j + 1] + MR[3 * i + 2] * M[3 * j + 2];
            icetdev::cod_pinv3_lane(Rp9, W9);
            W[0] = W9[0]; W[1] = W9[1]; W[2] = W9[2]; W[3] = W9[4]; W[4] = W9[5]; W[5] = W9[8];
        } else
        icetdev::pinv3_sym_fast(Rp, 3.0f * FLT_EPSILON, W);                 // src/icet.cpp:320-321
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
            if constexpr (kRefW) {                                  // (W is not exactly symmetric there)
                WH[j]      = W9[0] * Hz[j] + W9[1] * Hz[6 + j] + W9[2] * Hz[12 + j];
                WH[6 + j]  = W9[3] * Hz[j] + W9[4] * Hz[6 + j] + W9[5] * Hz[12 + j];
                WH[12 + j] = W9[6] * Hz[j] + W9[7] * Hz[6 + j] + W9[8] * Hz[12 + j];
            } else {
            WH[j]      = W[0] * Hz[j] + W[1] * Hz[6 + j] + W[2] * Hz[12 + j];
            WH[6 + j]  = W[1] * Hz[j] + W[3] * Hz[6 + j] + W[4] * Hz[12 + j];
            WH[12 + j] = W[2] * Hz[j] + W[4] * Hz[6 + j] + W[5] * Hz[12 + j];
            }
        }
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = a; b < 6; b++) { S[q] += Hz[a] * WH[b] + Hz[6 + a] * WH[6 + b] + Hz[12 + a] * WH[12 + b]; q++; }
        }
#pragma unroll
        for (int a = 0; a < 6; a++) S[21 + a] += WH[a] * dz[0] + WH[6 + a] * dz[1] + WH[12 + a] * dz[2];
      } while (0);
      if (kCanon) {                                                   // this virtual block's 27 sums, now
#pragma unroll
          for (int k = 0; k < 27; k++) { float t = wave_total(S[k]); if (lane == 0) red[wave][k] = t; S[k] = 0.f; }
          __syncthreads();
          if (threadIdx.x < 27) {
              float t = 0.f; for (int w = 0; w < kT / 64; w++) t += red[w][threadIdx.x];
              if (kStage == 1) part[((size_t)pair * kMaxVirtualBlocks + vb) * 27 + threadIdx.x] = t; else vpart[vb][threadIdx.x] = t;
          }
          __syncthreads();
      }
    }
    if (ICET_SOLVE_PHASE == 2) { float t = 0.f; for (int k = 0; k < 27; k++) t += S[k]; if (t == 1.2345e-30f) out[0] = t; return; }
    if (kStage == 1) return;
    // Two waves stay for the 6 x 6 part: the first does what it always did; the second waits for the matrix and, on the literal route, takes pinv(HTWH) beside the first
    // wave's eigen-decomposition (two words in LDS hand over: the matrix is staged / the pseudo-inverse is in the workspace).
    __shared__ int s_tail_flag[3];                                  // [0] the sums are staged, [1] the pseudo-inverse is in the workspace, [2] the second wave has read the sums
    if (threadIdx.x == 0) { s_tail_flag[0] = 0; s_tail_flag[1] = 0; s_tail_flag[2] = 0; }
    if (!kCanon) {
#pragma unroll
        for (int k = 0; k < 27; k++) { float t = wave_total(S[k]); if (lane == 0 && wave < kT / 64) red[wave][k] = t; }      // DPP scan, not 6 x 27 trips through the LDS crossbar
    }
    __syncthreads();
    if (wave > 1) return;
    volatile int* tail_flag = s_tail_flag;
    // The 6 x 6 part runs on lane 0 of the first wave; its lanes assemble the input and write the results out (one lane doing the
    // 27 four-way sums and ~100 scalar stores was a quarter of the tail).
    __shared__ float stage[kXf + 48];                               // transform record | X, pred_stds, covariance
    __shared__ icetdev::GnTailWs tail_ws;
    if (wave == 1) {
        while (tail_flag[0] == 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float Hh[36];
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) { const float t = stage[q]; Hh[a * 6 + b] = t; Hh[b * 6 + a] = t; q++; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (lane == 0) tail_flag[2] = 1;                            // (the first wave reuses `stage` for its results)
        gn_tail_helper(Hh, cond_bound2, tail_ws, tail_flag + 1);
        return;
    }
    if (lane < 27) {
        float t = 0.f;
        if (kCanon) {                                               // the virtual blocks in index order, from wherever they were reduced
            const int nv = (ns + kT - 1) / kT;
            if (from_partials) { for (int b = 0; b < nv; b++) t += part[((size_t)pair * kMaxVirtualBlocks + b) * 27 + lane]; }
            else { for (int b = 0; b < nv; b++) t += vpart[b][lane]; }
        } else { for (int w = 0; w < kT / 64; w++) t += red[w][lane]; }
        stage[lane] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) tail_flag[0] = 1;                                // the second wave may read the sums
    float Hm[36], g[6];
    {
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) { const float t = stage[q]; Hm[a * 6 + b] = t; Hm[b * 6 + a] = t; q++; }
#pragma unroll
        for (int a = 0; a < 6; a++) g[a] = stage[21 + a];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    if (ICET_SOLVE_PHASE == 3) { float t = 0.f; for (int k = 0; k < 36; k++) t += Hm[k]; if (t == 1.2345e-30f) out[0] = t; return; }
    float ev[6], cov[36], ps[6], dx[6];
    int route, pruned;
    gn_tail(Hm, g, cond_bound2, cov, ps, dx, ev, route, pruned, tail_ws, lane == 0, tail_flag + 1);
    // (every lane of the wave ran the scalar algebra above on the same inputs -- a wave costs what a lane costs -- so the results are
    // wave-uniform; lane 0 stages them and the lanes store them)
    if (ICET_SOLVE_PHASE == 4) { float t = 0.f; for (int k = 0; k < 36; k++) t += cov[k]; for (int k = 0; k < 6; k++) t += dx[k] + ps[k]; if (t == 1.2345e-30f) out[0] = t; return; }
    float Xn[6];
    for (int k = 0; k < 6; k++) Xn[k] = X[k] + dx[k];
    while (tail_flag[2] == 0) __builtin_amdgcn_s_sleep(1);          // (long since: the second wave read the sums before this wave's Cholesky attempt was over)
    if (lane == 0) {
        write_xf(stage, Xn);
        float* r = stage + kXf;
        for (int k = 0; k < 6; k++) { r[k] = Xn[k]; r[6 + k] = ps[k]; }
        for (int k = 0; k < 36; k++) r[12 + k] = cov[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    if constexpr (kStage != 1 && kBlockT == kT) { if (keep.on == 1) keep_budget_check(keep, pair, stage, s_keep_ref, lane, keep_n); }
    if (lane < kXf) xf_all[pair * kXf + lane] = stage[lane];
    if (aux.xf_last && iter == runlen - 2 && lane < kXf) aux.xf_last[pair * kXf + lane] = stage[lane];      // what the last point pass will use (`points2`)
    // (the results of the LAST iteration are what a caller reads; a sequential caller's `out` is pinned host memory, where every iteration's 48 stores cross PCIe before the kernel can end)
    if (lane < 48 && iter == runlen - 1) out[(size_t)(aux.pair_user ? aux.pair_user[pair] : pair) * 48 + lane] = stage[kXf + lane];      // (the caller's pair of this slot: ragged throughput batches)
    if (lane < 6) X[lane] = stage[kXf + lane];
    if (aux.done_flag) {                                          // a sequential caller watches this word of pinned host memory instead of synchronising the stream: results first, then the word
        __threadfence_system();
        if (lane == 0) *reinterpret_cast<volatile int32_t*>(aux.done_flag) = 1;
    }
    if (lane == 0) {
        if (aux.x_hist) for (int k = 0; k < 6; k++) aux.x_hist[((size_t)pair * runlen + iter) * 6 + k] = Xn[k];
        if (aux.htwh) for (int k = 0; k < 36; k++) aux.htwh[((size_t)pair * runlen + iter) * 36 + k] = Hm[k];
        if (aux.htwdz) for (int k = 0; k < 6; k++) aux.htwdz[((size_t)pair * runlen + iter) * 6 + k] = g[k];
        if (aux.cond) { float* ci = aux.cond + ((size_t)pair * runlen + iter) * 8; for (int k = 0; k < 6; k++) ci[k] = ev[k]; ci[6] = (float)pruned; ci[7] = (float)route; }
    }
}

}  // namespace
}  // namespace icet
