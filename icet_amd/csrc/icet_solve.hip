// icet_amd/csrc/icet_solve.hip -- the per-voxel and 6x6 part of one Gauss-Newton iteration (ICET::fitScan2,
// /root/reference/src/icet.cpp:372-436, after the point pass of icet_accumulate.hip):
//     k_init_state   X = X0, transform record
//     k_gn_solve     per-voxel fitCells2 algebra (:314-338), block reduction of H^T W H and H^T W dz (:401-402), 6x6
//                    covariance (:410-417), conditioning (:443-492), dx and X += dx (:427-433); one block per pair
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include "icet_internal.h"
#include "icet_device_common.h"
#include "icet_device_math.h"
#include "icet_solve_body.h"

namespace icet {
namespace {

// (desc / n2: the scan-2 row counts that only the device knows -- k_patch_counts' job, folded in when this is the loop's first kernel: one launch
// less in front of a sequential caller's first iteration)
__global__ void k_init_state(const float* __restrict__ x0, float* __restrict__ X, float* __restrict__ xf, int n_pairs, float* __restrict__ xf_last,
                             PairDesc* __restrict__ desc, const int32_t* __restrict__ n2, uint32_t* __restrict__ done, KeepState* __restrict__ keep, int32_t* __restrict__ keep_modes,
                             const PairDesc* __restrict__ h_desc, const int32_t* __restrict__ h_seg, int32_t* __restrict__ seg, const int32_t* __restrict__ pair_user) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    // (h_desc: k_upload_desc's job as well, when this is the first kernel that reads the descriptors -- the register half of a sequential caller: one launch less)
    if (h_desc) { desc[p] = h_desc[p]; seg[p] = h_seg[p]; if (p == 0) seg[n_pairs] = h_seg[n_pairs]; }
    if (done) done[p] = 0u;                                      // the point pass' block tickets (k_gn_accumulate_solve)
    if (keep) {                                                  // every solve starts on the whole scan, the pairs in index order
        keep[p].mode = 0; keep[p].n_keep = 0; keep[p].list_passes = 0; keep[p].builds = 0;
        keep_modes[p] = 0; keep_modes[n_pairs + p] = 0;
    }
    if (n2) desc[p].n2 = max(0, min(n2[p], desc[p].n2));
    float x[6];
    const int pu = pair_user ? pair_user[p] : p;                 // (a ragged throughput batch sits XCD-balanced in the tables: the caller's X0 of this slot's pair)
    for (int k = 0; k < 6; k++) { x[k] = x0 ? x0[pu * 6 + k] : 0.f; X[p * 6 + k] = x[k]; }
    write_xf(xf + p * kXf, x);
    if (xf_last) for (int k = 0; k < kXf; k++) xf_last[p * kXf + k] = xf[p * kXf + k];      // a one-iteration solve transforms scan 2 by X0 (`points2`)
}

// One launch per iteration behind the point pass: gn_solve_body (icet_solve_body.h) with the block it was written for.
template <int kT, int kStage, bool kRefW = false>
__global__ __launch_bounds__(kT) void k_gn_solve(const int32_t* __restrict__ n_slots, const SlotFit* __restrict__ fitS, uint32_t* __restrict__ acc,
                                                     float* __restrict__ X_all, float* __restrict__ xf_all, float* __restrict__ out, AuxDev aux,
                                                     int V, int n, int iter, int runlen, NearOverflow over, int reject_moving, float* __restrict__ part, int nblk, float cond_bound2, KeepArgs keep) {
    gn_solve_body<kT, kStage, kT, kRefW>(n_slots, fitS, acc, X_all, xf_all, out, aux, V, n, iter, runlen, over, reject_moving, part, nblk, cond_bound2, keep);
}


// Test hook (icet_debug_gn_tail): n independent (HTWH, HTWdz), one WAVE each, through the SAME gn_tail k_gn_solve runs.
// out: 56 floats per matrix: cov[36] | pred_stds[6] | dx[6] | eigenvalues[6] | pruned | route.
__global__ __launch_bounds__(64) void k_gn_tail_debug(const float* __restrict__ H, const float* __restrict__ gv, float* __restrict__ out, int n, float bound2) {
    const int i = blockIdx.x;
    float Hm[36], g[6], cov[36], ps[6], dx[6], ev[6];
    for (int k = 0; k < 36; k++) Hm[k] = H[(size_t)i * 36 + k];
    for (int k = 0; k < 6; k++) g[k] = gv[(size_t)i * 6 + k];
    int route, pruned;
    __shared__ icetdev::GnTailWs tail_ws;
#ifdef ICET_TAIL_TIMING
    const unsigned long long t_in = wall_clock64();
#endif
    gn_tail(Hm, g, bound2, cov, ps, dx, ev, route, pruned, tail_ws, threadIdx.x == 0);
    if (threadIdx.x != 0) return;
#ifdef ICET_TAIL_TIMING
    if (i == 0) { const unsigned long long t_out = wall_clock64(); const unsigned long long* ts = tail_ws.ts;
        printf("tail (10 ns ticks): whole %llu | entry %llu pinv(H) %llu eig %llu prune %llu setup %llu products %llu pinv(innards) %llu products %llu exit %llu\n", t_out - t_in, ts[0] - t_in, ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], 0ull, ts[5] - ts[4], ts[6] - ts[5], t_out - ts[6]); }
#endif
    float* o = out + (size_t)i * 56;
    for (int k = 0; k < 36; k++) o[k] = cov[k];
    for (int k = 0; k < 6; k++) { o[36 + k] = ps[k]; o[42 + k] = dx[k]; o[48 + k] = ev[k]; }
    o[54] = (float)pruned; o[55] = (float)route;
}
// Test hook (icet_debug_pinv3): n independent 3 x 3 matrices through the per-voxel pseudo-inverse of ICET_FLAG_REFERENCE_W, one lane each.
__global__ __launch_bounds__(64) void k_pinv3_debug(const float* __restrict__ A, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float a[9], w[9];
    for (int k = 0; k < 9; k++) a[k] = A[(size_t)i * 9 + k];
    icetdev::cod_pinv3_lane(a, w);
    for (int k = 0; k < 9; k++) out[(size_t)i * 9 + k] = w[k];
}
}  // namespace

hipError_t launch_pinv3_debug(const float* d_A, float* d_out, int n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    k_pinv3_debug<<<(n + 63) / 64, 64, 0, st>>>(d_A, d_out, n);
    return hipGetLastError();
}

hipError_t launch_gn_tail_debug(const float* d_H, const float* d_g, float* d_out, int n, float bound2, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    k_gn_tail_debug<<<n, 64, 0, st>>>(d_H, d_g, d_out, n, bound2);
    return hipGetLastError();
}

#define ICET_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

// Row counts that only the device knows (a sequential caller's range filter, include/icet_nodes.h): the descriptors were uploaded with upper
// bounds -- the launch geometry is sized from those -- and take the actual counts here; every kernel reads n1 / n2 from the descriptor.
__global__ void k_patch_counts(PairDesc* __restrict__ desc, const int32_t* __restrict__ n1, const int32_t* __restrict__ n2, int n_pairs) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    if (n1) desc[p].n1 = max(0, min(n1[p], desc[p].n1));
    if (n2) desc[p].n2 = max(0, min(n2[p], desc[p].n2));
}

hipError_t launch_patch_counts(const Workspace& w, const LaunchCfg& c, const int32_t* d_n1, const int32_t* d_n2, hipStream_t st) {
    k_patch_counts<<<(c.n_pairs + 63) / 64, 64, 0, st>>>(w.desc, d_n1, d_n2, c.n_pairs);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

__global__ __launch_bounds__(256) void k_upload_desc(const uint32_t* __restrict__ h_desc, uint32_t* __restrict__ desc, int n_desc_words, const int32_t* __restrict__ h_seg, int32_t* __restrict__ seg, int n_seg) {
    for (int i = threadIdx.x; i < n_desc_words; i += 256) desc[i] = h_desc[i];
    for (int i = threadIdx.x; i < n_seg; i += 256) seg[i] = h_seg[i];
}
hipError_t launch_upload_desc(const Workspace& w, const PairDesc* h_desc, const int32_t* h_seg, int n_pairs, hipStream_t st) {
    static_assert(sizeof(PairDesc) % 4 == 0, "copied as 32-bit words");
    k_upload_desc<<<1, 256, 0, st>>>(reinterpret_cast<const uint32_t*>(h_desc), reinterpret_cast<uint32_t*>(w.desc), n_pairs * (int)(sizeof(PairDesc) / 4), h_seg, w.seg_off, n_pairs + 1);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_init_state(const Workspace& w, const LaunchCfg& c, const float* d_x0, hipStream_t st, float* xf_last, const int32_t* d_n2, const PairDesc* h_desc, const int32_t* h_seg) {
    k_init_state<<<(c.n_pairs + 63) / 64, 64, 0, st>>>(d_x0, w.X, w.xf, c.n_pairs, xf_last, w.desc, d_n2, w.gn_done(), c.keep ? w.keep_state : nullptr, w.keep_modes, h_desc, h_seg, w.seg_off, c.pair_user);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

// `points2` member of the reference object (include/icet.h:80; src/icet.cpp:375-378): pair 0's scan 2 under the transform the last point pass
// uses (AuxDev::xf_last) -- the same transform_point, so the same bits.  `out` may be pinned host memory (one coalesced write per
// coordinate); runs on a side stream beside the last iteration.
__global__ __launch_bounds__(kBlock) void k_points2(const PairDesc* __restrict__ desc, const float* __restrict__ xf, float* __restrict__ out) {
    __shared__ float sxf[12];                                     // the record may live in pinned host memory: one read per block, not per point
    if (threadIdx.x < 12) sxf[threadIdx.x] = xf[threadIdx.x];
    __syncthreads();
    const PairDesc d = desc[0];
    const float* px = d.s2; const float* py = px + d.ld2; const float* pz = px + 2 * (size_t)d.ld2;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.n2; i += gridDim.x * kBlock) {
        float qx, qy, qz;
        transform_point(px[i], py[i], pz[i], sxf, qx, qy, qz);
        out[i] = qx; out[(size_t)d.n2 + i] = qy; out[2 * (size_t)d.n2 + i] = qz;
    }
}

hipError_t launch_points2(const Workspace& w, const LaunchCfg& c, const float* xf, float* out, hipStream_t st) {
    if (c.max_n2 <= 0) return hipSuccess;
    const int blocks = std::min(1024, (c.max_n2 + kBlock - 1) / kBlock);
    k_points2<<<blocks, kBlock, 0, st>>>(w.desc, xf, out);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_gn_solve(const Workspace& w, const LaunchCfg& c, int iter, float* d_out, const AuxDev* auxp, hipStream_t st, int keep_pass) {
    AuxDev aux{}; if (auxp) aux = *auxp;
    aux.pair_user = c.pair_user;
    aux.done_flag = (iter == c.runlen - 1) ? c.done_flag : nullptr;
    const KeepArgs keep{w.desc, w.keep_mask, w.keep_list, w.keep_state, c.keep_bt * c.keep_bt * c.keep_check_scale * c.keep_check_scale, c.keep_br * c.keep_br * c.keep_check_scale * c.keep_check_scale, w.keep_state ? keep_pass : 0,      // 1: build / check; 2: behind the last pass (statistics only)
                        w.keep_modes ? w.keep_modes + (size_t)((iter + 1) & 1) * c.n_pairs : nullptr, c.n_pairs};
    NearOverflow over{w.desc, w.slot_of_voxel, w.hotS, w.thr, w.near_over, w.near_over_count, c.T, c.P, c.rt2};
#define ICET_SOLVE_LAUNCHES(RW) do {                                                                                                                                   \
    if (c.V > 4096 && c.n_pairs <= kTwoStageMaxPairs && w.gn_part) {                                                                                                      \
        /* two stages: several blocks per pair reduce their share of the slots to 27 partial sums each, one block per pair adds them and solves */                        \
        const int nblk = kTwoStageBlocks;                                                                                                                                 \
        k_gn_solve<512, 1, RW><<<c.n_pairs * nblk, 512, 0, st>>>(w.n_slots, w.fitS, w.acc, w.X, w.xf, d_out, aux, c.V, c.n, iter, c.runlen, over, c.reject_moving, w.gn_part, nblk, c.gn_cond_bound2, keep); \
        ICET_LAUNCH_CHECK();                                                                                                                                              \
        k_gn_solve<512, 2, RW><<<c.n_pairs, 512, 0, st>>>(w.n_slots, w.fitS, w.acc, w.X, w.xf, d_out, aux, c.V, c.n, iter, c.runlen, over, c.reject_moving, w.gn_part, nblk, c.gn_cond_bound2, keep); \
    }                                                                                                                                                                     \
    else if (c.V > 4096) k_gn_solve<512, 0, RW><<<c.n_pairs, 512, 0, st>>>(w.n_slots, w.fitS, w.acc, w.X, w.xf, d_out, aux, c.V, c.n, iter, c.runlen, over, c.reject_moving, nullptr, 1, c.gn_cond_bound2, keep); \
    else k_gn_solve<kBlock, 0, RW><<<c.n_pairs, kBlock, 0, st>>>(w.n_slots, w.fitS, w.acc, w.X, w.xf, d_out, aux, c.V, c.n, iter, c.runlen, over, c.reject_moving, nullptr, 1, c.gn_cond_bound2, keep); \
} while (0)
    if (c.ref_w) ICET_SOLVE_LAUNCHES(true); else ICET_SOLVE_LAUNCHES(false);
#undef ICET_SOLVE_LAUNCHES
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace icet
