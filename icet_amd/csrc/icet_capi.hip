// icet_amd/csrc/icet_capi.hip -- host side of the C ABI declared in include/icet_hip.h.
//
// Owns the device workspace, stages host scans into HBM, enqueues the keyframe build and the
// Gauss-Newton loop on one HIP stream and copies the 48 result floats per pair back.  No CPU
// implementation of the algorithm lives here: if the device or the kernels are unavailable every
// entry point returns ICET_ERR_NO_DEVICE / ICET_ERR_HIP.
#include "../../include/icet_hip.h"
#include "../../include/icet_nodes.h"
#include "icet_internal.h"
#include "icet_layout.h"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <cstdint>
#include <new>

using namespace icet;

struct icet_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    Workspace w;
    Tuning tune;                    // icet_set_option
    int32_t kf_pairs = 0; icet_params kf_params{};      // a keyframe parked by icet_keyframe_device (0 pairs = none)
    int max_lds = 160 * 1024;       // hipDeviceAttributeMaxSharedMemoryPerBlock of the device
    int lds_rank_ok = 0;            // this device passed lds_rank_selftest (icet_create)
    std::string err;
    // host staging (pinned) for descriptors and results
    PairDesc* h_desc = nullptr; int32_t* h_seg = nullptr; int32_t h_cap_pairs = 0;
    bool desc_kf_valid = false, desc_reg_valid = false;   // the pinned descriptors hold what the last icet_keyframe_device_n / icet_register_device_n call wrote (any other writer, and a re-allocation, clears both)
    PairDesc* h_desc_rt = nullptr; int32_t h_cap_rt = 0;       // ICET_FLAG_ROUNDTRIP_SCAN2: descriptors of the round-tripped copy of scan 2
    // device staging for host-pointer entry points
    float* d_stage1 = nullptr; float* d_stage2 = nullptr; int64_t cap_stage1 = 0, cap_stage2 = 0;
    float* d_out = nullptr; float* d_x0 = nullptr; int32_t cap_out_pairs = 0;
    float* h_out = nullptr;
    // aux (single pair): every side table of a solve lives in ONE device block (`d_pack`, words of 4 bytes, layout aux_layout()) behind the
    // 48 result floats, so that results and side tables come back in one DMA into the pinned `h_pack`
    AuxDev aux_dev{}; int aux_V = 0, aux_runlen = 0;
    uint32_t* d_pack = nullptr; uint32_t* h_pack = nullptr; size_t cap_pack = 0;
    float* h_pts2 = nullptr; float* d_pts2 = nullptr; size_t cap_pts2 = 0;   // `points2` (scan 2 under the last iteration's transform): device buffer + pinned host copy
    float* h_x0 = nullptr;                                       // pinned, 6 x cap_out_pairs
    float* d_sph1 = nullptr; int32_t* d_idx1 = nullptr; size_t cap_side1 = 0;      // points1Spherical / pointIndices1 on request (icet_sidetables.hip)
    float* d_sph2 = nullptr; int32_t* d_vox2 = nullptr; size_t cap_side2 = 0;      // points2Spherical / the rows' voxels on request
    // host-pointer entry points: scan 2 is uploaded on a stream of its own, beside the keyframe build of scan 1
    hipStream_t st_copy = nullptr; hipEvent_t ev_s2 = nullptr;
    hipEvent_t ev_kf = nullptr, ev_kfd = nullptr, ev_prev = nullptr, ev_pts2 = nullptr;   // keyframe built / its tables on the host / transform of the last iteration known / points2 on the host
    // icet_solve_begin .. icet_solve_end
    struct Pending { bool active = false; float* x_out = nullptr; float* ps_out = nullptr; float* cov_out = nullptr; icet_aux aux{}; bool has_aux = false;
                     int V = 0, rl = 0; int64_t n2 = 0; bool kf_tables = false, kf_done = false, pts2 = false, pts2_dev = false, tail_ints = false, side1 = false, side2 = false; int64_t n1 = 0;
                     const float* scan2 = nullptr; int64_t ld2 = 0; } pend;
    // timing
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;
    std::vector<hipEvent_t> ev_acc;
    float last_ms[4] = {0, 0, 0, 0};
    bool timing_valid = false;
    int last_iters = 0;
    // Large device batches are cut into contiguous parts, each solved by a helper context on its own stream, so that
    // the keyframe build of one part (latency / LDS bound) overlaps the Gauss-Newton loop of another (VALU bound).
    std::vector<icet_ctx*> helpers;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_stage = nullptr; int stage_at = 0;            // see LaunchCfg::stage_event
    hipEvent_t ev_desc = nullptr; bool desc_in_flight = false;   // completion of the last copy out of the pinned descriptor staging
    // Small device batches whose launch geometry repeats call after call are replayed from a captured hipGraph (option "graph"): the ~33
    // launches of a single-pair solve then cost one hipGraphLaunch on the host, and the command processor runs them back to back.
    struct GraphKey { int64_t v[45]; };                        // every LaunchCfg field + the pointers the launches take + the prologue's key (graph_key_of)
    struct GraphSlot { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; GraphKey key{}, seen{}; bool have_seen = false, have_graph = false; };
    bool capturing = false; int graph_mode = -1;               // -1: replay batches of <= 8 pairs whose launch key repeats; 0 never; 1 same as -1
    GraphSlot g_solve, g_keyframe, g_loop;                     // the whole solve (icet_solve_batch_device) and its two halves (icet_keyframe_device_n / icet_register_device_n)
    hipEvent_t ev_graph = nullptr; bool graph_in_flight = false;
    // A caller inside this library (the sequential nodes, icet_nodes.hip) can put work of its own at the head of the NEXT icet_register_device_n call's launch
    // sequence -- enqueued on the context's stream right before the loop's first kernel, captured into the same graph: the range filter and the loop of a frame
    // are then ONE hipGraphLaunch (round 6: the loop's graph used to start 30-40 us after the filter's last kernel).  `prologue_key` names what the hook's
    // launches depend on (buffers, grid): it is part of the graph key.  Cleared by the caller after the call (icet_ctx_set_prologue).
    // A ragged throughput batch is laid out XCD-balanced (solve_device_part): slot s of the internal tables holds the caller's pair h_seg[n_pairs + 1 + s]
    bool perm_active = false; int32_t perm_pairs = 0;
    hipError_t (*prologue)(void*, hipStream_t) = nullptr; void* prologue_user = nullptr; int64_t prologue_key = 0;
    // ... and have the LAST solve of the next icet_register_device_n call store 1 into a word of (coherent) pinned host memory once the results are written: the caller
    // watches that word instead of synchronising the stream (icet_ctx_set_done_flag; part of the graph key)
    int32_t* done_flag = nullptr;
    // icet_sync after ONE small device-resident solve (icet_solve_batch_device, replayed graph) watches a word of its own the same way: h_sync_word, raised by that solve's
    // last kernel.  armed_calls counts such solves since the last icet_sync; anything else enqueued on the context (or a second solve, whose reset of the word races with the
    // first one's store) makes it 2 or more and icet_sync synchronises the stream as before.
    int32_t* h_sync_word = nullptr; int armed_calls = 2;
};

namespace {

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); \
    return e_ == hipErrorOutOfMemory ? ICET_ERR_NOMEM : ICET_ERR_HIP; } } while (0)

template <typename T> hipError_t dev_realloc(T*& p, size_t count) {
    if (p) { hipError_t e = hipFree(p); p = nullptr; if (e != hipSuccess) return e; }
    if (count == 0) return hipSuccess;
    return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
}

bool params_ok(const icet_params* p) {
    if (!p) return false;
    if (p->runlen < 0 || p->runlen > 4096) return false;
    if (p->bins_phi <= 0 || p->bins_theta <= 0 || p->n < 1) return false;
    return true;
}

icet_status ensure_workspace(icet_ctx* c, const icet_params* p, int32_t n_pairs, int64_t total_n1, int64_t total_n2) {
    Workspace& w = c->w;
    const int V = p->bins_phi * p->bins_theta;
    if ((int64_t)p->bins_phi * p->bins_theta > kMaxVoxels) { c->err = "bins_phi*bins_theta exceeds the voxel limit (10000: the multi-split keeps 12 B per voxel, the Gauss-Newton pass a 2-byte map entry and its look-up tables, in one block's LDS)"; return ICET_ERR_UNSUPPORTED; }
    if ((size_t)V * 12 + 8 + 4096 > (size_t)c->max_lds) { c->err = "grid too fine for this device's LDS (k_bin_scatter keeps 12 B per voxel in one block)"; return ICET_ERR_UNSUPPORTED; }
    if (total_n1 >= (int64_t)1 << 31) { c->err = "total scan-1 points per call must be < 2^31"; return ICET_ERR_UNSUPPORTED; }
    const bool grow_pairs = n_pairs > w.cap_pairs || V > w.cap_V;
    if (grow_pairs) {
        c->kf_pairs = 0;                       // the parked keyframe's tables (hotS / fitS / slot_of_voxel / n_slots / acc) are about to be re-allocated
        const int np = n_pairs > w.cap_pairs ? n_pairs : w.cap_pairs;
        const int VV = V > w.cap_V ? V : w.cap_V;
        const size_t pv = (size_t)np * VV;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, dev_realloc(w.desc, np));
        HIPCHK(c, dev_realloc(w.seg_off, 2 * (size_t)np + 1));              // segment offsets, then (ragged throughput batches) the caller's pair of every slot
        HIPCHK(c, dev_realloc(w.bin_count, pv));
        HIPCHK(c, dev_realloc(w.bin_start, (size_t)np * (VV + 1)));
        HIPCHK(c, dev_realloc(w.live_bins, pv * 4)); HIPCHK(c, dev_realloc(w.n_live, np));
        HIPCHK(c, dev_realloc(w.hotD, pv)); HIPCHK(c, dev_realloc(w.fitD, pv)); HIPCHK(c, dev_realloc(w.activeD, pv)); HIPCHK(c, dev_realloc(w.midD, pv));
        HIPCHK(c, dev_realloc(w.hotS, pv)); HIPCHK(c, dev_realloc(w.fitS, pv));
        HIPCHK(c, dev_realloc(w.slot_of_voxel, (size_t)np * ((VV + 1) & ~1)));
        HIPCHK(c, dev_realloc(w.n_slots, np)); HIPCHK(c, dev_realloc(w.near_over_count, 2 * (size_t)np));
        HIPCHK(c, dev_realloc(w.acc, pv * kAccWords));
        HIPCHK(c, hipMemset(w.acc, 0, pv * kAccWords * sizeof(uint32_t)));
        HIPCHK(c, dev_realloc(w.xf, (size_t)np * 48));
        HIPCHK(c, dev_realloc(w.X, (size_t)np * 6));
        HIPCHK(c, dev_realloc(w.gn_part, (size_t)kGnPartWords));
        HIPCHK(c, dev_realloc(w.flags, np));
        HIPCHK(c, dev_realloc(w.zero_rows, (size_t)np * 4));
        HIPCHK(c, dev_realloc(w.vrange, (size_t)np * 2));
        HIPCHK(c, dev_realloc(w.splitters, (size_t)np * kRankSortMaxBuckets)); HIPCHK(c, dev_realloc(w.n_buckets, np)); HIPCHK(c, dev_realloc(w.bucket_start, (size_t)np * (kRankSortMaxBuckets + 1)));
        if (c->h_desc) { HIPCHK(c, hipHostFree(c->h_desc)); c->h_desc = nullptr; }
        if (c->h_seg) { HIPCHK(c, hipHostFree(c->h_seg)); c->h_seg = nullptr; }
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_desc), sizeof(PairDesc) * np));
        std::memset(c->h_desc, 0, sizeof(PairDesc) * np); c->desc_kf_valid = c->desc_reg_valid = false;
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_seg), sizeof(int32_t) * (2 * (size_t)np + 1)));
        c->h_cap_pairs = np;
        w.cap_pairs = np; w.cap_V = VV;
    }
    if (total_n2 >= (int64_t)1 << 31) { c->err = "total scan-2 points per call must be < 2^31"; return ICET_ERR_UNSUPPORTED; }
    if (total_n2 > w.cap_n2) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, dev_realloc(w.near_over, (size_t)total_n2));
        w.cap_n2 = total_n2;
    }
    if (total_n1 > w.cap_n1 || grow_pairs) {
        const int64_t n = total_n1 > w.cap_n1 ? total_n1 : w.cap_n1;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (n > w.cap_n1) {
            c->kf_pairs = 0;
            HIPCHK(c, dev_realloc(w.r1, n)); HIPCHK(c, dev_realloc(w.cart1, (size_t)3 * n));
            HIPCHK(c, dev_realloc(w.key64A, n)); HIPCHK(c, dev_realloc(w.key64B, n)); HIPCHK(c, dev_realloc(w.bin16, n)); HIPCHK(c, dev_realloc(w.binpos, n)); HIPCHK(c, dev_realloc(w.bkt, n));
            HIPCHK(c, dev_realloc(w.keyA, n)); HIPCHK(c, dev_realloc(w.keyB, n)); HIPCHK(c, dev_realloc(w.valA, n)); HIPCHK(c, dev_realloc(w.valB, n));
            HIPCHK(c, dev_realloc(w.pred, n)); HIPCHK(c, dev_realloc(w.src, n));
            w.cap_n1 = n;
        }
        HIPCHK(c, dev_realloc(w.execbits, (size_t)(w.cap_n1 / 64 + w.cap_pairs + 2)));      // exec_word_base: off1 / 64 + pair
        {
            const size_t need_items = (size_t)(w.cap_n1 / 64 + (int64_t)w.cap_pairs * (w.cap_V + 1) + 64);
            if (need_items > w.cap_fit_items) {
                if (w.fit_items) { HIPCHK(c, hipFree(w.fit_items)); w.fit_items = nullptr; }
                HIPCHK(c, hipMalloc(&w.fit_items, need_items * 16));
                w.cap_fit_items = need_items;
            }
            HIPCHK(c, dev_realloc(w.fit_n_items, (size_t)w.cap_pairs));
        }
#ifdef ICET_DIAG_LIBSORT
        const size_t need = sort_temp_bytes(w.cap_n1);
        if (need > w.sort_tmp_bytes) {
            if (w.sort_tmp) { HIPCHK(c, hipFree(w.sort_tmp)); w.sort_tmp = nullptr; }
            HIPCHK(c, hipMalloc(&w.sort_tmp, need));
            w.sort_tmp_bytes = need;
        }
#endif
    }
    return ICET_OK;
}

// thr[k] = smallest float32 whose reference bin index int((double)a / period * nb) (src/icet.cpp:545-546) is >= k.
void build_thresholds(int nb, double period, float* out) {
    auto bin = [&](float t) { return static_cast<long long>(((double)t / period) * nb); };
    out[0] = 0.f;
    for (int k = 1; k <= nb; k++) {
        float t = (float)(period * (double)k / (double)nb);
        while (bin(t) >= k) t = std::nextafterf(t, -INFINITY);
        while (bin(t) < k) t = std::nextafterf(t, INFINITY);
        out[k] = t;
    }
}

// Classification LUT over a monotone coordinate c in [lo, lo + range): M cells, each naming the edge nearest to its centre.
// M is the smallest power of two whose cells are narrower than 0.45 x a reference bin width -- the narrowest bin
// (quantile 0: azimuth, whose bins differ by 2x at most) or a low quantile of the widths (polar angle: in w = -cos(phi) the
// two bins at the poles are 15x narrower than those at the horizon, where the lidar's points are; sizing the table for them
// made it 8 KB of LDS and block start-up time for rows nobody visits).  A cell is usable by the fast path iff no OTHER edge
// lies between any of its points and the edge it names, nor within the guard band of the cell: then "compare with the named
// edge" yields the bin and "far from the named edge" implies far from every edge.  Cells that fail this (near the poles) get
// edge = NaN, which fails the kernel's guard test, so their points take the literal path.
struct HostCell { float edge; int32_t idx; };
int build_lut(const std::vector<double>& edges, double lo, double range, double quantile, double guard, std::vector<HostCell>& out, int max_cells = 1 << 16) {
    std::vector<double> widths;
    for (size_t k = 1; k < edges.size(); k++) widths.push_back(edges[k] - edges[k - 1]);
    std::sort(widths.begin(), widths.end());
    const double w_ref = widths.empty() ? range : widths[(size_t)(quantile * (double)(widths.size() - 1))];
    int M = 64;
    while (range / M > 0.45 * w_ref && 2 * M <= max_cells) M *= 2;     // (a cell that still holds two edges is marked ambiguous below: its points take the literal path)
    out.resize(M);
    const double cw = range / M, slack = 0.02 * cw + guard;      // the kernel's float cell index may be off by a rounding at a cell boundary
    for (int c = 0; c < M; c++) {
        const double ctr = lo + (c + 0.5) * cw;
        size_t best = 0;
        for (size_t k = 1; k < edges.size(); k++) if (std::fabs(edges[k] - ctr) < std::fabs(edges[best] - ctr)) best = k;
        const double e = edges[best];
        const double h0 = std::min(lo + c * cw - slack, e) - guard, h1 = std::max(lo + (c + 1) * cw + slack, e) + guard;
        bool ambiguous = false;
        for (size_t k = 0; k < edges.size(); k++) if (k != best && edges[k] >= h0 && edges[k] <= h1) ambiguous = true;
        out[c].edge = ambiguous ? std::nanf("") : (float)e; out[c].idx = (int32_t)best;
    }
    return M;
}

icet_status ensure_thresholds(icet_ctx* c, int T, int P) {
    Workspace& w = c->w;
    if (w.thr && w.thr_T == T && w.thr_P == P) return ICET_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {
        // azimuth: diamond angle pa(theta) = y/(|x|+|y|) unfolded to [0,4]; polar: w = -cos(phi) in [-1,1]
        std::vector<double> et(T + 1), ep(P + 1);
        for (int k = 0; k <= T; k++) {
            const double th = 2 * M_PI * k / T, x = std::cos(th), y = std::sin(th), q = y / (std::fabs(x) + std::fabs(y));
            et[k] = (k == T) ? 4.0 : (k == 0 ? 0.0 : (x >= 0 ? (y >= 0 ? q : 4.0 + q) : 2.0 - q));
        }
        for (int k = 0; k <= P; k++) ep[k] = (k == 0) ? -1.0 : (k == P ? 1.0 : -std::cos(M_PI * k / P));
        // Guard bands: a few float ulps of the coordinate plus the ulp-level disagreement between the LUT's
        // true edges and the literal evaluation's float thresholds ("exact fast path").
        w.guard_t = 5e-6f * (float)c->tune.guard_scale; w.guard_p = 2.5e-6f * (float)c->tune.guard_scale;       // scale: experiments / tests only
        std::vector<HostCell> lt, lp;
        const double qp = c->tune.lut_polar_quantile;
        // Both tables live in the LDS of k_scan1_spherical and k_gn_accumulate blocks next to the voxel map.  A grid that is extremely fine in ONE direction
        // (thousands of azimuth bins on one ring) would want more cells than fit: the tables are then capped -- coarser cells hold two edges, are marked
        // ambiguous and send their points through the literal formulas: slower, same result -- and only a grid whose MINIMUM does not fit is refused.
        int cap_t = 1 << 16, cap_p = 1 << 16, Mt = 0, Mp = 0;
        for (;;) {
            Mt = build_lut(et, 0.0, 4.0, 0.0, w.guard_t, lt, cap_t); Mp = build_lut(ep, -1.0, 2.0, qp, w.guard_p, lp, cap_p);
            const size_t need_acc = acc_fixed_lds_bytes(T, P, Mt, Mp, true) + 32 * acc_row_lds_bytes();
            const size_t need_scan1 = (size_t)(Mt + Mp + 2) * sizeof(HostCell) + 4096;      // + the kernel's static tables (init_keyframe_kernels)
            if (std::max(need_acc, need_scan1) <= (size_t)c->max_lds) break;
            if (Mt <= 64 && Mp <= 64) { c->err = "grid too fine for this device's LDS (voxel map + look-up tables of one block)"; return ICET_ERR_UNSUPPORTED; }
            if (Mt >= Mp) cap_t = Mt / 2; else cap_p = Mp / 2;
        }
        // one spare cell per table: pa == 4 / w == 1 index cell M (it names the last edge, so the point goes to the literal path)
        for (HostCell& c : lp) c.idx *= T;                                 // polar cells carry the map row offset T * edge index
        std::vector<HostCell> all(lt); all.push_back(HostCell{4.0f, T}); all.insert(all.end(), lp.begin(), lp.end()); all.push_back(HostCell{1.0f, P * T});
        if (w.lut) { HIPCHK(c, hipFree(w.lut)); w.lut = nullptr; }
        HIPCHK(c, hipMalloc(&w.lut, all.size() * sizeof(HostCell)));
        HIPCHK(c, hipMemcpy(w.lut, all.data(), all.size() * sizeof(HostCell), hipMemcpyHostToDevice));
        w.lut_Mt = Mt; w.lut_Mp = Mp;
        // the same edges as floats, for the keep masks of the point pass (margins of 1e-2 rad: the float rounding of an edge is far inside their slack)
        std::vector<float> ef; for (double e : et) ef.push_back((float)e); for (double e : ep) ef.push_back((float)e);
        HIPCHK(c, dev_realloc(w.edges, ef.size()));
        HIPCHK(c, hipMemcpy(w.edges, ef.data(), ef.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    std::vector<float> h((size_t)T + P + 2);
    build_thresholds(T, 2 * M_PI, h.data());
    build_thresholds(P, M_PI, h.data() + T + 1);
    HIPCHK(c, dev_realloc(w.thr, h.size()));
    HIPCHK(c, hipMemcpy(w.thr, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    // the voxel of an exact-zero row, by the sign bits of (y, x): r = 0, phi = acos(NaN) -> 1000, theta = atan2(+-0, +-0) (src/utils.cpp:103-116) binned as
    // sortSphericalCoordinates bins them (src/icet.cpp:545-546) -- the formulas of theta_cr / phi_cr / voxel_of (icet_device_common.h) in host arithmetic
    for (int k = 0; k < 4; k++) {
        float th = (float)std::atan2((k & 2) ? -0.0 : 0.0, (k & 1) ? -0.0 : 0.0);
        if (th < 0.0f) th = (float)((double)th + 2.0 * M_PI);
        const float ph = 1000.0f;
        const int bt = static_cast<int>(((double)th / (2.0 * M_PI)) * (double)T) % T, bp = static_cast<int>(((double)ph / M_PI) * (double)P) % P;
        w.zero_voxel[k] = T * bp + bt;
    }
    w.thr_T = T; w.thr_P = P;
    return ICET_OK;
}

icet_status ensure_out(icet_ctx* c, int32_t n_pairs) {
    if (n_pairs <= c->cap_out_pairs) return ICET_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, dev_realloc(c->d_out, (size_t)n_pairs * 48));
    HIPCHK(c, dev_realloc(c->d_x0, (size_t)n_pairs * 6));
    if (c->h_out) { HIPCHK(c, hipHostFree(c->h_out)); c->h_out = nullptr; }
    if (c->h_x0) { HIPCHK(c, hipHostFree(c->h_x0)); c->h_x0 = nullptr; }
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_out), sizeof(float) * 48 * (size_t)n_pairs));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_x0), sizeof(float) * 6 * (size_t)n_pairs));
    c->cap_out_pairs = n_pairs;
    return ICET_OK;
}

// The copy stream of the host-pointer entry points and its event (created on first use).
// Uploads go through the runtime's own pageable path, one hipMemcpy2DAsync per scan: measured on the GPU box (profiles/r04_stage_probe.txt,
// scripts/hip/stage_probe.hip) it moves a 1.45 MB scan in 35 us (41 GB/s), as fast as from pinned memory; a ring of pinned chunks filled by
// copy threads -- built first -- was 3-4x SLOWER (every chunk's DMA command costs ~10 us, and this host copies 50 GB/s on one thread).
// What made the round-3 constructor path slow were the thirteen D2H copies into pageable arrays and the host loop for `points2`, not the upload.
icet_status ensure_host_path(icet_ctx* c) {
    if (!c->st_copy) HIPCHK(c, hipStreamCreateWithFlags(&c->st_copy, hipStreamNonBlocking));
    for (hipEvent_t* e : {&c->ev_s2, &c->ev_kf, &c->ev_kfd, &c->ev_prev, &c->ev_pts2})
        if (!*e) HIPCHK(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    return ICET_OK;
}

// Column-major N x 3 host scan (leading dimension ld) -> staging buffer (leading dimension l) on `st`.
hipError_t upload_scan(float* dst, int64_t l, const float* src, int64_t n, int64_t ld, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    return hipMemcpy2DAsync(dst, l * sizeof(float), src, ld * sizeof(float), n * sizeof(float), 3, hipMemcpyHostToDevice, st);
}

// (p + t) * R over a column-major scan (src/icet.cpp:375-378) with the device's own t and R (the transform record of write_xf): the host
// half of `points2`.  Plain float arithmetic; the AVX2 + FMA build of the same loop is taken when the CPU has it.
#define ICET_POINTS2_BODY                                                                                                     \
    const float tx = xf[0], ty = xf[1], tz = xf[2];                                                                           \
    const float R00 = xf[3], R01 = xf[4], R02 = xf[5], R10 = xf[6], R11 = xf[7], R12 = xf[8], R20 = xf[9], R21 = xf[10], R22 = xf[11]; \
    const float* __restrict__ px = s; const float* __restrict__ py = s + ld; const float* __restrict__ pz = s + 2 * ld;      \
    float* __restrict__ ox = out; float* __restrict__ oy = out + n; float* __restrict__ oz = out + 2 * n;                    \
    for (int64_t i = 0; i < n; i++) {                                                                                         \
        const float a = px[i] + tx, b = py[i] + ty, c = pz[i] + tz;                                                           \
        ox[i] = a * R00 + b * R10 + c * R20; oy[i] = a * R01 + b * R11 + c * R21; oz[i] = a * R02 + b * R12 + c * R22;       \
    }
#if defined(__x86_64__)
__attribute__((target("avx2,fma"))) void host_points2_avx2(const float* xf, const float* s, int64_t ld, int64_t n, float* out) { ICET_POINTS2_BODY }
#endif
void host_points2(const float* xf, const float* s, int64_t ld, int64_t n, float* out) {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) { host_points2_avx2(xf, s, ld, n, out); return; }
#endif
    ICET_POINTS2_BODY
}
#undef ICET_POINTS2_BODY

// Word offsets (4-byte words, every table 16-byte aligned) of the single-pair result block, which exists twice with one layout: in HBM
// (d_pack) and in pinned host memory (h_pack).  [0, small_end): the 48 result floats and the per-iteration 6 / 36 / 6-float tables, written
// in HBM and copied in ONE small D2H behind the loop (written straight into pinned memory by k_gn_solve they cost every iteration a PCIe
// round trip at its end: measured on the map maker, +6 us per iteration); xf_last: the transform record of the last iteration (for
// `points2`), written ONCE, straight into the pinned copy; [bounds, kf_end): the keyframe tables, final once the keyframe is built, copied to
// the host on the copy stream while the loop runs; then the rarely requested integer tables (copied at the end).
struct AuxLayout { size_t out, xf_last, x_hist, htwh, htwdz, cond, small_end, bounds, has_fit, mu1, sigma1, evecs1, l_diag, test_points, kf_end, n1_raw, n2_raw, n2_in, ints_end, total; };
AuxLayout aux_layout(int V, int runlen) {
    AuxLayout L{}; size_t o = 0;
    auto take = [&o](size_t n) { const size_t at = o; o += (n + 3) & ~(size_t)3; return at; };
    const size_t rl = runlen > 0 ? runlen : 1, v = (size_t)V;
    L.out = take(48); L.x_hist = take(rl * 6); L.htwh = take(rl * 36); L.htwdz = take(rl * 6); L.cond = take(rl * 8); L.small_end = o; L.xf_last = take(48);
    L.bounds = take(v * 6); L.has_fit = take(v); L.mu1 = take(v * 3); L.sigma1 = take(v * 9); L.evecs1 = take(v * 9); L.l_diag = take(v * 3);
    L.test_points = take(v * 18); L.kf_end = o; L.n1_raw = take(v); L.n2_raw = take(rl * v); L.n2_in = take(rl * v); L.ints_end = o;
    L.total = o;
    return L;
}

void free_aux(icet_ctx* c) {
    if (c->d_pack) { (void)hipFree(c->d_pack); c->d_pack = nullptr; }
    if (c->h_pack) { (void)hipHostFree(c->h_pack); c->h_pack = nullptr; }
    c->cap_pack = 0;
    c->aux_dev = AuxDev{};
    c->aux_V = 0; c->aux_runlen = 0;
}

// Device and pinned host copies of the result block for (V, runlen); c->aux_dev points into the device block.
icet_status ensure_pack(icet_ctx* c, int V, int runlen) {
    const AuxLayout L = aux_layout(V, runlen);
    if (L.total > c->cap_pack) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        free_aux(c);
        HIPCHK(c, dev_realloc(c->d_pack, L.total));
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_pack), L.total * sizeof(uint32_t)));
        c->cap_pack = L.total;
    }
    if (c->aux_V != V || c->aux_runlen != runlen) {
        AuxDev& a = c->aux_dev; uint32_t* d = c->d_pack;
        auto F = [d](size_t at) { return reinterpret_cast<float*>(d + at); };
        auto I = [d](size_t at) { return reinterpret_cast<int32_t*>(d + at); };
        float* hs = reinterpret_cast<float*>(c->h_pack);                     // pinned host memory is device-accessible under the same address
        a.bounds = F(L.bounds); a.n1_raw = I(L.n1_raw); a.has_fit = I(L.has_fit); a.mu1 = F(L.mu1); a.sigma1 = F(L.sigma1); a.evecs1 = F(L.evecs1);
        a.l_diag = F(L.l_diag); a.x_hist = F(L.x_hist); a.htwh = F(L.htwh); a.htwdz = F(L.htwdz); a.cond = F(L.cond); a.n2_raw = I(L.n2_raw); a.n2_in = I(L.n2_in);
        a.test_points = F(L.test_points); a.xf_last = hs + L.xf_last;
        c->aux_V = V; c->aux_runlen = runlen;
    }
    return ICET_OK;
}

// Enqueue the whole solve for descriptors already sitting in c->h_desc[0..n_pairs).
// The solve in two halves, so that a sequential caller can build the keyframe of a scan while the previous pair is still iterating
// (icet_keyframe_device / icet_register_device): enqueue_keyframe = ICET::fitScan1 (src/icet.cpp:68-107) for the scan-1 halves of the
// descriptors in c->h_desc[0..n_pairs), enqueue_loop = prepScan2 + runlen x fitScan2 (:254-277, :372-436) for their scan-2 halves
// against the tables the keyframe left in the workspace.
LaunchCfg make_cfg(icet_ctx* c, const icet_params* p, int32_t n_pairs) {
    LaunchCfg cfg{};
    cfg.T = p->bins_theta; cfg.P = p->bins_phi; cfg.V = cfg.T * cfg.P; cfg.n = p->n; cfg.runlen = p->runlen;
    cfg.thresh = p->thresh; cfg.buff = p->buff; cfg.n_pairs = n_pairs;
    cfg.half_gap = (p->flags & ICET_FLAG_HALF_GAP_BOUNDS) ? 1 : 0;
    cfg.true_sort = (p->flags & (ICET_FLAG_TRUE_SORT | ICET_FLAG_HALF_GAP_BOUNDS)) ? 1 : 0;
    cfg.reject_moving = (p->flags & ICET_FLAG_REJECT_MOVING) ? 1 : 0;
    cfg.rt2 = (p->flags & ICET_FLAG_ROUNDTRIP_SCAN2) ? 1 : 0;
    cfg.ref_w = (p->flags & ICET_FLAG_DOUBLE_W) ? 0 : 1;
    cfg.lds_slots = c->tune.lds_slots; cfg.acc_min_pts_per_thread = c->tune.acc_pts; cfg.acc_target_blocks = c->tune.acc_blocks;
    cfg.force_exact = c->tune.force_exact; cfg.use_library_sort = c->tune.library_sort; cfg.kf_pts_per_thread = c->tune.kf_pts; cfg.rs_cap = c->tune.rs_cap; cfg.rs_max_cell = c->tune.rs_max_cell; cfg.exec_bits_lds = c->tune.exec_bits_lds; cfg.exec_pairwise = c->tune.exec_pairwise; cfg.fuse_solve = c->tune.fuse_solve != 0 ? 1 : 0; cfg.lds_rank = (c->tune.lds_rank != 0 && c->lds_rank_ok) ? 1 : 0;
    cfg.gn_cond_bound2 = (float)(c->tune.gn_cond_bound * c->tune.gn_cond_bound);
    cfg.pair_user = (c->perm_active && c->perm_pairs == n_pairs) ? c->w.seg_off + n_pairs + 1 : nullptr;
    cfg.done_flag = c->done_flag;
    // the keep list of the point pass (KeepState, icet_internal.h): throughput batches only -- a small batch's point pass is a few microseconds of launch floor --,
    // never with the scan-2 round trip (a kernel of its own) nor when fewer than two passes could walk a list; same bits either way
    cfg.keep_from = c->tune.keep_from < 0 ? 0 : c->tune.keep_from; cfg.keep_bt = (float)c->tune.keep_budget_t; cfg.keep_br = (float)c->tune.keep_budget_r; cfg.keep_check_scale = (float)c->tune.keep_check_scale;
    cfg.keep = (c->tune.keep != 0 && n_pairs >= 32 && !cfg.rt2 && p->runlen >= cfg.keep_from + 3 && c->w.lut_Mt > 0 &&
                acc_fixed_lds_bytes(cfg.T, cfg.P, c->w.lut_Mt, c->w.lut_Mp, false, true) + 64 * acc_row_lds_bytes() <= (size_t)c->max_lds) ? 1 : 0;
    if (cfg.kf_pts_per_thread > kKfMaxPtsPerThread) cfg.kf_pts_per_thread = kKfMaxPtsPerThread;      // k_bin_scatter: a tile is at most 4 waves x that many rounds x 64 positions
    if (cfg.kf_pts_per_thread < 1) cfg.kf_pts_per_thread = 1;
    cfg.stage_event = c->stage_at ? c->ev_stage : nullptr; cfg.stage_at = c->stage_at;
    cfg.vec4_ok = 1;
    int64_t tot = 0, tot2 = 0; int mx1 = 0, mx2 = 0;
    for (int k = 0; k < n_pairs; k++) {
        if ((reinterpret_cast<uintptr_t>(c->h_desc[k].s2) & 15u) || (c->h_desc[k].ld2 & 3)) cfg.vec4_ok = 0;
        c->h_seg[k] = (int32_t)tot; c->h_desc[k].off1 = (int32_t)tot; tot += c->h_desc[k].n1;
        c->h_desc[k].off2 = (int32_t)tot2; tot2 += c->h_desc[k].n2;
        if (c->h_desc[k].n1 > mx1) mx1 = c->h_desc[k].n1;
        if (c->h_desc[k].n2 > mx2) mx2 = c->h_desc[k].n2;
    }
    c->h_seg[n_pairs] = (int32_t)tot;
    cfg.total_n1 = tot; cfg.max_n1 = mx1; cfg.max_n2 = mx2;
    // A small batch of ordinary scans leaves most CUs without a keyframe tile at the full tile size (one 120 k-row scan: 59 tiles on 256 CUs): half-size tiles
    // then (measured, single pair: keyframe 0.170 -> 0.160 ms; a 485 k-row scan keeps the full size: 0.29 vs 0.31).  Same bits either way.
    if (cfg.kf_pts_per_thread == kKfMaxPtsPerThread && kKfMaxPtsPerThread >= 8 && (int64_t)n_pairs * ((mx1 + 256 * kKfMaxPtsPerThread - 1) / (256 * kKfMaxPtsPerThread)) < 128)
        cfg.kf_pts_per_thread = kKfMaxPtsPerThread / 2;
    cfg.kf_chunks = (mx1 + 256 * cfg.kf_pts_per_thread - 1) / (256 * cfg.kf_pts_per_thread);
    if (cfg.kf_chunks < 1) cfg.kf_chunks = 1;
    return cfg;
}

icet_status upload_desc(icet_ctx* c, int32_t n_pairs, bool by_next_kernel = false) {
    Workspace& w = c->w;
    if (by_next_kernel) {
        // (the caller's next launch, k_init_state, copies the pinned words itself)
    } else if (n_pairs <= kUploadDescMaxPairs) {
        HIPCHK(c, launch_upload_desc(w, c->h_desc, c->h_seg, n_pairs, c->stream));       // small batch: a kernel reads the pinned words (no copy command, no memcpy node)
    } else {
        HIPCHK(c, hipMemcpyAsync(w.desc, c->h_desc, sizeof(PairDesc) * n_pairs, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(w.seg_off, c->h_seg, sizeof(int32_t) * (n_pairs + 1 + (c->perm_active ? n_pairs : 0)), hipMemcpyHostToDevice, c->stream));
    }
    if (!c->ev_desc) HIPCHK(c, hipEventCreateWithFlags(&c->ev_desc, hipEventDisableTiming));
    if (!c->capturing) { HIPCHK(c, hipEventRecord(c->ev_desc, c->stream)); c->desc_in_flight = true; }     // (a captured event cannot be waited for on the host: the replay path orders the staging itself)
    return ICET_OK;
}

icet_status enqueue_keyframe(icet_ctx* c, const icet_params* p, int32_t n_pairs, const AuxDev* aux, const int32_t* d_counts1 = nullptr) {
    Workspace& w = c->w;
    { icet_status ts = ensure_thresholds(c, p->bins_theta, p->bins_phi); if (ts != ICET_OK) return ts; }
    const LaunchCfg cfg = make_cfg(c, p, n_pairs);
    {
        const size_t need = (size_t)n_pairs * cfg.kf_chunks * (cfg.V > kRankSortMaxBuckets ? cfg.V : kRankSortMaxBuckets);
        if (need > w.cap_counts) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, dev_realloc(w.counts, need)); HIPCHK(c, dev_realloc(w.tile_base, need));
            w.cap_counts = need;
        }
        const size_t need_vr = (size_t)n_pairs * cfg.kf_chunks * 2;        // (its own capacity: tiles per voxel count vary with the grid)
        if (need_vr > w.cap_tile_vr) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, dev_realloc(w.tile_vr, need_vr));
            w.cap_tile_vr = need_vr;
        }
    }
    // a small batch: k_rs_splitters, the keyframe's first kernel, takes the descriptors from the pinned staging itself
    const bool first_uploads = n_pairs <= kUploadDescMaxPairs && !cfg.use_library_sort;
    { icet_status us = upload_desc(c, n_pairs, first_uploads); if (us != ICET_OK) return us; }
    if (!c->capturing) HIPCHK(c, hipEventRecord(c->ev_a, c->stream));
    LaunchCfg kcfg = cfg; if (first_uploads) { kcfg.h_desc_up = c->h_desc; kcfg.h_seg_up = c->h_seg; }
    HIPCHK(c, launch_keyframe(w, kcfg, aux, c->stream, d_counts1));      // (d_counts1: the descriptors hold upper bounds, the device knows the row counts: patched by the first kernel)
    if (first_uploads && !c->capturing) { HIPCHK(c, hipEventRecord(c->ev_desc, c->stream)); c->desc_in_flight = true; }      // (the staging has been read once these kernels have run)
    return ICET_OK;
}

icet_status enqueue_loop(icet_ctx* c, const icet_params* p, int32_t n_pairs, const float* d_x0, float* d_out, const AuxDev* aux, bool reupload, float* pts2_out = nullptr, hipEvent_t scan2_ready = nullptr, const int32_t* d_counts2 = nullptr) {
    const bool want_pts2 = aux && aux->xf_last && p->runlen > 0;      // pts2_out: the device computes `points2` (else only the transform snapshot + its event: the host does)
    Workspace& w = c->w;
    const LaunchCfg cfg = make_cfg(c, p, n_pairs);
    // the scan-2 halves arrived after the keyframe call.  A small batch without the scan-2 round trip: k_init_state, the first kernel below that reads a descriptor, uploads them
    const bool init_uploads = reupload && n_pairs <= kUploadDescMaxPairs && !cfg.rt2 && !cfg.keep;
    if (reupload) { icet_status us = upload_desc(c, n_pairs, init_uploads); if (us != ICET_OK) return us; }
    if (d_counts2 && cfg.rt2) HIPCHK(c, launch_patch_counts(w, cfg, nullptr, d_counts2, c->stream));      // (otherwise k_init_state, the loop's first kernel, patches them)
    while ((int)c->ev_acc.size() < 2 * p->runlen) { hipEvent_t e; HIPCHK(c, hipEventCreate(&e)); c->ev_acc.push_back(e); }
    Workspace wl = w;                                    // what the loop kernels see: with ICET_FLAG_ROUNDTRIP_SCAN2 their scan 2 is the round-tripped copy
    if (!c->capturing) HIPCHK(c, hipEventRecord(c->ev_b, c->stream));      // keyframe_ms ends here: the scan-2 pre-pass of ICET_FLAG_ROUNDTRIP_SCAN2 belongs to the loop
    if (cfg.rt2) {
        int64_t tot = 0;
        for (int k = 0; k < n_pairs; k++) tot += (c->h_desc[k].n2 + 63) / 64 * 64;
        if (tot > w.cap_rt2 || !w.desc_rt || n_pairs > c->h_cap_rt) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (tot > w.cap_rt2) { HIPCHK(c, dev_realloc(w.rt2, (size_t)3 * tot)); w.cap_rt2 = tot; }
            if (n_pairs > c->h_cap_rt || !w.desc_rt) {
                const int np = n_pairs > w.cap_pairs ? n_pairs : w.cap_pairs;
                HIPCHK(c, dev_realloc(w.desc_rt, np));
                if (c->h_desc_rt) { HIPCHK(c, hipHostFree(c->h_desc_rt)); c->h_desc_rt = nullptr; }
                HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_desc_rt), sizeof(PairDesc) * np));
                c->h_cap_rt = np;
            }
        } else if (c->desc_in_flight) { HIPCHK(c, hipEventSynchronize(c->ev_desc)); c->desc_in_flight = false; }
        int64_t o = 0;
        for (int k = 0; k < n_pairs; k++) {
            const int64_t l = (c->h_desc[k].n2 + 63) / 64 * 64;
            PairDesc dr = c->h_desc[k];
            dr.s2 = w.rt2 + 3 * o; dr.ld2 = (int32_t)l;
            c->h_desc_rt[k] = dr; o += l;
        }
        HIPCHK(c, hipMemcpyAsync(w.desc_rt, c->h_desc_rt, sizeof(PairDesc) * n_pairs, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_desc, c->stream)); c->desc_in_flight = true;
        if (scan2_ready) HIPCHK(c, hipStreamWaitEvent(c->stream, scan2_ready, 0));
        HIPCHK(c, launch_rt2_prepare(w, cfg, c->stream));
        wl = w; wl.desc = w.desc_rt;
    }
    LaunchCfg lcfg = cfg; if (cfg.rt2) lcfg.vec4_ok = 1;     // the copy is 64-float aligned whatever the caller's layout was
    if (cfg.keep) {                                          // masks, list and per-pair state of the keep list (grown here: only throughput batches use them)
        int64_t tot2 = 0; for (int k = 0; k < n_pairs; k++) tot2 += c->h_desc[k].n2;
        const size_t need_mask = (size_t)(tot2 >> 8) + (size_t)n_pairs + 2, need_list = (size_t)(tot2 >> 2) + (size_t)n_pairs + 4;
        if (need_mask > w.cap_keep_mask || need_list > w.cap_keep_list || n_pairs > w.cap_keep_pairs) {
            if (c->capturing) { c->err = "keep list: workspace growth during capture"; return ICET_ERR_HIP; }
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (need_mask > w.cap_keep_mask) { HIPCHK(c, dev_realloc(w.keep_mask, need_mask)); w.cap_keep_mask = need_mask; }
            if (need_list > w.cap_keep_list) { HIPCHK(c, dev_realloc(w.keep_list, need_list)); w.cap_keep_list = need_list; }
            if (n_pairs > w.cap_keep_pairs) { HIPCHK(c, dev_realloc(w.keep_state, (size_t)n_pairs)); HIPCHK(c, dev_realloc(w.keep_modes, 2 * (size_t)n_pairs)); w.cap_keep_pairs = n_pairs; }
            wl.keep_modes = w.keep_modes; wl.keep_mask = w.keep_mask; wl.keep_list = w.keep_list; wl.keep_state = w.keep_state; wl.cap_keep_mask = w.cap_keep_mask; wl.cap_keep_list = w.cap_keep_list; wl.cap_keep_pairs = w.cap_keep_pairs;
        }
    }
    HIPCHK(c, launch_init_state(w, cfg, d_x0, c->stream, want_pts2 ? aux->xf_last : nullptr, cfg.rt2 ? nullptr : d_counts2, init_uploads ? c->h_desc : nullptr, init_uploads ? c->h_seg : nullptr));
    if (init_uploads && !c->capturing) { HIPCHK(c, hipEventRecord(c->ev_desc, c->stream)); c->desc_in_flight = true; }      // (the staging is read by THIS kernel: the event of upload_desc, moved behind it)
    // `points2` (include/icet.h:80): scan 2 as the LAST fitScan2 transforms it (src/icet.cpp:375-378).  That transform is known as soon as the
    // solve of iteration runlen - 2 has run: k_gn_solve / k_init_state snapshot its record in aux->xf_last (pinned host memory) and ev_prev
    // marks the moment.  icet_solve_end then transforms scan 2 ON THE HOST while the last iteration still runs on the device (measured: a
    // device kernel + 1.45 MB D2H + the copy into the caller's pageable array cost 110 us behind the loop, the host pass ~25 us under it).
    // Only with ICET_FLAG_ROUNDTRIP_SCAN2, whose points2_OG exists on the device alone, the device computes it (copy stream).
    auto enqueue_points2 = [&]() -> icet_status {
        HIPCHK(c, hipEventRecord(c->ev_prev, c->stream));
        if (!pts2_out) return ICET_OK;
        HIPCHK(c, hipStreamWaitEvent(c->st_copy, c->ev_prev, 0));
        HIPCHK(c, launch_points2(wl, lcfg, aux->xf_last, c->d_pts2, c->st_copy));
        HIPCHK(c, hipMemcpyAsync(pts2_out, c->d_pts2, sizeof(float) * 3 * (size_t)c->h_desc[0].n2, hipMemcpyDeviceToHost, c->st_copy));
        HIPCHK(c, hipEventRecord(c->ev_pts2, c->st_copy));
        return ICET_OK;
    };
    if (scan2_ready) HIPCHK(c, hipStreamWaitEvent(c->stream, scan2_ready, 0));     // host-pointer entries: scan 2 was uploaded on the copy stream beside the keyframe build
    if (want_pts2 && p->runlen == 1) { const icet_status ps = enqueue_points2(); if (ps != ICET_OK) return ps; }
    const bool per_iter = (p->flags & ICET_FLAG_TIMING) != 0;
    for (int it = 0; it < p->runlen; it++) {
        if (per_iter) HIPCHK(c, hipEventRecord(c->ev_acc[2 * it], c->stream));
        // (per-iteration timing wants the two halves apart; otherwise a small batch runs the solve inside the point pass' launch)
        const FuseArgs fa{it, d_out, aux};
        bool fused = false;
        // keep list: from iteration keep_from on the point pass is the kernel that walks a pair's list or, for a pair without a valid one, its whole scan + keep masks;
        // the solve behind it builds / checks the lists (not behind the last pass: nothing follows)
        const int keep_pass = (lcfg.keep && it >= lcfg.keep_from) ? 1 : 0;
        HIPCHK(c, launch_gn_accumulate(wl, lcfg, c->stream, per_iter ? nullptr : &fa, &fused, keep_pass ? 1 + 2 * (it & 1) : 0));
        if (per_iter) HIPCHK(c, hipEventRecord(c->ev_acc[2 * it + 1], c->stream));
        if (!fused) HIPCHK(c, launch_gn_solve(wl, lcfg, it, d_out, aux, c->stream, keep_pass ? (it + 1 < p->runlen ? 1 : 2) : 0));
        if (want_pts2 && it == p->runlen - 2) { const icet_status ps = enqueue_points2(); if (ps != ICET_OK) return ps; }
    }
    if (!c->capturing) { HIPCHK(c, hipEventRecord(c->ev_c, c->stream)); c->timing_valid = true; c->last_iters = per_iter ? p->runlen : 0; }
    return ICET_OK;
}

// Enqueue the whole solve for descriptors already sitting in c->h_desc[0..n_pairs).
icet_status enqueue(icet_ctx* c, const icet_params* p, int32_t n_pairs, const float* d_x0, float* d_out, const AuxDev* aux) {
    c->kf_pairs = 0;                                                    // whatever keyframe a sequential caller had parked here is overwritten
    icet_status s = enqueue_keyframe(c, p, n_pairs, aux);
    if (s != ICET_OK) return s;
    return enqueue_loop(c, p, n_pairs, d_x0, d_out, aux, false);
}

// Staging buffers of the host-pointer entry points (3 x l floats per scan, leading dimension l = n rounded up to 64).
icet_status ensure_stage(icet_ctx* c, int64_t tot1, int64_t tot2) {
    if (3 * tot1 > c->cap_stage1 || 3 * tot2 > c->cap_stage2) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->st_copy) HIPCHK(c, hipStreamSynchronize(c->st_copy));
        if (3 * tot1 > c->cap_stage1) { HIPCHK(c, dev_realloc(c->d_stage1, (size_t)3 * tot1)); c->cap_stage1 = 3 * tot1; }
        if (3 * tot2 > c->cap_stage2) { HIPCHK(c, dev_realloc(c->d_stage2, (size_t)3 * tot2)); c->cap_stage2 = 3 * tot2; }
    }
    return ICET_OK;
}

icet_status write_runlen0(icet_ctx* c, int32_t n_pairs, const float* d_x0, float* d_out) {
    // runlen == 0: the reference constructor returns X = X0, pred_stds = 0 (src/icet.cpp:36-37,47).
    HIPCHK(c, hipMemsetAsync(d_out, 0, sizeof(float) * 48 * (size_t)n_pairs, c->stream));
    if (d_x0) HIPCHK(c, hipMemcpy2DAsync(d_out, 48 * sizeof(float), d_x0, 6 * sizeof(float), 6 * sizeof(float), n_pairs, hipMemcpyDeviceToDevice, c->stream));
    return ICET_OK;
}

}  // namespace

extern "C" {

const char* icet_version(void) { return "icet_hip 0.1 (gfx950)"; }

const char* icet_last_error(const icet_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

icet_status icet_create(icet_ctx** out, int device_id, void* hip_stream) {
    if (!out) return ICET_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ICET_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= ndev) return ICET_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return ICET_ERR_NO_DEVICE;
    icet_ctx* c = new (std::nothrow) icet_ctx();
    if (!c) return ICET_ERR_NOMEM;
    c->device = device_id;
    if (hip_stream) { c->stream = reinterpret_cast<hipStream_t>(hip_stream); c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return ICET_ERR_HIP; }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev_a) != hipSuccess || hipEventCreate(&c->ev_b) != hipSuccess || hipEventCreate(&c->ev_c) != hipSuccess) { delete c; return ICET_ERR_HIP; }
    // per context, with its device current: raise the dynamic-LDS limits of the kernels that need it (no process-global flags)
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess && lds > 0) c->max_lds = lds;
    if (init_keyframe_kernels() != hipSuccess || init_accumulate_kernels() != hipSuccess || init_rank_sort_kernels() != hipSuccess) {
        (void)hipGetLastError();
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c; return ICET_ERR_HIP;       // no usable kernel image for this device: fail loudly, there is no fallback
    }
    {   // which form of the stable multi-splits this device gets (Tuning::lds_rank): ~0.1 ms once per context
        int32_t* d_t = nullptr; int ok = 0;
        if (hipMalloc(reinterpret_cast<void**>(&d_t), sizeof(int32_t)) == hipSuccess) { if (lds_rank_selftest(d_t, c->stream, &ok) != hipSuccess) { ok = 0; (void)hipGetLastError(); } (void)hipFree(d_t); }
        c->lds_rank_ok = ok;
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_sync_word), sizeof(int32_t), hipHostMallocCoherent) == hipSuccess) *c->h_sync_word = 0;      // (without it icet_sync synchronises the stream)
    else { c->h_sync_word = nullptr; (void)hipGetLastError(); }
    *out = c;
    return ICET_OK;
}

icet_status icet_destroy(icet_ctx* c) {
    if (!c) return ICET_ERR_BAD_ARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    Workspace& w = c->w;
    void* ps[] = {w.key64A, w.key64B, w.bin16, w.execbits, w.binpos, w.bkt, w.splitters, w.n_buckets, w.bucket_start, w.counts, w.tile_base, w.desc, w.seg_off, w.r1, w.cart1, w.keyA, w.keyB, w.valA, w.valB, w.pred, w.src,
                  w.desc_rt, w.rt2, w.gn_part, w.bin_count, w.bin_start, w.hotD, w.fitD, w.activeD, w.midD, w.hotS, w.fitS, w.slot_of_voxel, w.n_slots, w.near_over, w.near_over_count, w.acc, w.xf, w.X, w.flags, w.zero_rows, w.vrange, w.tile_vr,
                  w.sort_tmp, w.fit_items, w.fit_n_items, w.live_bins, w.n_live, w.thr, w.lut, w.keep_mask, w.keep_list, w.keep_state, w.keep_modes, w.edges, c->d_stage1, c->d_stage2, c->d_out, c->d_x0};
    for (void* p : ps) if (p) (void)hipFree(p);
    free_aux(c);
    if (c->st_copy) (void)hipStreamSynchronize(c->st_copy);
    if (c->h_pts2) (void)hipHostFree(c->h_pts2);
    if (c->d_pts2) (void)hipFree(c->d_pts2);
    for (void* q : {(void*)c->d_sph1, (void*)c->d_idx1, (void*)c->d_sph2, (void*)c->d_vox2}) if (q) (void)hipFree(q);
    if (c->h_x0) (void)hipHostFree(c->h_x0);
    if (c->h_sync_word) (void)hipHostFree(c->h_sync_word);
    for (hipEvent_t e : {c->ev_s2, c->ev_kf, c->ev_kfd, c->ev_prev, c->ev_pts2}) if (e) (void)hipEventDestroy(e);
    if (c->st_copy) (void)hipStreamDestroy(c->st_copy);
    if (c->h_desc) (void)hipHostFree(c->h_desc);
    if (c->h_seg) (void)hipHostFree(c->h_seg);
    if (c->h_desc_rt) (void)hipHostFree(c->h_desc_rt);
    if (c->h_out) (void)hipHostFree(c->h_out);
    for (hipEvent_t e : c->ev_acc) (void)hipEventDestroy(e);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->ev_c) (void)hipEventDestroy(c->ev_c);
    for (icet_ctx* h : c->helpers) (void)icet_destroy(h);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_desc) (void)hipEventDestroy(c->ev_desc);
    for (icet_ctx::GraphSlot* g : {&c->g_solve, &c->g_keyframe, &c->g_loop}) if (g->have_graph) { (void)hipGraphExecDestroy(g->exec); (void)hipGraphDestroy(g->graph); }
    if (c->ev_graph) (void)hipEventDestroy(c->ev_graph);
    if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return ICET_OK;
}

icet_status icet_sync(icet_ctx* c) {
    if (!c) return ICET_ERR_BAD_ARG;
    if (c->armed_calls == 1 && c->h_sync_word) {
        // exactly one small solve is in flight and its last kernel raises h_sync_word behind its results: watch the word (hipStreamSynchronize answers several
        // microseconds after the queue has drained), asking the stream only now and then so that a failed launch still ends the wait
        volatile int32_t* w = c->h_sync_word;
        for (long spins = 1; *w == 0; spins++)
            if ((spins & 8191) == 0 && hipStreamQuery(c->stream) != hipErrorNotReady) break;
        (void)hipGetLastError();
        if (*w != 0) { __atomic_thread_fence(__ATOMIC_ACQUIRE); c->armed_calls = 0; return ICET_OK; }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->armed_calls = 0;
    return ICET_OK;
}

static int batch_parts(const icet_ctx* c, const icet_params* p, int32_t n_pairs);
static icet_status ensure_helpers(icet_ctx* c, int parts);

icet_status icet_reserve(icet_ctx* c, const icet_params* p, int32_t n_pairs, int64_t total_n1, int64_t total_n2) {
    if (!c || !params_ok(p) || n_pairs < 0 || total_n1 < 0 || total_n2 < 0) return ICET_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    // (a parked keyframe survives a reservation that fits the current capacity: ensure_workspace un-parks it exactly when its tables move)
    const int parts = batch_parts(c, p, n_pairs);
    if (parts > 1) {            // the batch will be solved in parts (icet_solve_batch_device); 12.5 % headroom for uneven scans
        icet_status hs = ensure_helpers(c, parts);
        if (hs != ICET_OK) return hs;
        const int np = (n_pairs + parts - 1) / parts;
        const int64_t n1 = total_n1 / parts + total_n1 / (8 * parts) + 1;
        const int64_t n2 = total_n2 / parts + total_n2 / (8 * parts) + 1;
        for (icet_ctx* h : c->helpers) { hs = ensure_workspace(h, p, np, n1, n2); if (hs != ICET_OK) { c->err = h->err; return hs; } }
        hs = ensure_workspace(c, p, np, n1, n2);
        if (hs != ICET_OK) return hs;
        return ensure_out(c, n_pairs);
    }
    icet_status s = ensure_workspace(c, p, n_pairs, total_n1, total_n2);
    if (s != ICET_OK) return s;
    return ensure_out(c, n_pairs);
}

// How many parts a device batch is cut into (see icet_ctx::helpers).  Timed calls stay in one part so that every
// kernel is measured alone on the device.
static int batch_parts(const icet_ctx* c, const icet_params* p, int32_t n_pairs) {
    if (p->flags & ICET_FLAG_TIMING) return 1;
    // One part.  Cutting the batch into staggered parts on separate streams paid in round 1 (1 part 84.7 k pairs/s, 2 parts 89.2 k) and has
    // been a wash since the keyframe and loop kernels were tightened: round 3, 256 pairs, 1 part 100.7 k, 2 parts 100.2 k, 3 parts 96.7 k,
    // 4 parts 82.1 k (scripts/exp_parts.sh) -- every phase fills the device on its own.  The mechanism stays behind `batch_parts`.
    int parts = 1;
    if (c->tune.batch_parts > 0) parts = c->tune.batch_parts > 8 ? 8 : c->tune.batch_parts;
    if (parts > n_pairs) parts = n_pairs > 0 ? n_pairs : 1;
    return parts;
}

static icet_status ensure_helpers(icet_ctx* c, int parts) {
    while ((int)c->helpers.size() < parts - 1) {
        icet_ctx* h = nullptr;
        icet_status s = icet_create(&h, c->device, nullptr);
        if (s != ICET_OK) { c->err = "helper context: cannot create"; return s; }
        if (hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) { icet_destroy(h); c->err = "helper context: event"; return ICET_ERR_HIP; }
        c->helpers.push_back(h);
    }
    if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    return ICET_OK;
}

static icet_status solve_device_part(icet_ctx* c, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan1, const icet_dev_scan* scan2,
                                     const float* d_x0, float* d_out, int64_t tot1);

icet_status icet_solve_batch_device(icet_ctx* c, const icet_params* p, int32_t n_pairs,
                                    const icet_dev_scan* scan1, const icet_dev_scan* scan2,
                                    const float* d_x0, float* d_out) {
    if (!c) return ICET_ERR_BAD_ARG;
    if (!params_ok(p) || n_pairs < 0 || (n_pairs > 0 && (!scan1 || !scan2 || !d_out))) { c->err = "bad argument"; return ICET_ERR_BAD_ARG; }
    if (n_pairs == 0) return ICET_OK;
    for (int k = 0; k < n_pairs; k++) {
        const icet_dev_scan &a = scan1[k], &b = scan2[k];
        if (a.n < 0 || b.n < 0 || a.ld < a.n || b.ld < b.n || (a.n > 0 && !a.ptr) || (b.n > 0 && !b.ptr) ||
            a.ld >= ((int64_t)1 << 30) || b.ld >= ((int64_t)1 << 30)) { c->err = "bad scan descriptor"; return ICET_ERR_BAD_ARG; }
    }
    HIPCHK(c, hipSetDevice(c->device));
    const int parts = (p->runlen == 0) ? 1 : batch_parts(c, p, n_pairs);
    auto range_tot = [&](int b, int e) { int64_t t = 0; for (int k = b; k < e; k++) t += scan1[k].n; return t; };
    if (parts == 1) return solve_device_part(c, p, n_pairs, scan1, scan2, d_x0, d_out, range_tot(0, n_pairs));
    c->armed_calls = 2;
    { icet_status s = ensure_helpers(c, parts); if (s != ICET_OK) return s; }
    // fork: helpers start after whatever the caller queued on this context's stream (e.g. the writes of the scans)
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    // Stagger: part i+1 starts once part i has finished the first keyframe stage, so the parts do not move through the
    // same phase in lock step (measured on 256 pairs with the kernels of that day: 1 part 70.6k pairs/s; 3 parts in lock
    // step 73.2k; staggered 76.8k).
    const int stage = c->tune.batch_stage;
    icet_status status = ICET_OK;
    int started = 0;                        // helpers whose stream has work queued
    for (int i = 0; i < parts && status == ICET_OK; i++) {
        const int b = (int)((int64_t)n_pairs * i / parts), e = (int)((int64_t)n_pairs * (i + 1) / parts);
        icet_ctx* h = (i == 0) ? c : c->helpers[i - 1];
        if (h != c) { const bool relut = h->tune.guard_scale != c->tune.guard_scale || h->tune.lut_polar_quantile != c->tune.lut_polar_quantile; h->tune = c->tune; if (relut) h->w.thr_T = 0; }
        hipError_t he = hipSuccess;
        if (stage && !h->ev_stage) he = hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming);
        h->stage_at = (i + 1 < parts) ? stage : 0;
        if (he == hipSuccess && i > 0) {
            he = hipStreamWaitEvent(h->stream, c->ev_fork, 0);
            if (he == hipSuccess && stage) { icet_ctx* prev = (i == 1) ? c : c->helpers[i - 2]; he = hipStreamWaitEvent(h->stream, prev->ev_stage, 0); }
        }
        if (he != hipSuccess) { c->err = std::string("batch part fork: ") + hipGetErrorString(he); status = ICET_ERR_HIP; break; }
        if (i > 0) started = i;
        status = solve_device_part(h, p, e - b, scan1 + b, scan2 + b, d_x0 ? d_x0 + 6 * (size_t)b : nullptr, d_out + 48 * (size_t)b, range_tot(b, e));
        if (status != ICET_OK && h != c) c->err = h->err;
    }
    // join -- ALSO on failure: every helper that was started may still be reading the caller's scans / x0 and writing d_out, so
    // the caller's stream must not pass this call (and the caller must not free those buffers) before they have drained
    for (int i = 1; i <= started; i++) {
        icet_ctx* h = c->helpers[i - 1];
        if (hipEventRecord(h->ev_join, h->stream) != hipSuccess || hipStreamWaitEvent(c->stream, h->ev_join, 0) != hipSuccess) {
            (void)hipStreamSynchronize(h->stream);          // last resort: block the host until the helper is idle
            if (status == ICET_OK) { c->err = "batch part join failed"; status = ICET_ERR_HIP; }
        }
    }
    return status;
}

extern "C++" {
// ---- hipGraph replay of small device batches ------------------------------------------------------------------------------------
// The kernels read the scans' addresses and sizes from the descriptor table (re-uploaded from pinned memory by a memcpy node of the graph on
// every replay, and patched with device-side row counts where the caller has them); what the launches themselves depend on is the LaunchCfg
// (grids, LDS sizes, point counts passed by value) and the workspace pointers.  A call whose key equals the previous call's is captured;
// later calls with that key replay: one hipGraphLaunch instead of 15 - 35 launches on the host.
static icet_ctx::GraphKey graph_key_of(icet_ctx* c, const icet_params* p, int32_t n_pairs, const void* a0, const void* a1, const void* a2, const void* a3) {
    icet_ctx::GraphKey key{};
    const LaunchCfg k = make_cfg(c, p, n_pairs);
    auto bits = [](float f) { int32_t i; std::memcpy(&i, &f, 4); return (int64_t)i; };
    const int64_t vals[] = {k.T, k.P, k.V, k.n, k.runlen, bits(k.thresh), bits(k.buff), k.n_pairs, k.max_n1, k.max_n2, k.total_n1, k.lds_slots, k.acc_min_pts_per_thread,
                            k.acc_target_blocks, k.kf_chunks, k.kf_pts_per_thread, k.use_library_sort, k.vec4_ok, k.true_sort, k.force_exact, k.rs_cap, k.rs_max_cell,
                            k.exec_bits_lds, k.exec_pairwise, k.lds_rank, k.reject_moving, k.half_gap, k.rt2 + 2 * k.ref_w, p->flags, (int64_t)(intptr_t)a0, (int64_t)(intptr_t)a1, (int64_t)(intptr_t)a2, (int64_t)(intptr_t)a3,
                            (int64_t)(intptr_t)c->w.desc, (int64_t)(intptr_t)c->w.thr, (int64_t)(intptr_t)c->w.lut, (int64_t)(intptr_t)c->w.r1, (int64_t)(intptr_t)c->w.counts,
                            (int64_t)(intptr_t)c->w.near_over, (int64_t)(intptr_t)c->w.acc, (int64_t)(intptr_t)c->w.fit_items, (int64_t)(intptr_t)c->w.sort_tmp, (int64_t)(intptr_t)c->w.tile_vr,
                            c->prologue ? c->prologue_key : 0, (int64_t)(intptr_t)c->done_flag};
    static_assert(sizeof(vals) == sizeof(key.v), "GraphKey size");
    std::memcpy(key.v, vals, sizeof(vals));
    return key;
}
static bool graph_eligible(const icet_ctx* c, const icet_params* p, int32_t n_pairs) {
#ifdef ICET_DIAG_ENV      /* experiment builds only (make EXTRA=-DICET_DIAG_ENV): the shipped library never reads the environment -- use icet_set_option("graph", 0) */
    static const bool env_off = getenv("ICET_NO_GRAPH") != nullptr;
#else
    constexpr bool env_off = false;
#endif
    return !env_off && c->graph_mode && n_pairs <= 8 && !(p->flags & (ICET_FLAG_TIMING | ICET_FLAG_ROUNDTRIP_SCAN2)) && !c->stage_at;
}
// Runs `enq` (which enqueues on c->stream) eagerly, or captured into `slot` and replayed.  The pinned descriptor staging must already hold this call's
// descriptors; the caller has waited for a replay in flight before it wrote them.
template <typename Enq> static icet_status run_or_replay(icet_ctx* c, icet_ctx::GraphSlot& slot, const icet_ctx::GraphKey& key, Enq enq) {
    auto same = [](const icet_ctx::GraphKey& a, const icet_ctx::GraphKey& b) { return std::memcmp(&a, &b, sizeof(a)) == 0; };
    if (slot.have_graph && same(key, slot.key)) {
        HIPCHK(c, hipGraphLaunch(slot.exec, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_graph, c->stream)); c->graph_in_flight = true;
        c->timing_valid = false;
        return ICET_OK;
    }
    if (slot.have_seen && same(key, slot.seen)) {
        if (slot.have_graph) { (void)hipGraphExecDestroy(slot.exec); (void)hipGraphDestroy(slot.graph); slot.have_graph = false; }
        if (!c->ev_graph) HIPCHK(c, hipEventCreateWithFlags(&c->ev_graph, hipEventDisableTiming));
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        c->capturing = true;
        const icet_status es = enq();
        c->capturing = false;
        hipGraph_t g = nullptr;
        const hipError_t ee = hipStreamEndCapture(c->stream, &g);
        if (es != ICET_OK || ee != hipSuccess || !g) {           // could not capture (a capacity grew, an unsupported call): run this call eagerly and stop trying for this key
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            slot.have_seen = false;
            return enq();
        }
        hipGraphExec_t ge = nullptr;
        if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(g); (void)hipGetLastError(); slot.have_seen = false; return enq(); }
        slot.graph = g; slot.exec = ge; slot.key = key; slot.have_graph = true;
        HIPCHK(c, hipGraphLaunch(slot.exec, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_graph, c->stream)); c->graph_in_flight = true;
        c->timing_valid = false;
        return ICET_OK;
    }
    slot.seen = key; slot.have_seen = true;
    return enq();
}
}  // extern "C++"

static icet_status solve_device_part(icet_ctx* c, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan1, const icet_dev_scan* scan2,
                                     const float* d_x0, float* d_out, int64_t tot1) {
    int64_t tot2 = 0; for (int k = 0; k < n_pairs; k++) tot2 += scan2[k].n;
    c->kf_pairs = 0;                           // whatever keyframe was parked here is gone (runlen == 0 included: h_desc is overwritten below)
    icet_status s = ensure_workspace(c, p, n_pairs, tot1, tot2);
    if (s != ICET_OK) return s;
    // the previous call may still be copying out of the pinned descriptor staging; its kernels may still be running
    // (the device entry point never waits for them: calls queue up behind each other on the stream)
    if (c->desc_in_flight) { HIPCHK(c, hipEventSynchronize(c->ev_desc)); c->desc_in_flight = false; }
    if (c->graph_in_flight) { HIPCHK(c, hipEventSynchronize(c->ev_graph)); c->graph_in_flight = false; }     // a replay re-reads the pinned descriptor staging when it RUNS
    c->desc_kf_valid = c->desc_reg_valid = false;
    // A RAGGED throughput batch is laid out XCD-balanced (round 6; icet_layout.h says why and how).  The caller's order comes back in k_init_state (X0) and
    // k_gn_solve (results): LaunchCfg::pair_user.  Same bits: a pair's result does not depend on its slot (section 6 of DESIGN.md).
    c->perm_active = false; c->perm_pairs = n_pairs;
    std::vector<int32_t> order((size_t)n_pairs);
    for (int k = 0; k < n_pairs; k++) order[(size_t)k] = k;
    if (n_pairs > kUploadDescMaxPairs && p->runlen > 0) {
        std::vector<int64_t> size((size_t)n_pairs);
        for (int k = 0; k < n_pairs; k++) size[(size_t)k] = scan1[k].n + scan2[k].n;
        if (icet_layout::is_ragged(size)) {
            order = icet_layout::balanced_slot_order(size);
            for (int s = 0; s < n_pairs; s++) c->h_seg[n_pairs + 1 + s] = order[(size_t)s];
            c->perm_active = true;
        }
    }
    for (int s = 0; s < n_pairs; s++) {
        const int k = order[(size_t)s];
        PairDesc& d = c->h_desc[s];
        d.s1 = scan1[k].ptr; d.s2 = scan2[k].ptr;
        d.n1 = (int32_t)scan1[k].n; d.ld1 = (int32_t)scan1[k].ld; d.n2 = (int32_t)scan2[k].n; d.ld2 = (int32_t)scan2[k].ld;
        d.off1 = 0; d.off2 = 0;
    }
    if (p->runlen == 0) { c->armed_calls = 2; return write_runlen0(c, n_pairs, d_x0, d_out); }
    if (graph_eligible(c, p, n_pairs)) {
        icet_status ts = ensure_thresholds(c, p->bins_theta, p->bins_phi);
        if (ts != ICET_OK) return ts;
        // (the solve's last kernel raises the context's sync word: icet_sync)
        const bool own_word = !c->done_flag && c->h_sync_word;
        if (own_word) { if (c->armed_calls == 0) *static_cast<volatile int32_t*>(c->h_sync_word) = 0; c->done_flag = c->h_sync_word; }
        const icet_status rs = run_or_replay(c, c->g_solve, graph_key_of(c, p, n_pairs, d_x0, d_out, nullptr, nullptr), [&]() { return enqueue(c, p, n_pairs, d_x0, d_out, nullptr); });
        if (own_word) { c->done_flag = nullptr; c->armed_calls = (rs == ICET_OK && c->g_solve.have_graph) ? c->armed_calls + 1 : 2; }
        return rs;
    }
    c->armed_calls = 2;
    return enqueue(c, p, n_pairs, d_x0, d_out, nullptr);
}

icet_status icet_keyframe_device(icet_ctx* c, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan1) {
    return icet_keyframe_device_n(c, p, n_pairs, scan1, nullptr);
}

icet_status icet_keyframe_device_n(icet_ctx* c, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan1, const int32_t* d_rows) {
    if (!c) return ICET_ERR_BAD_ARG;
    if (!params_ok(p) || n_pairs < 1 || !scan1) { c->err = "bad argument"; return ICET_ERR_BAD_ARG; }
    int64_t tot1 = 0;
    for (int k = 0; k < n_pairs; k++) {
        const icet_dev_scan& a = scan1[k];
        if (a.n < 0 || a.ld < a.n || (a.n > 0 && !a.ptr) || a.ld >= ((int64_t)1 << 30)) { c->err = "bad scan descriptor"; return ICET_ERR_BAD_ARG; }
        tot1 += a.n;
    }
    HIPCHK(c, hipSetDevice(c->device));
    c->kf_pairs = 0; c->perm_active = false; c->armed_calls = 2;
    icet_status s = ensure_workspace(c, p, n_pairs, tot1, 0);
    if (s != ICET_OK) return s;
    // A sequential caller hands the SAME buffers to this half frame after frame (include/icet_nodes.h): the pinned descriptor staging then already holds
    // this call's scan-1 halves and is left alone -- rewriting it means waiting, on the HOST, for whatever replay or copy of this context still reads it,
    // and a burst of frames (icet_node_push_many_device) would have the host wait for the device twice per frame instead of running ahead
    bool same_desc = c->desc_kf_valid;                                     // (only what the previous call of THIS entry wrote counts: raw staging memory proves nothing)
    for (int k = 0; k < n_pairs && same_desc; k++) {
        const PairDesc& d = c->h_desc[k];
        same_desc = d.s1 == scan1[k].ptr && d.n1 == (int32_t)scan1[k].n && d.ld1 == (int32_t)scan1[k].ld;
    }
    if (!same_desc) {
        c->desc_reg_valid = false;                                         // the scan-2 halves are reset below
        if (c->desc_in_flight) { HIPCHK(c, hipEventSynchronize(c->ev_desc)); c->desc_in_flight = false; }
        if (c->graph_in_flight) { HIPCHK(c, hipEventSynchronize(c->ev_graph)); c->graph_in_flight = false; }
        for (int k = 0; k < n_pairs; k++) {
            PairDesc& d = c->h_desc[k];
            d.s1 = scan1[k].ptr; d.n1 = (int32_t)scan1[k].n; d.ld1 = (int32_t)scan1[k].ld;
            d.s2 = nullptr; d.n2 = 0; d.ld2 = 0; d.off1 = 0; d.off2 = 0;
        }
        c->desc_kf_valid = true;
    }
    if (graph_eligible(c, p, n_pairs)) {
        s = ensure_thresholds(c, p->bins_theta, p->bins_phi);
        if (s == ICET_OK) s = run_or_replay(c, c->g_keyframe, graph_key_of(c, p, n_pairs, d_rows, nullptr, nullptr, (const void*)1), [&]() { return enqueue_keyframe(c, p, n_pairs, nullptr, d_rows); });
    } else s = enqueue_keyframe(c, p, n_pairs, nullptr, d_rows);
    if (s != ICET_OK) return s;
    c->kf_pairs = n_pairs; c->kf_params = *p;
    return ICET_OK;
}

icet_status icet_register_device(icet_ctx* c, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan2, const float* d_x0, float* d_out) {
    return icet_register_device_n(c, p, n_pairs, scan2, nullptr, d_x0, d_out);
}

icet_status icet_register_device_n(icet_ctx* c, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan2, const int32_t* d_rows, const float* d_x0, float* d_out) {
    if (!c) return ICET_ERR_BAD_ARG;
    if (!params_ok(p) || n_pairs < 1 || !scan2 || !d_out) { c->err = "bad argument"; return ICET_ERR_BAD_ARG; }
    const icet_params& q = c->kf_params;
    if (c->kf_pairs != n_pairs || q.bins_phi != p->bins_phi || q.bins_theta != p->bins_theta || q.n != p->n || q.thresh != p->thresh || q.buff != p->buff ||
        ((q.flags ^ p->flags) & (ICET_FLAG_TRUE_SORT | ICET_FLAG_HALF_GAP_BOUNDS))) { c->err = "no keyframe with these parameters is parked in this context (icet_keyframe_device)"; return ICET_ERR_BAD_ARG; }
    int64_t tot2 = 0;
    for (int k = 0; k < n_pairs; k++) {
        const icet_dev_scan& b = scan2[k];
        if (b.n < 0 || b.ld < b.n || (b.n > 0 && !b.ptr) || b.ld >= ((int64_t)1 << 30)) { c->err = "bad scan descriptor"; return ICET_ERR_BAD_ARG; }
        tot2 += b.n;
    }
    HIPCHK(c, hipSetDevice(c->device));
    c->armed_calls = 2;
    if (p->runlen == 0) return write_runlen0(c, n_pairs, d_x0, d_out);
    icet_status s = ensure_workspace(c, p, n_pairs, 0, tot2);              // only the scan-2 overflow list can grow here: the keyframe tables stay
    if (s != ICET_OK) return s;
    bool same_desc = c->desc_reg_valid;                                    // (see icet_keyframe_device_n)
    for (int k = 0; k < n_pairs && same_desc; k++) {
        const PairDesc& d = c->h_desc[k];
        same_desc = d.s2 == scan2[k].ptr && d.n2 == (int32_t)scan2[k].n && d.ld2 == (int32_t)scan2[k].ld;
    }
    if (!same_desc) {
        if (c->desc_in_flight) { HIPCHK(c, hipEventSynchronize(c->ev_desc)); c->desc_in_flight = false; }
        if (c->graph_in_flight) { HIPCHK(c, hipEventSynchronize(c->ev_graph)); c->graph_in_flight = false; }
        for (int k = 0; k < n_pairs; k++) { PairDesc& d = c->h_desc[k]; d.s2 = scan2[k].ptr; d.n2 = (int32_t)scan2[k].n; d.ld2 = (int32_t)scan2[k].ld; }
        c->desc_reg_valid = true;
    }
    auto enq = [&]() -> icet_status {
        if (c->prologue) HIPCHK(c, c->prologue(c->prologue_user, c->stream));
        return enqueue_loop(c, p, n_pairs, d_x0, d_out, nullptr, true, nullptr, nullptr, d_rows);
    };
    if (graph_eligible(c, p, n_pairs)) return run_or_replay(c, c->g_loop, graph_key_of(c, p, n_pairs, d_x0, d_out, d_rows, (const void*)2), enq);
    return enq();
}

icet_status icet_solve_batch(icet_ctx* c, const icet_params* p, int32_t n_pairs,
                             const float* const* scan1, const int64_t* n1, const float* const* scan2, const int64_t* n2,
                             const float* x0, float* x_out, float* pred_stds_out, float* cov_out) {
    if (!c) return ICET_ERR_BAD_ARG;
    if (!params_ok(p) || n_pairs < 0 || (n_pairs > 0 && (!scan1 || !n1 || !scan2 || !n2 || !x_out || !pred_stds_out))) { c->err = "bad argument"; return ICET_ERR_BAD_ARG; }
    if (c->pend.active) { c->err = "icet_solve_begin without icet_solve_end on this context"; return ICET_ERR_BAD_ARG; }
    if (n_pairs == 0) return ICET_OK;
    int64_t tot1 = 0, tot2 = 0;
    for (int k = 0; k < n_pairs; k++) {
        if (n1[k] < 0 || n2[k] < 0 || (n1[k] > 0 && !scan1[k]) || (n2[k] > 0 && !scan2[k])) { c->err = "bad scan"; return ICET_ERR_BAD_ARG; }
        tot1 += (n1[k] + 63) / 64 * 64; tot2 += (n2[k] + 63) / 64 * 64;
    }
    HIPCHK(c, hipSetDevice(c->device));
    icet_status s = ensure_workspace(c, p, n_pairs, tot1, tot2);
    if (s == ICET_OK) s = ensure_out(c, n_pairs);
    if (s == ICET_OK) s = ensure_host_path(c);
    if (s == ICET_OK) s = ensure_stage(c, tot1, tot2);
    if (s != ICET_OK) return s;
    if (c->desc_in_flight) { HIPCHK(c, hipEventSynchronize(c->ev_desc)); c->desc_in_flight = false; }
    // Scan 1 of every pair goes up first on the solve stream (the keyframe build needs nothing else); the scan 2s follow on the copy
    // stream while the keyframe kernels run, and the loop waits for them.  Host scans are dense column-major (ld == n) in this entry point.
    int64_t o1 = 0, o2 = 0;
    c->desc_kf_valid = c->desc_reg_valid = false;
    for (int k = 0; k < n_pairs; k++) {
        const int64_t l1 = (n1[k] + 63) / 64 * 64, l2 = (n2[k] + 63) / 64 * 64;
        PairDesc& d = c->h_desc[k];
        d.s1 = c->d_stage1 + 3 * o1; d.s2 = c->d_stage2 + 3 * o2;
        d.n1 = (int32_t)n1[k]; d.ld1 = (int32_t)l1; d.n2 = (int32_t)n2[k]; d.ld2 = (int32_t)l2; d.off1 = 0; d.off2 = 0;
        HIPCHK(c, upload_scan(c->d_stage1 + 3 * o1, l1, scan1[k], n1[k], n1[k], c->stream));
        o1 += l1; o2 += l2;
    }
    const float* dx0 = nullptr;
    if (x0) { std::memcpy(c->h_x0, x0, sizeof(float) * 6 * n_pairs); HIPCHK(c, hipMemcpyAsync(c->d_x0, c->h_x0, sizeof(float) * 6 * n_pairs, hipMemcpyHostToDevice, c->stream)); dx0 = c->d_x0; }
    auto upload_scan2s = [&]() -> hipError_t {
        for (int k = 0; k < n_pairs; k++) {
            const PairDesc& d = c->h_desc[k];
            hipError_t e = upload_scan(const_cast<float*>(d.s2), d.ld2, scan2[k], n2[k], n2[k], c->st_copy);
            if (e != hipSuccess) return e;
        }
        return hipEventRecord(c->ev_s2, c->st_copy);
    };
    if (p->runlen == 0) s = write_runlen0(c, n_pairs, dx0, c->d_out);
    else {
        c->kf_pairs = 0; c->perm_active = false; c->armed_calls = 2;
        s = enqueue_keyframe(c, p, n_pairs, nullptr);
        if (s == ICET_OK) {
            const hipError_t e = upload_scan2s();
            if (e != hipSuccess) { (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->st_copy); c->err = std::string("scan-2 upload: ") + hipGetErrorString(e); return ICET_ERR_HIP; }
            s = enqueue_loop(c, p, n_pairs, dx0, c->d_out, nullptr, false, nullptr, c->ev_s2);
        }
    }
    if (s != ICET_OK) { (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->st_copy); return s; }
    HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_out, sizeof(float) * 48 * n_pairs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < n_pairs; k++) {
        std::memcpy(x_out + 6 * k, c->h_out + 48 * k, 6 * sizeof(float));
        std::memcpy(pred_stds_out + 6 * k, c->h_out + 48 * k + 6, 6 * sizeof(float));
        if (cov_out) std::memcpy(cov_out + 36 * k, c->h_out + 48 * k + 12, 36 * sizeof(float));
    }
    return ICET_OK;
}

// The constructor replacement in two halves.  icet_solve_begin enqueues the uploads, the whole registration and the copy of the results
// and returns while the device works (the scans must stay untouched until icet_solve_end: the runtime may still be reading them);
// icet_solve_end waits and fills the outputs named at begin.  What the host does in between -- the reference's constructor deep-copies
// both scans into its members (src/icet.cpp:30,33), include/icet.h does the same there -- overlaps with the device.
icet_status icet_solve_begin(icet_ctx* c, const icet_params* p, const float* scan1, int64_t n1, int64_t ld1,
                             const float* scan2, int64_t n2, int64_t ld2, const float x0[6],
                             float x_out[6], float pred_stds_out[6], float cov_out[36], icet_aux* aux) {
    if (!c) return ICET_ERR_BAD_ARG;
    if (!params_ok(p) || n1 < 0 || n2 < 0 || ld1 < n1 || ld2 < n2 || (n1 > 0 && !scan1) || (n2 > 0 && !scan2) || !x0 || !x_out || !pred_stds_out) {
        c->err = "bad argument"; return ICET_ERR_BAD_ARG;
    }
    if (c->pend.active) { c->err = "icet_solve_begin without icet_solve_end on this context"; return ICET_ERR_BAD_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const int V = p->bins_phi * p->bins_theta;
    const int64_t l1 = (n1 + 63) / 64 * 64, l2 = (n2 + 63) / 64 * 64;
    icet_status s = ensure_workspace(c, p, 1, l1, l2);
    if (s == ICET_OK) s = ensure_out(c, 1);
    if (s == ICET_OK) s = ensure_host_path(c);
    if (s == ICET_OK) s = ensure_stage(c, l1, l2);
    if (s == ICET_OK) s = ensure_pack(c, V, p->runlen);
    if (s != ICET_OK) return s;
    const bool want_pts2 = aux && aux->points2 && n2 > 0 && p->runlen > 0;
    if (want_pts2 && (p->flags & ICET_FLAG_ROUNDTRIP_SCAN2) && (size_t)3 * n2 > c->cap_pts2) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->st_copy) HIPCHK(c, hipStreamSynchronize(c->st_copy));
        if (c->h_pts2) { HIPCHK(c, hipHostFree(c->h_pts2)); c->h_pts2 = nullptr; c->cap_pts2 = 0; }
        const size_t want = (size_t)3 * n2 + (size_t)3 * n2 / 8;
        HIPCHK(c, dev_realloc(c->d_pts2, want));
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_pts2), want * sizeof(float)));
        c->cap_pts2 = want;
    }
    const bool want_side1 = aux && p->runlen > 0 && n1 > 0 && (aux->points1_spherical || aux->point_index1 || aux->bin_start1);
    const bool want_side2 = aux && p->runlen > 0 && n2 > 0 && (aux->points2_spherical || aux->voxel2);
    if (want_side1 && (size_t)n1 > c->cap_side1) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, dev_realloc(c->d_sph1, (size_t)3 * n1)); HIPCHK(c, dev_realloc(c->d_idx1, (size_t)n1)); c->cap_side1 = (size_t)n1;
    }
    if (want_side2 && ((size_t)n2 > c->cap_side2 || (size_t)3 * n2 > c->cap_pts2)) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->st_copy) HIPCHK(c, hipStreamSynchronize(c->st_copy));
        HIPCHK(c, dev_realloc(c->d_sph2, (size_t)3 * n2)); HIPCHK(c, dev_realloc(c->d_vox2, (size_t)n2)); c->cap_side2 = (size_t)n2;
        if ((size_t)3 * n2 > c->cap_pts2) {
            if (c->h_pts2) { HIPCHK(c, hipHostFree(c->h_pts2)); c->h_pts2 = nullptr; }
            HIPCHK(c, dev_realloc(c->d_pts2, (size_t)3 * n2));
            HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_pts2), (size_t)3 * n2 * sizeof(float)));
            c->cap_pts2 = (size_t)3 * n2;
        }
    }
    if (c->desc_in_flight) { HIPCHK(c, hipEventSynchronize(c->ev_desc)); c->desc_in_flight = false; }
    // (no stream synchronisation up front: the staging buffers, the result block and the pinned x0 are only ever used by the host-pointer
    // entry points, each of which has drained the stream before it returned)
    std::memcpy(c->h_x0, x0, 6 * sizeof(float));                              // pinned: k_init_state reads X0 straight from it (no H2D command)
    HIPCHK(c, upload_scan(c->d_stage1, l1, scan1, n1, ld1, c->stream));
    c->desc_kf_valid = c->desc_reg_valid = false;
    PairDesc& d = c->h_desc[0];
    d.s1 = c->d_stage1; d.s2 = c->d_stage2; d.n1 = (int32_t)n1; d.ld1 = (int32_t)l1; d.n2 = (int32_t)n2; d.ld2 = (int32_t)l2; d.off1 = 0; d.off2 = 0;
    const AuxLayout L = aux_layout(V, p->runlen);
    float* h_res = reinterpret_cast<float*>(c->h_pack) + L.out;               // the 48 result floats on the host
    float* d_res = reinterpret_cast<float*>(c->d_pack) + L.out;               // ... and where k_gn_solve writes them
    AuxDev ad = c->aux_dev;
    icet_ctx::Pending& q = c->pend;
    q = icet_ctx::Pending{};
    icet_status st = ICET_OK;
    if (p->runlen == 0) {
        // the reference's constructor leaves X = X0, pred_stds = 0 when runlen == 0 (src/icet.cpp:36-37,47); nothing to enqueue
        std::memset(h_res, 0, 48 * sizeof(float)); std::memcpy(h_res, x0, 6 * sizeof(float));
        if (aux && aux->points2 && n2 > 0)                                    // `points2` of an object that never iterated is its copy of scan 2 (src/icet.cpp:33)
            for (int k = 0; k < 3; k++) std::memcpy(aux->points2 + (size_t)k * n2, scan2 + (size_t)k * ld2, (size_t)n2 * sizeof(float));
    } else {
        if (aux) {
            const size_t rl = p->runlen, v = (size_t)V;
            // per-iteration integer tables: k_gn_solve writes the active voxels only, so they start from zero -- and are neither cleared
            // nor written when nobody asked for them
            if (aux->n2_raw) HIPCHK(c, hipMemsetAsync(ad.n2_raw, 0, sizeof(int32_t) * rl * v, c->stream)); else ad.n2_raw = nullptr;
            if (aux->n2_in) HIPCHK(c, hipMemsetAsync(ad.n2_in, 0, sizeof(int32_t) * rl * v, c->stream)); else ad.n2_in = nullptr;
            q.kf_tables = aux->cluster_bounds || aux->has_fit || aux->mu1 || aux->sigma1 || aux->evecs1 || aux->l_diag || aux->test_points;
            q.tail_ints = aux->n1_raw || aux->n2_raw || aux->n2_in;
            q.pts2 = want_pts2 && !want_side2; q.pts2_dev = q.pts2 && (p->flags & ICET_FLAG_ROUNDTRIP_SCAN2) != 0;
            q.side1 = want_side1; q.side2 = want_side2; q.n1 = n1;
            q.scan2 = scan2; q.ld2 = ld2;
            if (!want_pts2 && !want_side2) ad.xf_last = nullptr;
        }
        c->kf_pairs = 0; c->perm_active = false; c->armed_calls = 2;
        st = enqueue_keyframe(c, p, 1, aux ? &ad : nullptr);
        if (st == ICET_OK && want_side1) {                                     // points1Spherical / pointIndices1 from the tables the keyframe build has just left
            const LaunchCfg scfg = make_cfg(c, p, 1);
            HIPCHK(c, launch_side_scan1(c->w, scfg, c->d_sph1, c->d_idx1, c->stream));
        }
        if (st == ICET_OK) {
            // scan 2 goes up on the copy stream while the keyframe kernels enqueued above run; the loop waits for ev_s2
            hipError_t e = upload_scan(c->d_stage2, l2, scan2, n2, ld2, c->st_copy);
            if (e == hipSuccess) e = hipEventRecord(c->ev_s2, c->st_copy);
            if (e == hipSuccess && q.kf_tables) {
                // the keyframe tables are final now: they travel to the host on the copy stream while the loop iterates
                e = hipEventRecord(c->ev_kf, c->stream);
                if (e == hipSuccess) e = hipStreamWaitEvent(c->st_copy, c->ev_kf, 0);
                if (e == hipSuccess) e = hipMemcpyAsync(c->h_pack + L.bounds, c->d_pack + L.bounds, (L.kf_end - L.bounds) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->st_copy);
                if (e == hipSuccess) e = hipEventRecord(c->ev_kfd, c->st_copy);
            }
            if (e != hipSuccess) { (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->st_copy); c->err = std::string("scan-2 upload / table download: ") + hipGetErrorString(e); return ICET_ERR_HIP; }
            st = enqueue_loop(c, p, 1, c->h_x0, d_res, aux ? &ad : nullptr, false, q.pts2_dev ? c->h_pts2 : nullptr, c->ev_s2);
            if (st == ICET_OK) HIPCHK(c, hipMemcpyAsync(c->h_pack + L.out, c->d_pack + L.out, (aux ? L.small_end : 48) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        }
        if (st == ICET_OK && want_side2) {
            // points2 / points2Spherical / the rows' voxels of the LAST fitScan2: one kernel behind the loop, with the transform record that
            // iteration used (it reads the scan the loop read: the round-tripped copy under ICET_FLAG_ROUNDTRIP_SCAN2)
            LaunchCfg scfg = make_cfg(c, p, 1);
            Workspace wl = c->w; if (scfg.rt2) wl.desc = c->w.desc_rt;
            HIPCHK(c, launch_side_scan2(wl, scfg, ad.xf_last, c->d_pts2, c->d_sph2, c->d_vox2, c->stream));
        }
        if (st == ICET_OK && q.tail_ints)
            HIPCHK(c, hipMemcpyAsync(c->h_pack + L.n1_raw, c->d_pack + L.n1_raw, (L.ints_end - L.n1_raw) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    }
    if (st != ICET_OK) { (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->st_copy); return st; }
    q.active = true; q.x_out = x_out; q.ps_out = pred_stds_out; q.cov_out = cov_out; q.has_aux = aux != nullptr && p->runlen > 0;
    q.aux = aux ? *aux : icet_aux{}; q.V = V; q.rl = p->runlen; q.n2 = n2;
    return ICET_OK;
}

// The keyframe half of a pending solve: returns when the keyframe tables named at begin (cluster_bounds, has_fit, mu1, sigma1, evecs1,
// l_diag, test_points) are in the caller's arrays -- the loop is still iterating on the device.  Optional; icet_solve_end does it otherwise.
icet_status icet_solve_keyframe_tables(icet_ctx* c) {
    if (!c) return ICET_ERR_BAD_ARG;
    icet_ctx::Pending& q = c->pend;
    if (!q.active) { c->err = "icet_solve_keyframe_tables without icet_solve_begin"; return ICET_ERR_BAD_ARG; }
    if (!q.has_aux || !q.kf_tables || q.kf_done) return ICET_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->ev_kfd));
    const AuxLayout L = aux_layout(q.V, q.rl);
    const icet_aux& a = q.aux; const size_t v = (size_t)q.V;
    auto add = [&](void* dst, size_t at, size_t n) { if (dst && n) std::memcpy(dst, c->h_pack + at, n * sizeof(uint32_t)); };
    add(a.cluster_bounds, L.bounds, v * 6); add(a.has_fit, L.has_fit, v); add(a.mu1, L.mu1, v * 3); add(a.sigma1, L.sigma1, v * 9);
    add(a.evecs1, L.evecs1, v * 9); add(a.l_diag, L.l_diag, v * 3); add(a.test_points, L.test_points, v * 18);
    q.kf_done = true;
    return ICET_OK;
}

icet_status icet_solve_end(icet_ctx* c) {
    if (!c) return ICET_ERR_BAD_ARG;
    icet_ctx::Pending& q = c->pend;
    if (!q.active) { c->err = "icet_solve_end without icet_solve_begin"; return ICET_ERR_BAD_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    // what is ready before the loop has finished is moved into the caller's arrays while the device still iterates
    icet_status ks = icet_solve_keyframe_tables(c);
    q.active = false;
    if (q.has_aux && q.pts2 && ks == ICET_OK) {
        if (q.pts2_dev) {
            HIPCHK(c, hipEventSynchronize(c->ev_pts2));
            std::memcpy(q.aux.points2, c->h_pts2, (size_t)3 * q.n2 * sizeof(float));
        } else {
            // the transform of the last iteration is known (its solve is still running): scan 2 is transformed here, under the device's last iteration
            HIPCHK(c, hipEventSynchronize(c->ev_prev));
            const AuxLayout Lp = aux_layout(q.V, q.rl);
            host_points2(reinterpret_cast<const float*>(c->h_pack) + Lp.xf_last, q.scan2, q.ld2, q.n2, q.aux.points2);
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (q.rl > 0) HIPCHK(c, hipStreamSynchronize(c->st_copy));
    if (ks != ICET_OK) return ks;
    const AuxLayout L = aux_layout(q.V, q.rl);
    const float* out = reinterpret_cast<const float*>(c->h_pack) + L.out;
    std::memcpy(q.x_out, out, 6 * sizeof(float));
    std::memcpy(q.ps_out, out + 6, 6 * sizeof(float));
    if (q.cov_out) std::memcpy(q.cov_out, out + 12, 36 * sizeof(float));
    if (q.has_aux) {
        const icet_aux& a = q.aux; const size_t rl = q.rl, v = (size_t)q.V;
        auto add = [&](void* dst, size_t at, size_t n) { if (dst && n) std::memcpy(dst, c->h_pack + at, n * sizeof(uint32_t)); };
        add(a.x_hist, L.x_hist, rl * 6); add(a.htwh, L.htwh, rl * 36); add(a.htwdz, L.htwdz, rl * 6); add(a.cond_info, L.cond, rl * 8);
        add(a.n1_raw, L.n1_raw, v); add(a.n2_raw, L.n2_raw, rl * v); add(a.n2_in, L.n2_in, rl * v);
        // the per-point tables go from HBM straight into the caller's arrays (several MB; only when asked for)
        if (q.side1) {
            if (a.points1_spherical) HIPCHK(c, hipMemcpy(a.points1_spherical, c->d_sph1, sizeof(float) * 3 * (size_t)q.n1, hipMemcpyDeviceToHost));
            if (a.point_index1) HIPCHK(c, hipMemcpy(a.point_index1, c->d_idx1, sizeof(int32_t) * (size_t)q.n1, hipMemcpyDeviceToHost));
            if (a.bin_start1) HIPCHK(c, hipMemcpy(a.bin_start1, c->w.bin_start, sizeof(int32_t) * (v + 1), hipMemcpyDeviceToHost));
        }
        if (q.side2) {
            if (a.points2) HIPCHK(c, hipMemcpy(a.points2, c->d_pts2, sizeof(float) * 3 * (size_t)q.n2, hipMemcpyDeviceToHost));
            if (a.points2_spherical) HIPCHK(c, hipMemcpy(a.points2_spherical, c->d_sph2, sizeof(float) * 3 * (size_t)q.n2, hipMemcpyDeviceToHost));
            if (a.voxel2) HIPCHK(c, hipMemcpy(a.voxel2, c->d_vox2, sizeof(int32_t) * (size_t)q.n2, hipMemcpyDeviceToHost));
        }
    }
    return ICET_OK;
}

icet_status icet_solve(icet_ctx* c, const icet_params* p, const float* scan1, int64_t n1, int64_t ld1,
                       const float* scan2, int64_t n2, int64_t ld2, const float x0[6],
                       float x_out[6], float pred_stds_out[6], float cov_out[36], icet_aux* aux) {
    const icet_status s = icet_solve_begin(c, p, scan1, n1, ld1, scan2, n2, ld2, x0, x_out, pred_stds_out, cov_out, aux);
    return s == ICET_OK ? icet_solve_end(c) : s;
}

icet_status icet_debug_fetch(icet_ctx* c, int32_t what, void* out, int64_t count) {
    if (!c || !out || count < 0) return ICET_ERR_BAD_ARG;
    const Workspace& w = c->w;
    if (what == 6) { if (count != 1) return ICET_ERR_BAD_ARG; *static_cast<int32_t*>(out) = c->lds_rank_ok; return ICET_OK; }   // no device array: the verdict of lds_rank_selftest
    const void* src = nullptr; int64_t cap = w.cap_n1; size_t elem = 4;
    switch (what) {
        case 0: src = w.r1; break;
        case 1: src = w.bin16; elem = 2; break;
        case 3: src = w.src; break;
        case 4: src = w.flags; cap = w.cap_pairs; break;
        case 5: src = w.rt2; cap = 3 * w.cap_rt2; break;      // ICET_FLAG_ROUNDTRIP_SCAN2: the round-tripped copy of scan 2 (x | y | z, leading dimension = n2 rounded up to 64)
        default: c->err = "unknown array id"; return ICET_ERR_BAD_ARG;
    }
    if (!src || count > cap) { c->err = "nothing to fetch / count too large"; return ICET_ERR_BAD_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, src, (size_t)count * elem, hipMemcpyDeviceToHost));
    return ICET_OK;
}

// Test hook: the 6x6 tail of an iteration (src/icet.cpp:410-433) on n host-side (HTWH, HTWdz), evaluated by the device function k_gn_solve runs.
icet_status icet_debug_gn_tail(icet_ctx* c, const float* htwh, const float* htwdz, int32_t n, float* out) {
    if (!c || !htwh || !htwdz || !out || n < 0) return ICET_ERR_BAD_ARG;
    if (n == 0) return ICET_OK;
    HIPCHK(c, hipSetDevice(c->device));
    float* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), sizeof(float) * (size_t)n * (36 + 6 + 56)));
    float* d_H = d; float* d_g = d + (size_t)n * 36; float* d_o = d_g + (size_t)n * 6;
    hipError_t e = hipMemcpyAsync(d_H, htwh, sizeof(float) * (size_t)n * 36, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_g, htwdz, sizeof(float) * (size_t)n * 6, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_gn_tail_debug(d_H, d_g, d_o, n, (float)(c->tune.gn_cond_bound * c->tune.gn_cond_bound), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_o, sizeof(float) * (size_t)n * 56, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) { c->err = std::string("icet_debug_gn_tail: ") + hipGetErrorString(e); return ICET_ERR_HIP; }
    return ICET_OK;
}

// Test hook: the per-voxel weight's pseudo-inverse under ICET_FLAG_REFERENCE_W (src/icet.cpp:320-321) on n host-side 3 x 3 matrices, through the device function the solve runs.
icet_status icet_debug_pinv3(icet_ctx* c, const float* a, int32_t n, float* out) {
    if (!c || !a || !out || n < 0) return ICET_ERR_BAD_ARG;
    if (n == 0) return ICET_OK;
    HIPCHK(c, hipSetDevice(c->device));
    float* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), sizeof(float) * (size_t)n * 18));
    hipError_t e = hipMemcpyAsync(d, a, sizeof(float) * (size_t)n * 9, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_pinv3_debug(d, d + (size_t)n * 9, n, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d + (size_t)n * 9, sizeof(float) * (size_t)n * 9, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) { c->err = std::string("icet_debug_pinv3: ") + hipGetErrorString(e); return ICET_ERR_HIP; }
    return ICET_OK;
}

icet_status icet_set_option(icet_ctx* c, const char* name, double value) {
    if (!c || !name) return ICET_ERR_BAD_ARG;
    Tuning& t = c->tune;
    const std::string k(name);
    const int iv = (int)value;
    if (k == "lds_slots") t.lds_slots = iv < 0 ? 0 : iv;
    else if (k == "acc_pts") t.acc_pts = iv < 1 ? 1 : iv;
    else if (k == "acc_blocks") t.acc_blocks = iv < 1 ? 1 : iv;
    else if (k == "force_exact") t.force_exact = iv != 0;
    else if (k == "library_sort") {
#ifdef ICET_DIAG_LIBSORT
        t.library_sort = iv != 0;
#else
        if (iv != 0) { c->err = "library_sort: this build has ONE sort, the hand-written rank sort (the rocPRIM A/B backend is compiled in with `make EXTRA=-DICET_DIAG_LIBSORT`)"; return ICET_ERR_UNSUPPORTED; }
#endif
    }
    else if (k == "kf_pts") t.kf_pts = iv < 1 ? 1 : (iv > kKfMaxPtsPerThread ? kKfMaxPtsPerThread : iv);
    else if (k == "batch_parts") t.batch_parts = iv < 0 ? 0 : (iv > 8 ? 8 : iv);
    else if (k == "batch_stage") t.batch_stage = (iv < 0 || iv > 4) ? 0 : iv;
    else if (k == "rs_cap") t.rs_cap = iv < 0 ? 0 : iv;
    else if (k == "rs_max_cell") t.rs_max_cell = iv < 0 ? 0 : iv;
    else if (k == "exec_bits_lds") t.exec_bits_lds = iv != 0;
    else if (k == "lds_rank") t.lds_rank = iv < 0 ? -1 : (iv != 0);
    else if (k == "exec_pairwise") t.exec_pairwise = iv < 0 ? -1 : (iv != 0);
    else if (k == "fuse_solve") { t.fuse_solve = iv < 0 ? -1 : (iv != 0); c->g_solve.have_seen = c->g_keyframe.have_seen = c->g_loop.have_seen = false; }
    else if (k == "keep") t.keep = iv != 0;
    else if (k == "keep_from") t.keep_from = iv < 0 ? 0 : iv;
    else if (k == "keep_budget_t") { if (!(value > 0.0 && value <= 100.0)) { c->err = "keep_budget_t must lie in (0, 100]"; return ICET_ERR_BAD_ARG; } t.keep_budget_t = value; }
    else if (k == "keep_budget_r") { if (!(value > 0.0 && value <= 1.0)) { c->err = "keep_budget_r must lie in (0, 1]"; return ICET_ERR_BAD_ARG; } t.keep_budget_r = value; }
    else if (k == "keep_check_scale") t.keep_check_scale = value > 0 ? value : 1.0;      // timing experiments (Tuning)
    else if (k == "graph") { c->graph_mode = iv < 0 ? -1 : (iv != 0); c->g_solve.have_seen = c->g_keyframe.have_seen = c->g_loop.have_seen = false; }
    else if (k == "gn_cond_bound") { if (!(value >= 0.0 && value <= 1e6)) { c->err = "gn_cond_bound must lie in [0, 1e6]"; return ICET_ERR_BAD_ARG; } t.gn_cond_bound = value; c->g_solve.have_seen = c->g_keyframe.have_seen = c->g_loop.have_seen = false; }
    else if (k == "guard_scale") { if (!(value >= 1.0 && value <= 1024.0)) { c->err = "guard_scale must lie in [1, 1024]"; return ICET_ERR_BAD_ARG; } t.guard_scale = value; c->w.thr_T = 0; }   // tables are rebuilt by the next call
    else if (k == "lut_polar_quantile") { if (!(value >= 0.0 && value <= 1.0)) { c->err = "lut_polar_quantile must lie in [0, 1]"; return ICET_ERR_BAD_ARG; } t.lut_polar_quantile = value; c->w.thr_T = 0; }
    else { c->err = "unknown option: " + k; return ICET_ERR_BAD_ARG; }
    return ICET_OK;
}

} // extern "C" (reopened below)
void icet_ctx_set_stream(icet_ctx* c, hipStream_t s) { if (c) c->stream = s; }
void icet_ctx_set_done_flag(icet_ctx* c, int32_t* pinned_word) { if (c) c->done_flag = pinned_word; }
void icet_ctx_set_prologue(icet_ctx* c, hipError_t (*fn)(void*, hipStream_t), void* user, int64_t key) { if (c) { c->prologue = fn; c->prologue_user = user; c->prologue_key = key; } }
extern "C" {
void* icet_stream(icet_ctx* c) { return c ? reinterpret_cast<void*>(c->stream) : nullptr; }
int icet_device(const icet_ctx* c) { return c ? c->device : -1; }

icet_status icet_last_timing(icet_ctx* c, float out_ms[4]) {
    if (!c || !out_ms) return ICET_ERR_BAD_ARG;
    if (!c->timing_valid) { c->err = "no timed call yet"; return ICET_ERR_BAD_ARG; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float a = 0, b = 0, acc = 0;
    HIPCHK(c, hipEventElapsedTime(&a, c->ev_a, c->ev_b));
    HIPCHK(c, hipEventElapsedTime(&b, c->ev_b, c->ev_c));
    for (int it = 0; it < c->last_iters; it++) { float t = 0; HIPCHK(c, hipEventElapsedTime(&t, c->ev_acc[2 * it], c->ev_acc[2 * it + 1])); acc += t; }
    out_ms[0] = a; out_ms[1] = b; out_ms[2] = c->last_iters ? acc : -1.f; out_ms[3] = (float)c->last_iters;
    return ICET_OK;
}

icet_status icet_last_timing_iters(icet_ctx* c, float* acc_ms, int32_t cap, int32_t* n_out) {
    if (!c || !acc_ms || !n_out || cap < 0) return ICET_ERR_BAD_ARG;
    if (!c->timing_valid) { c->err = "no timed call yet"; return ICET_ERR_BAD_ARG; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int n = c->last_iters < cap ? c->last_iters : cap;
    for (int it = 0; it < n; it++) HIPCHK(c, hipEventElapsedTime(&acc_ms[it], c->ev_acc[2 * it], c->ev_acc[2 * it + 1]));
    *n_out = n;
    return ICET_OK;
}

icet_status icet_keep_stats(icet_ctx* c, int32_t n_pairs, int32_t* out) {
    if (!c || !out || n_pairs < 0) return ICET_ERR_BAD_ARG;
    if (!c->w.keep_state || n_pairs > c->w.cap_keep_pairs) { c->err = "no keep-list state for that many pairs (the last call did not use the keep list)"; return ICET_ERR_BAD_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<KeepState> h((size_t)n_pairs);
    HIPCHK(c, hipMemcpy(h.data(), c->w.keep_state, sizeof(KeepState) * (size_t)n_pairs, hipMemcpyDeviceToHost));
    for (int s = 0; s < n_pairs; s++) {
        const int k = (c->perm_active && c->perm_pairs == n_pairs) ? c->h_seg[n_pairs + 1 + s] : s;      // (a ragged batch sits XCD-balanced in the tables: back to the caller's order)
        out[4 * k] = h[s].mode; out[4 * k + 1] = h[s].n_keep; out[4 * k + 2] = h[s].list_passes; out[4 * k + 3] = h[s].builds;
    }
    return ICET_OK;
}

}  // extern "C"
