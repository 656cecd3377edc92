// icet_amd/csrc/icet_nodes.hip -- the callers on either side of the hot path, on the device (include/icet_nodes.h):
// the per-frame body of the reference's odometry_node / map_maker_node (src/odometry.cpp:46-98,
// src/simpleMapMaker.cpp:86-172) and the HD-map FIFO `EigenQueue` (src/simpleMapMaker.cpp:18-59).
//
// Built only on the public C ABI of icet_hip.h (one single-pair icet_solve_batch_device per frame) plus three small
// HBM-bound kernels of its own:
//   k_range_count / k_range_scan / k_range_scatter   the `row.norm() > minD` filter as a stable stream compaction
//                                                    (12 B read twice + 12 B written per kept row)
//   k_map_add_scan                                    EigenQueue::add_new_scan: the down-sampled rows enter the ring and
//                                                    the whole ring is re-expressed as (row - t) * R^-1 in ONE pass
//                                                    (12 B read + 12 B written per ring row)
// Host-side scalar work (pose chaining, quaternion, the 3x3 inverse, std::shuffle of the index vector) stays on the
// host as in the reference: it is O(1) or inherently sequential (Fisher-Yates with one RNG stream).
// No CPU implementation of the solve lives here; without a device every entry point fails with an error status.
#include "../../include/icet_nodes.h"
#include "icet_shuffle.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <numeric>
#include <random>
#include <future>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <string>
#include <vector>

namespace {

constexpr int kFB = 256;            // threads per block of the filter kernels
constexpr int kFRows = 8;           // rows per thread: a block owns 2048 consecutive rows

// Where a frame's raw scan is, for filter launches that are part of a captured graph (round 6): the record lives in pinned host memory, the host fills it in before
// every replay and the kernels read it (the same arrangement as the loop's X0) -- the launches themselves (grid from the buffers' CAPACITY) never change.
struct FrameDesc { const float* x; int32_t n, ld; };

__device__ __forceinline__ bool keep_row(float x, float y, float z, float min_range) {
    float d;
    {
#pragma clang fp contract(off)
        float s = x * x + y * y;      // Eigen's row(i).norm(): sqrt of the plain sum of squares (src/odometry.cpp:61-64)
        s = s + z * z;
        d = sqrtf(s);
    }
    return d > min_range;
}

// pass 1: kept rows per block
__global__ __launch_bounds__(kFB) void k_range_count(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n,
                                                    float min_range, int32_t* __restrict__ counts, const FrameDesc* __restrict__ fd = nullptr) {
    __shared__ int wsum[kFB / 64];
    if (fd) { const FrameDesc d = *fd; x = d.x; y = d.x + d.ld; z = d.x + 2 * (size_t)d.ld; n = d.n; }
    const int base = blockIdx.x * kFB * kFRows;
    int c = 0;
#pragma unroll
    for (int k = 0; k < kFRows; k++) {
        const int i = base + k * kFB + threadIdx.x;
        if (i < n) c += keep_row(x[i], y[i], z[i], min_range) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < kFB / 64; w++) t += wsum[w]; counts[blockIdx.x] = t; }
}

// pass 2: exclusive scan of the block counts (one block; at most a few thousand entries), total -> n_kept
__global__ __launch_bounds__(kFB) void k_range_scan(const int32_t* __restrict__ counts, int32_t* __restrict__ bases, int n_blocks, int32_t* __restrict__ n_kept, int32_t* __restrict__ n_kept_copy = nullptr) {
    __shared__ int part[kFB];
    const int per = (n_blocks + kFB - 1) / kFB;
    const int lo = threadIdx.x * per, hi = min(n_blocks, lo + per);
    int s = 0;
    for (int i = lo; i < hi; i++) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int t = 0; t < kFB; t++) { const int v = part[t]; part[t] = run; run += v; } *n_kept = run; if (n_kept_copy) { *n_kept_copy = run; __threadfence_system(); } }      // (the copy lives in pinned host memory: a host thread may be watching it)
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; i++) { bases[i] = run; run += counts[i]; }
}

// pass 3: stable scatter.  Row order inside a block is k-major (row = base + k * kFB + thread), so the rank of a kept
// row is: kept rows in earlier k-slices + kept rows of lower threads in its own slice.
__global__ __launch_bounds__(kFB) void k_range_scatter(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n,
                                                      float min_range, const int32_t* __restrict__ bases,
                                                      float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz, const FrameDesc* __restrict__ fd = nullptr,
                                                      const int32_t* __restrict__ counts = nullptr, int n_blocks = 0, int32_t* __restrict__ n_kept = nullptr, int32_t* __restrict__ n_kept_copy = nullptr) {
    __shared__ int wcnt[kFRows][kFB / 64];
    __shared__ int s_part[kFB / 64], s_tot[kFB / 64];
    if (fd) { const FrameDesc d = *fd; x = d.x; y = d.x + d.ld; z = d.x + 2 * (size_t)d.ld; n = d.n; }
    // counts: pass 2 folded in (a one-launch frame, at most a few hundred blocks): every block adds up the counts of the blocks in front of it itself, block 0 also the
    // total -- k_range_scan's 4.6 us launch is what the frame saves
    int part = 0, tot = 0;
    if (counts) for (int i = threadIdx.x; i < n_blocks; i += kFB) { const int c = counts[i]; tot += c; part += (i < (int)blockIdx.x) ? c : 0; }
    const int base = blockIdx.x * kFB * kFRows;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float vx[kFRows], vy[kFRows], vz[kFRows];
    bool keep[kFRows];
    int below[kFRows];
#pragma unroll
    for (int k = 0; k < kFRows; k++) {
        const int i = base + k * kFB + threadIdx.x;
        keep[k] = false; vx[k] = vy[k] = vz[k] = 0.f;
        if (i < n) { vx[k] = x[i]; vy[k] = y[i]; vz[k] = z[i]; keep[k] = keep_row(vx[k], vy[k], vz[k], min_range); }
        const unsigned long long m = __ballot(keep[k]);
        below[k] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[k][wave] = __popcll(m);
    }
    if (counts) {
        for (int o = 32; o > 0; o >>= 1) { part += __shfl_down(part, o); tot += __shfl_down(tot, o); }
        if (lane == 0) { s_part[wave] = part; s_tot[wave] = tot; }
    }
    __syncthreads();
    int run;
    if (counts) {
        run = 0; int t = 0;
#pragma unroll
        for (int w = 0; w < kFB / 64; w++) { run += s_part[w]; t += s_tot[w]; }
        if (blockIdx.x == 0 && threadIdx.x == 0) { *n_kept = t; if (n_kept_copy) { *n_kept_copy = t; __threadfence_system(); } }
    } else run = bases[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kFRows; k++) {
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kFB / 64; w++) { const int cw = wcnt[k][w]; before += (w < wave) ? cw : 0; total += cw; }
        if (keep[k]) { const int o = run + before + below[k]; ox[o] = vx[k]; oy[o] = vy[k]; oz[o] = vz[k]; }
        run += total;
    }
}

// EigenQueue::add_new_scan (src/simpleMapMaker.cpp:34-41): rows [pos, pos + m) (mod cap) take the down-sampled scan
// rows, then EVERY ring row becomes (row - trans) * Rinv.  One pass over the ring.
__global__ __launch_bounds__(256) void k_map_add_scan(float* __restrict__ qx, float* __restrict__ qy, float* __restrict__ qz, int cap, int pos, int m,
                                                     const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                                                     const int32_t* __restrict__ idx, float tx, float ty, float tz,
                                                     float i00, float i01, float i02, float i10, float i11, float i12, float i20, float i21, float i22) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) {
        int j = i - pos; if (j < 0) j += cap;             // position in this frame's write window
        float a, b, c;
        if (j < m) { const int r = idx[j]; a = sx[r]; b = sy[r]; c = sz[r]; }
        else { a = qx[i]; b = qy[i]; c = qz[i]; }
        a -= tx; b -= ty; c -= tz;
        {
#pragma clang fp contract(off)
            qx[i] = (a * i00 + b * i10) + c * i20;
            qy[i] = (a * i01 + b * i11) + c * i21;
            qz[i] = (a * i02 + b * i12) + c * i22;
        }
    }
}

// scan2_in_scan1_frame = (pcl_matrix * rot_mat.inverse()).rowwise() - trans  (src/scanMatcher.cpp:76): rotate first, then subtract
__global__ __launch_bounds__(256) void k_align_cloud(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, int n,
                                                    float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz, float tx, float ty, float tz,
                                                    float i00, float i01, float i02, float i10, float i11, float i12, float i20, float i21, float i22) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float a = sx[i], b = sy[i], c = sz[i];
        {
#pragma clang fp contract(off)
            ox[i] = ((a * i00 + b * i10) + c * i20) - tx;
            oy[i] = ((a * i01 + b * i11) + c * i21) - ty;
            oz[i] = ((a * i02 + b * i12) + c * i22) - tz;
        }
    }
}

// getQueue (src/simpleMapMaker.cpp:43-50): oldest row first
__global__ __launch_bounds__(256) void k_map_unroll(const float* __restrict__ qx, const float* __restrict__ qy, const float* __restrict__ qz, int cap, int pos,
                                                   int filled, int rows, float* __restrict__ out, int ld) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += gridDim.x * blockDim.x) {
        const int s = filled ? (pos + i) % cap : i;
        out[i] = qx[s]; out[ld + i] = qy[s]; out[2 * (size_t)ld + i] = qz[s];
    }
}

// utils::R (src/utils.cpp:144-152) in host float arithmetic, as the nodes evaluate it (odometry.cpp:85)
void euler_R_host(float phi, float theta, float psi, float* R) {
    const float cph = std::cos(phi), sph = std::sin(phi), cth = std::cos(theta), sth = std::sin(theta), cps = std::cos(psi), sps = std::sin(psi);
    R[0] = cth * cps;  R[1] = sps * cph + sph * sth * cps;  R[2] = sph * sps - sth * cph * cps;
    R[3] = -sps * cth; R[4] = cph * cps - sph * sth * sps;  R[5] = sph * cps + sth * sps * cph;
    R[6] = sth;        R[7] = -sph * cth;                   R[8] = cph * cth;
}

// MatrixXf::inverse() of a dynamic matrix goes through PartialPivLU (Eigen/src/LU/InverseImpl.h): LU with row pivoting,
// then the two triangular solves against the permuted identity.
void inverse3_lu(const float* A, float* inv) {
    float lu[9]; std::memcpy(lu, A, sizeof(lu));
    int perm[3] = {0, 1, 2};
    for (int k = 0; k < 3; k++) {
        int piv = k; float best = std::fabs(lu[k * 3 + k]);
        for (int r = k + 1; r < 3; r++) if (std::fabs(lu[r * 3 + k]) > best) { best = std::fabs(lu[r * 3 + k]); piv = r; }
        if (piv != k) { for (int c = 0; c < 3; c++) std::swap(lu[k * 3 + c], lu[piv * 3 + c]); std::swap(perm[k], perm[piv]); }
        if (lu[k * 3 + k] == 0.f) continue;
        for (int r = k + 1; r < 3; r++) {
            lu[r * 3 + k] /= lu[k * 3 + k];
            for (int c = k + 1; c < 3; c++) lu[r * 3 + c] -= lu[r * 3 + k] * lu[k * 3 + c];
        }
    }
    for (int col = 0; col < 3; col++) {
        float b[3];
        for (int r = 0; r < 3; r++) b[r] = (perm[r] == col) ? 1.f : 0.f;
        for (int r = 1; r < 3; r++) for (int c = 0; c < r; c++) b[r] -= lu[r * 3 + c] * b[c];
        for (int r = 2; r >= 0; r--) { for (int c = r + 1; c < 3; c++) b[r] -= lu[r * 3 + c] * b[c]; b[r] /= lu[r * 3 + r]; }
        for (int r = 0; r < 3; r++) inv[r * 3 + col] = b[r];
    }
}

// Eigen::Quaternionf(Matrix3f) (Eigen/src/Geometry/Quaternion.h, Shoemake's method); q = x, y, z, w
void quat_of(const float* P /* 4x4 row-major */, float q[4]) {
    const float m00 = P[0], m01 = P[1], m02 = P[2], m10 = P[4], m11 = P[5], m12 = P[6], m20 = P[8], m21 = P[9], m22 = P[10];
    const float m[3][3] = {{m00, m01, m02}, {m10, m11, m12}, {m20, m21, m22}};
    float t = m00 + m11 + m22;
    if (t > 0.f) {
        t = std::sqrt(t + 1.0f); q[3] = 0.5f * t; t = 0.5f / t;
        q[0] = (m21 - m12) * t; q[1] = (m02 - m20) * t; q[2] = (m10 - m01) * t;
    } else {
        int i = 0;
        if (m11 > m00) i = 1;
        if (m22 > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0f);
        q[i] = 0.5f * t; t = 0.5f / t;
        q[3] = (m[k][j] - m[j][k]) * t; q[j] = (m[j][i] + m[i][j]) * t; q[k] = (m[k][i] + m[i][k]) * t;
    }
}

}  // namespace

void icet_ctx_set_stream(icet_ctx* c, hipStream_t s);      // icet_capi.hip (internal)
void icet_ctx_set_prologue(icet_ctx* c, hipError_t (*fn)(void*, hipStream_t), void* user, int64_t key);      // icet_capi.hip (internal)
void icet_ctx_set_done_flag(icet_ctx* c, int32_t* pinned_word);      // icet_capi.hip (internal)

// The helper thread of a pipelined node: it ENQUEUES the keyframe builds (icet_keyframe_device_n on the context the build goes into, then the event that says it is
// done) while the calling thread enqueues the frame's loop.  A build is ~20 launches or one graph launch of 20 nodes -- 35 to 140 us of host time that used to sit between
// the loop's launch and the build's start, so that the build of frame k ran into frame k + 1, whose loop needs it.  One job at a time in order; the calling thread waits
// for "idle" before it touches anything the helper may be using (kf_wait_idle).
struct FilterLaunch { const FrameDesc* fd; float min_range; int32_t* counts; int32_t* bases; int n_blocks; int32_t* d_cnt; int32_t* h_cnt; float* o; int64_t ld_o; };      // one range filter's three launches
namespace { hipError_t filter_prologue(void* user, hipStream_t st); }
// (f2_ev set: the build is preceded by the keyframe side's own range filter of the raw frame -- the one-launch frame of push_frame -- and f2_ev says when that has read the frame)
struct KfJob { icet_ctx* ctx; icet_params sp; icet_dev_scan b; const int32_t* d_cnt; hipStream_t sk; hipEvent_t done_ev; FilterLaunch f2{}; hipEvent_t f2_ev = nullptr; };
struct KfWorker {
    std::thread th; std::mutex m; std::condition_variable cv; std::deque<KfJob> q;
    long posted = 0, done = 0; bool stop = false; icet_status status = ICET_OK; std::string err; int device = 0;
};

struct icet_node {
    // Keyframe pipelining: scan 2 of frame k is scan 1 of frame k + 1, so the keyframe of a scan is built the
    // moment the scan arrives, on the OTHER of two contexts / streams, while the Gauss-Newton loop of the current pair iterates; the
    // frame-to-pose critical path is then range filter + loop.  kf[owner] holds the parked keyframe of the previous scan.
    // Both contexts are the node's own: a keyframe parked in the caller's context would be lost to the caller's next solve on it.
    icet_ctx* kf[2] = {nullptr, nullptr}; int owner = 0; bool pipelined = false;
    icet_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    int device = 0;
    icet_node_params p{};
    std::string err;
    bool initialized = false;
    // previous / current filtered scan (column-major, ld = cap rounded to 64)
    float* d_scan[2] = {nullptr, nullptr}; int64_t cap_scan[2] = {0, 0}; int64_t n_scan[2] = {0, 0}; int64_t ld_scan[2] = {0, 0};
    int prev = 0;
    float* d_stage = nullptr; int64_t cap_stage = 0;              // host scans land here first
    int32_t* d_counts = nullptr; int32_t* d_bases = nullptr; int cap_blocks = 0;
    int32_t* d_nkept = nullptr; int32_t* h_nkept = nullptr;       // d_nkept: TWO counters, one per scan buffer (a frame's count is still read by the keyframe build that runs into the next frame)
    float* d_x0 = nullptr; float* d_out = nullptr; float* h_out = nullptr; float* h_x0 = nullptr;
    float X0[6] = {0, 0, 0, 0, 0, 0};
    float pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::mt19937 gen;                                              // default seed: simpleMapMaker.cpp:258
    icet_shuffle::FastMt fgen; bool fast_shuffle = false;         // the same stream written out (icet_shuffle.h); used when it reproduced std::shuffle on this C++ library at creation
    std::vector<std::size_t> indices;
    float* d_map = nullptr; int64_t map_pos = 0; bool map_filled = false;
    int32_t* d_idx = nullptr; int32_t* h_idx = nullptr;
    float* d_aligned = nullptr; int64_t cap_aligned = 0, n_aligned = 0, ld_aligned = 0;     // scanMatcher.cpp:76
    std::vector<float> snail;                                                               // scanMatcher.cpp:27-28,79-84: rows x 3 row-major, host
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // [5]: start of the loop on the owner's stream (pipelined)
    bool timing_valid = false, timed_map = false;
    KfWorker* kw = nullptr;                                       // started with the first build that goes through it
    hipEvent_t ev_kfdone[2] = {nullptr, nullptr}; bool kf_built[2] = {false, false};          // per context: its last keyframe build has been enqueued / the event behind it
    // The one-launch frame (round 6, push_frame): the loop's context captures the range filter in front of its loop (FilterLaunch = what its hook enqueues), and the
    // keyframe build of the same scan runs on the other context's stream behind a filter of ITS OWN into a second buffer -- no dependency between the two streams inside a frame.
    FrameDesc* h_frame = nullptr;                                 // pinned, [2]: by scan buffer
    FilterLaunch fl[2];
    float* d_scan_kf[2] = {nullptr, nullptr}; int64_t cap_scan_kf[2] = {0, 0};
    int32_t* d_counts_kf = nullptr; int32_t* d_bases_kf = nullptr; int cap_blocks_kf = 0; int32_t* d_nkept_kf = nullptr;
    hipEvent_t ev_f2 = nullptr;                                   // the keyframe side's filter has read the caller's frame
    int32_t* h_done = nullptr;                                    // pinned, coherent: the frame's last solve stores 1 here behind its results (the host watches it instead of synchronising the stream)
#ifdef ICET_DIAG_ENV
    double tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long tr_n = 0, tr_seen = 0;       // ICET_NODE_TRACE: host microseconds of push_frame by section, summed (printed by icet_node_destroy)
#endif
};

namespace {

#define NCHK(nd, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (nd)->err = std::string(#call) + ": " + hipGetErrorString(e_); \
    return e_ == hipErrorOutOfMemory ? ICET_ERR_NOMEM : ICET_ERR_HIP; } } while (0)

bool kf_worker_start(icet_node* nd) {
    if (nd->kw) return true;
    try {
        KfWorker* w = new KfWorker(); w->device = nd->device;
        w->th = std::thread([w]() {
            (void)hipSetDevice(w->device);
            for (;;) {
                KfJob j;
                { std::unique_lock<std::mutex> lk(w->m); w->cv.wait(lk, [&] { return w->stop || !w->q.empty(); }); if (w->q.empty()) return; j = w->q.front(); w->q.pop_front(); }
                icet_status s = ICET_OK; std::string e;
                bool skip; { std::lock_guard<std::mutex> lk(w->m); skip = w->status != ICET_OK; }      // after a failure the jobs behind it are only counted
                if (!skip && j.f2_ev) {
                    if (filter_prologue(&j.f2, j.sk) != hipSuccess || hipEventRecord(j.f2_ev, j.sk) != hipSuccess) { s = ICET_ERR_HIP; e = "range filter in front of the keyframe build"; }
                }
                if (!skip && s == ICET_OK) {
                    s = icet_keyframe_device_n(j.ctx, &j.sp, 1, &j.b, j.d_cnt);
                    if (s != ICET_OK) e = icet_last_error(j.ctx);
                    else if (hipEventRecord(j.done_ev, j.sk) != hipSuccess) { s = ICET_ERR_HIP; e = "hipEventRecord(keyframe build done)"; }
                }
                { std::lock_guard<std::mutex> lk(w->m); if (s != ICET_OK && w->status == ICET_OK) { w->status = s; w->err = e; } w->done++; }
                w->cv.notify_all();
            }
        });
        nd->kw = w;
        return true;
    } catch (...) { return false; }                               // no thread to be had: the caller enqueues the builds itself
}
void kf_post(icet_node* nd, const KfJob& j) { { std::lock_guard<std::mutex> lk(nd->kw->m); nd->kw->q.push_back(j); nd->kw->posted++; } nd->kw->cv.notify_all(); }
// Blocks until the helper has nothing left to enqueue; a failure of one of its jobs is reported (once) here, to whoever waits next.
icet_status kf_wait_idle(icet_node* nd) {
    if (!nd->kw) return ICET_OK;
    std::unique_lock<std::mutex> lk(nd->kw->m);
    nd->kw->cv.wait(lk, [&] { return nd->kw->done >= nd->kw->posted; });
    const icet_status s = nd->kw->status;
    if (s != ICET_OK) { nd->err = nd->kw->err; nd->kw->status = ICET_OK; nd->kw->err.clear(); }
    return s;
}
void kf_worker_stop(icet_node* nd) {
    if (!nd->kw) return;
    { std::lock_guard<std::mutex> lk(nd->kw->m); nd->kw->stop = true; }
    nd->kw->cv.notify_all();
    if (nd->kw->th.joinable()) nd->kw->th.join();
    delete nd->kw; nd->kw = nullptr;
}

icet_status ensure_scan(icet_node* nd, int which, int64_t n) {
    if (n <= nd->cap_scan[which]) return ICET_OK;
    NCHK(nd, hipDeviceSynchronize());                             // both streams of a pipelined node may still read the buffer
    if (nd->d_scan[which]) { NCHK(nd, hipFree(nd->d_scan[which])); nd->d_scan[which] = nullptr; }
    const int64_t cap = (n + n / 8 + 63) / 64 * 64;
    NCHK(nd, hipMalloc(reinterpret_cast<void**>(&nd->d_scan[which]), sizeof(float) * 3 * (size_t)cap));
    nd->cap_scan[which] = cap;
    return ICET_OK;
}

// The frame's down-sample indices into pinned h_idx (simpleMapMaker.cpp:147-158: iota, std::shuffle with the node's generator, the first map_downsample entries): returns
// how many.  Only the first entries of the shuffled vector are ever used, so only they are tracked while the generator makes its n - 1 draws (icet_shuffle.h).
int draw_downsample(icet_node* nd, int64_t nk) {
    if (nd->fast_shuffle) icet_shuffle::head_of_shuffled_iota((std::size_t)nk, (std::size_t)nd->p.map_downsample, nd->fgen, nd->indices);
    else icet_shuffle::head_of_shuffled_iota((std::size_t)nk, (std::size_t)nd->p.map_downsample, nd->gen, nd->indices);
    const int m = (int)nd->indices.size();
    for (int i = 0; i < m; i++) nd->h_idx[i] = (int32_t)nd->indices[i];
    return m;
}

// ---- the one-launch frame (round 6) ----
hipError_t filter_prologue(void* user, hipStream_t st) {          // the hook icet_register_device_n runs in front of its loop (icet_ctx_set_prologue)
    const FilterLaunch& f = *static_cast<const FilterLaunch*>(user);
    k_range_count<<<f.n_blocks, kFB, 0, st>>>(nullptr, nullptr, nullptr, 0, f.min_range, f.counts, f.fd);
    if (f.n_blocks <= 1024) {                                     // the scan folded into the scatter (every block reads at most 4 KB of counts)
        k_range_scatter<<<f.n_blocks, kFB, 0, st>>>(nullptr, nullptr, nullptr, 0, f.min_range, nullptr, f.o, f.o + f.ld_o, f.o + 2 * f.ld_o, f.fd, f.counts, f.n_blocks, f.d_cnt, f.h_cnt);
        return hipGetLastError();
    }
    k_range_scan<<<1, kFB, 0, st>>>(f.counts, f.bases, f.n_blocks, f.d_cnt, f.h_cnt);
    k_range_scatter<<<f.n_blocks, kFB, 0, st>>>(nullptr, nullptr, nullptr, 0, f.min_range, f.bases, f.o, f.o + f.ld_o, f.o + 2 * f.ld_o, f.fd);
    return hipGetLastError();
}
int64_t filter_key(const FilterLaunch& f) {            // everything the hook's launches depend on (FNV-1a): part of the loop's graph key
    const uint64_t v[] = {(uint64_t)(uintptr_t)f.fd, (uint64_t)__builtin_bit_cast(uint32_t, f.min_range), (uint64_t)(uintptr_t)f.counts, (uint64_t)(uintptr_t)f.bases, (uint64_t)f.n_blocks,
                          (uint64_t)(uintptr_t)f.d_cnt, (uint64_t)(uintptr_t)f.h_cnt, (uint64_t)(uintptr_t)f.o, (uint64_t)f.ld_o};
    uint64_t h = 1469598103934665603ull;
    for (uint64_t x : v) for (int b = 0; b < 8; b++) { h ^= (x >> (8 * b)) & 0xffu; h *= 1099511628211ull; }
    return (int64_t)(h | 1ull);
}
// the keyframe side's own filtered copy of scan `which`, its block counters and its row counters
icet_status ensure_kf_side(icet_node* nd, int which, int64_t n) {
    if (!nd->h_frame) { NCHK(nd, hipHostMalloc(reinterpret_cast<void**>(&nd->h_frame), 2 * sizeof(FrameDesc))); std::memset(nd->h_frame, 0, 2 * sizeof(FrameDesc)); }
    if (!nd->d_nkept_kf) NCHK(nd, hipMalloc(reinterpret_cast<void**>(&nd->d_nkept_kf), 2 * sizeof(int32_t)));
    if (!nd->ev_f2) NCHK(nd, hipEventCreateWithFlags(&nd->ev_f2, hipEventDisableTiming));
    if (!nd->h_done) { NCHK(nd, hipHostMalloc(reinterpret_cast<void**>(&nd->h_done), sizeof(int32_t), hipHostMallocCoherent)); *nd->h_done = 0; }
    if (n > nd->cap_scan_kf[which]) {
        NCHK(nd, hipDeviceSynchronize());
        if (nd->d_scan_kf[which]) { NCHK(nd, hipFree(nd->d_scan_kf[which])); nd->d_scan_kf[which] = nullptr; }
        const int64_t cap = (n + n / 8 + 63) / 64 * 64;
        NCHK(nd, hipMalloc(reinterpret_cast<void**>(&nd->d_scan_kf[which]), sizeof(float) * 3 * (size_t)cap));
        nd->cap_scan_kf[which] = cap;
    }
    const int nbl = (int)((nd->cap_scan_kf[which] + kFB * kFRows - 1) / (kFB * kFRows));
    if (nbl > nd->cap_blocks_kf) {
        NCHK(nd, hipDeviceSynchronize());
        if (nd->d_counts_kf) NCHK(nd, hipFree(nd->d_counts_kf));
        if (nd->d_bases_kf) NCHK(nd, hipFree(nd->d_bases_kf));
        nd->d_counts_kf = nd->d_bases_kf = nullptr;
        NCHK(nd, hipMalloc(reinterpret_cast<void**>(&nd->d_counts_kf), sizeof(int32_t) * nbl));
        NCHK(nd, hipMalloc(reinterpret_cast<void**>(&nd->d_bases_kf), sizeof(int32_t) * nbl));
        nd->cap_blocks_kf = nbl;
    }
    return ICET_OK;
}

// One frame with the raw scan already in HBM (column-major, ld).
icet_status push_frame(icet_node* nd, const float* d_scan, int64_t n, int64_t ld, icet_node_result* res) {
#ifdef ICET_DIAG_ENV
    static const bool trace_on = getenv("ICET_NODE_TRACE") != nullptr;
    auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tq[6] = {0, 0, 0, 0, 0, 0}; tq[0] = now_us();
#define ICET_TR(k) do { if (trace_on) tq[k] = now_us(); } while (0)
#else
#define ICET_TR(k) do { } while (0)
#endif
    std::memset(res, 0, sizeof(*res));
    { const icet_status hs = kf_wait_idle(nd); if (hs != ICET_OK) return hs; }      // the previous frame's keyframe build has been enqueued (or says why not)
    hipStream_t st = nd->stream;
    const int cur = nd->prev ^ 1;
    nd->timing_valid = false;
    if (!nd->initialized) {
        // odometry.cpp:46-52: the first cloud is stored as it is (no range filter) and nothing is solved
        icet_status s = ensure_scan(nd, nd->prev, n); if (s != ICET_OK) return s;
        const int64_t l = nd->cap_scan[nd->prev];
        if (n) NCHK(nd, hipMemcpy2DAsync(nd->d_scan[nd->prev], l * sizeof(float), d_scan, ld * sizeof(float), n * sizeof(float), 3, hipMemcpyDeviceToDevice, st));
        NCHK(nd, hipStreamSynchronize(st));
        nd->n_scan[nd->prev] = n; nd->ld_scan[nd->prev] = l;
        if (nd->pipelined) {
            icet_dev_scan a{nd->d_scan[nd->prev], n, l};
            icet_params sp = nd->p.solve; sp.flags = (nd->p.flags & ICET_NODE_DOUBLE_W) ? ICET_FLAG_DOUBLE_W : ICET_FLAG_NONE;
            nd->owner = 0;
            icet_status ks = icet_keyframe_device(nd->kf[0], &sp, 1, &a);
            if (ks != ICET_OK) { nd->err = icet_last_error(nd->kf[0]); return ks; }
            NCHK(nd, hipEventRecord(nd->ev_kfdone[0], reinterpret_cast<hipStream_t>(icet_stream(nd->kf[0])))); nd->kf_built[0] = true; nd->kf_built[1] = false;
        }
        nd->initialized = true;
        res->solved = 0; res->n_kept = n;
        std::memcpy(res->pose, nd->pose, sizeof(nd->pose)); quat_of(nd->pose, res->quat);
        res->map_rows = nd->map_filled ? nd->p.map_capacity : nd->map_pos;
        return ICET_OK;
    }
    // ---- range filter: stable compaction into the "current" buffer (odometry.cpp:57-70) ----
    icet_status s = ensure_scan(nd, cur, n); if (s != ICET_OK) return s;
    const int64_t lcur = nd->cap_scan[cur];
    // Whoever needs the kept-row count on the HOST before the solve can be enqueued (the map maker's shuffle runs over exactly that many
    // indices; the aligned cloud and the unpipelined solve are sized by it) waits for the filter here.  The pipelined odometry frame does
    // not: the solve's two halves take the unfiltered row count as an upper bound for their launch geometry and read the actual count on
    // the device (icet_register_device_n / icet_keyframe_device_n), so the whole frame is enqueued without a host round trip in the middle.
    const bool fast = nd->pipelined && nd->p.map_capacity == 0 && n > 0 &&
                      !(nd->p.flags & (ICET_NODE_NO_RANGE_FILTER | ICET_NODE_ALIGNED_CLOUD | ICET_NODE_SNAIL_TRAIL));
#ifdef ICET_DIAG_ENV      /* the A/B switches of round 5's stream experiments: experiment builds only (make EXTRA=-DICET_DIAG_ENV); the shipped library never reads the environment */
    static const bool loop_on_filter_stream = getenv("ICET_NODE_LOOP_OWN_STREAM") == nullptr;
#else
    constexpr bool loop_on_filter_stream = true;
#endif
    // ... and since round 6 such a frame is ONE launch on the critical path: the loop's context captures the filter in front of its loop (one graph: the loop's first kernel
    // used to start 30-40 us after the filter's last one), and the keyframe build of this scan, on the other stream, filters the raw frame once more for itself instead of
    // waiting for this stream's filter (a dependency between two streams costs more than 20 us of duplicated filter).  ICET_NODE_TIME_PHASES keeps the phases apart.
    // The map maker's frame takes the same shape (round 6): its loop does not need the kept-row count on the host either -- only the down-sample shuffle does, and the
    // host gets the count by watching the pinned word the filter's kernel stores it into, a few microseconds into the frame, while the loop runs.
    const bool map_ok = nd->pipelined && nd->p.map_capacity > 0 && n > 0 && !(nd->p.flags & (ICET_NODE_NO_RANGE_FILTER | ICET_NODE_ALIGNED_CLOUD | ICET_NODE_SNAIL_TRAIL));
    const bool fused = (fast || map_ok) && loop_on_filter_stream && nd->p.solve.runlen > 0 && !(nd->p.flags & ICET_NODE_TIME_PHASES);
    const bool dev_count = fast || fused;                         // the solve's halves read the row count on the device
    const int n_blocks = fused ? (int)((lcur + kFB * kFRows - 1) / (kFB * kFRows)) : (int)((n + kFB * kFRows - 1) / (kFB * kFRows));      // fused: by CAPACITY (the launch does not change with n)
    if (n_blocks > nd->cap_blocks) {
        NCHK(nd, hipDeviceSynchronize());
        if (nd->d_counts) NCHK(nd, hipFree(nd->d_counts));
        if (nd->d_bases) NCHK(nd, hipFree(nd->d_bases));
        nd->d_counts = nd->d_bases = nullptr;
        NCHK(nd, hipMalloc(reinterpret_cast<void**>(&nd->d_counts), sizeof(int32_t) * n_blocks));
        NCHK(nd, hipMalloc(reinterpret_cast<void**>(&nd->d_bases), sizeof(int32_t) * n_blocks));
        nd->cap_blocks = n_blocks;
    }
    if (fused) { s = ensure_kf_side(nd, cur, n); if (s != ICET_OK) return s; }
    int32_t* d_cnt = nd->d_nkept + cur;
    if (!fused) NCHK(nd, hipEventRecord(nd->ev[0], st));         // (the one-launch frame records no timing events: each costs the host 5-10 us IN FRONT of the launch)
    if (nd->p.flags & ICET_NODE_NO_RANGE_FILTER) {               // scanMatcher.cpp:44: the cloud goes to the constructor as it is
        if (n) NCHK(nd, hipMemcpy2DAsync(nd->d_scan[cur], lcur * sizeof(float), d_scan, ld * sizeof(float), n * sizeof(float), 3, hipMemcpyDeviceToDevice, st));
        *nd->h_nkept = (int32_t)n;
    } else if (fused) {
        nd->h_frame[cur] = FrameDesc{d_scan, (int32_t)n, (int32_t)ld};                          // what both filters of this frame read
        *static_cast<volatile int32_t*>(nd->h_nkept) = -1;                                      // ("not yet": the map maker's host side watches this word)
        nd->fl[cur] = FilterLaunch{nd->h_frame + cur, nd->p.min_range, nd->d_counts, nd->d_bases, n_blocks, d_cnt, nd->h_nkept, nd->d_scan[cur], lcur};
    } else if (n > 0) {
        const float *x = d_scan, *y = d_scan + ld, *z = d_scan + 2 * ld;
        float* o = nd->d_scan[cur];
        k_range_count<<<n_blocks, kFB, 0, st>>>(x, y, z, (int)n, nd->p.min_range, nd->d_counts);
        // the kept-row count goes to the host by the kernel's own store into pinned memory: a hipMemcpyAsync of four bytes costs the stream ~20 us on this part (round 5's
        // burst timeline), and the frame's other small copy -- the 48 result floats -- went the same way in round 6 (below)
        k_range_scan<<<1, kFB, 0, st>>>(nd->d_counts, nd->d_bases, n_blocks, d_cnt, nd->h_nkept);
        k_range_scatter<<<n_blocks, kFB, 0, st>>>(x, y, z, (int)n, nd->p.min_range, nd->d_bases, o, o + lcur, o + 2 * lcur);
        NCHK(nd, hipGetLastError());
        if (dev_count) NCHK(nd, hipEventRecord(nd->ev[1], st));   // what the solve streams wait for: the filtered scan
    } else {
        *nd->h_nkept = 0;
    }
    if (!dev_count) NCHK(nd, hipEventRecord(nd->ev[1], st));
    if (!dev_count) NCHK(nd, hipStreamSynchronize(st));           // the solve's launch geometry needs the row count
    // fast: an upper bound until the end-of-frame synchronisation -- the buffer's CAPACITY, which does not change from frame to frame, so that the
    // two halves of the solve see the same launch key every frame and replay their captured graphs (one hipGraphLaunch each instead of ~35 launches)
    int64_t nk = dev_count ? lcur : (int64_t)*nd->h_nkept;
    nd->n_scan[cur] = nk; nd->ld_scan[cur] = lcur;
    // ---- ICET it(prev, cur, runlen, X0, bins_phi, bins_theta, n, thresh, buff)  (odometry.cpp:76) ----
    // The down-sample indices of this frame (simpleMapMaker.cpp:147-158) depend only on the row count and on the node's RNG stream,
    // and Fisher-Yates over ~10^5 indices costs about as much host time as the solve costs device time: a helper thread shuffles
    // while this one enqueues the solve.  (The previous frame's map kernel, which read d_idx, finished before the row-count sync.)
    int m_map = 0;
    bool flip_owner = false;
    std::future<int> shuffle;
    if (nd->p.map_capacity > 0 && !fused) {
        shuffle = std::async(std::launch::async, [nd, nk]() { return draw_downsample(nd, nk); });
    }
    std::memcpy(nd->h_x0, nd->X0, sizeof(nd->X0));
    icet_dev_scan a{nd->d_scan[nd->prev], nd->n_scan[nd->prev], nd->ld_scan[nd->prev]}, b{nd->d_scan[cur], nk, lcur};
    icet_params sp = nd->p.solve; sp.flags = (nd->p.flags & ICET_NODE_DOUBLE_W) ? ICET_FLAG_DOUBLE_W : ICET_FLAG_NONE;
    hipStream_t so = st;                                          // the stream the result arrives on
    if (nd->pipelined) {
        // the loop against the keyframe parked one frame ago (the host has synchronised the filter's stream above), then -- behind
        // it in host order, beside it on the device -- the keyframe of THIS scan on the other context, for the next frame
        icet_ctx* own = nd->kf[nd->owner]; icet_ctx* oth = nd->kf[nd->owner ^ 1];
        hipStream_t s_own = reinterpret_cast<hipStream_t>(icet_stream(own)), s_oth = reinterpret_cast<hipStream_t>(icet_stream(oth));
        // The loop runs on the FILTER's stream: a dependency between two streams costs ~60 us on this part before the waiting queue starts (measured on the device
        // timeline: filter end -> first loop kernel), and filter -> loop -> result is the frame's critical path.  The context keeps its own stream for its keyframe
        // builds; the build this loop needs (previous frame, that stream) finished long ago as a rule -- an event says so.
        so = loop_on_filter_stream ? st : s_own;
        if (loop_on_filter_stream) {
            // (the build is done, as a rule, by the time the next frame arrives: asking costs the host a microsecond, a wait command in the stream five to ten)
            if (nd->kf_built[nd->owner] && hipEventQuery(nd->ev_kfdone[nd->owner]) != hipSuccess) { (void)hipGetLastError(); NCHK(nd, hipStreamWaitEvent(st, nd->ev_kfdone[nd->owner], 0)); }
            if (dev_count && !fused) NCHK(nd, hipStreamWaitEvent(s_oth, nd->ev[1], 0));
            icet_ctx_set_stream(own, st);
        } else if (dev_count) {                                   // nobody waited for the filter: both solve streams do
            NCHK(nd, hipStreamWaitEvent(so, nd->ev[1], 0));
            NCHK(nd, hipStreamWaitEvent(s_oth, nd->ev[1], 0));
        }
        if (!fused) NCHK(nd, hipEventRecord(nd->ev[5], so));
        // The keyframe side of a one-launch frame is enqueued by the node's HELPER thread, posted BEFORE this thread launches the loop's graph: its
        // ~45 us of enqueueing (three filter launches, an event, a graph of 18 kernels) then run beside the loop's launch and not behind it -- the build, not the loop,
        // is what the next frame waits for, and it used to start 60 us into the frame.  (Round 5, frame in phases: measured, no gain, 4.83-4.93 k frames/s with, 4.82-4.92 k
        // without.)  A failure of the build is reported by the next call that waits for the helper.
        const bool via_helper = fused && !(nd->p.flags & ICET_NODE_SERIAL_ENQUEUE) && kf_worker_start(nd);
        const int64_t lkf = fused ? nd->cap_scan_kf[cur] : 0;
        const FilterLaunch f2{nd->h_frame ? nd->h_frame + cur : nullptr, nd->p.min_range, nd->d_counts_kf, nd->d_bases_kf, (int)((lkf + kFB * kFRows - 1) / (kFB * kFRows)),
                              nd->d_nkept_kf ? nd->d_nkept_kf + cur : nullptr, nullptr, fused ? nd->d_scan_kf[cur] : nullptr, lkf};      // the other stream's own filtered copy of this scan
        if (via_helper) {
            KfJob j{oth, sp, icet_dev_scan{nd->d_scan_kf[cur], lkf, lkf}, nd->d_nkept_kf + cur, s_oth, nd->ev_kfdone[nd->owner ^ 1]};
            j.f2 = f2; j.f2_ev = nd->ev_f2;
            kf_post(nd, j); nd->kf_built[nd->owner ^ 1] = true;
        }
        // X0 is read from pinned host memory by the kernel itself (no H2D command), and k_gn_solve stores the 48 result floats of every iteration straight into pinned
        // host memory (the same pointer is valid on the device): no copy command in the frame.  (runlen 0 is a memset + copy of X0 on the device side: it keeps the
        // device buffer and its copy.)
        float* out_dev = sp.runlen > 0 ? nd->h_out : nd->d_out;
        ICET_TR(1);
        if (fused) { *static_cast<volatile int32_t*>(nd->h_done) = 0; icet_ctx_set_done_flag(own, nd->h_done); }
        if (fused) icet_ctx_set_prologue(own, filter_prologue, &nd->fl[cur], filter_key(nd->fl[cur]));        // filter + loop: one graph, one launch
        s = icet_register_device_n(own, &sp, 1, &b, dev_count ? d_cnt : nullptr, nd->h_x0, out_dev);
        if (fused) { icet_ctx_set_prologue(own, nullptr, nullptr, 0); icet_ctx_set_done_flag(own, nullptr); }
        ICET_TR(2);
        icet_ctx_set_stream(own, s_own);
        if (s != ICET_OK) { nd->err = icet_last_error(own); if (via_helper) (void)kf_wait_idle(nd); return s; }
        if (out_dev != nd->h_out) NCHK(nd, hipMemcpyAsync(nd->h_out, nd->d_out, sizeof(float) * 48, hipMemcpyDeviceToHost, so));
        if (!fused) NCHK(nd, hipEventRecord(nd->ev[2], so));
        if (!via_helper) {
            if (fused) {
                // (the same kept rows in the same order: the same keyframe bits), then the build from it
                NCHK(nd, filter_prologue(const_cast<FilterLaunch*>(&f2), s_oth));
                NCHK(nd, hipEventRecord(nd->ev_f2, s_oth));       // the caller's frame has been read (waited for before this push returns)
                icet_dev_scan bk{nd->d_scan_kf[cur], lkf, lkf};
                s = icet_keyframe_device_n(oth, &sp, 1, &bk, nd->d_nkept_kf + cur);
            } else {
                s = icet_keyframe_device_n(oth, &sp, 1, &b, dev_count ? d_cnt : nullptr);
            }
            if (s != ICET_OK) { nd->err = icet_last_error(oth); return s; }
            NCHK(nd, hipEventRecord(nd->ev_kfdone[nd->owner ^ 1], s_oth)); nd->kf_built[nd->owner ^ 1] = true;
        }
        flip_owner = true;                                        // committed together with nd->prev once the frame has succeeded
    } else {
        float* out_dev = sp.runlen > 0 ? nd->h_out : nd->d_out;
        s = icet_solve_batch_device(nd->ctx, &sp, 1, &a, &b, nd->h_x0, out_dev);
        if (s != ICET_OK) { nd->err = icet_last_error(nd->ctx); return s; }
        if (out_dev != nd->h_out) NCHK(nd, hipMemcpyAsync(nd->h_out, nd->d_out, sizeof(float) * 48, hipMemcpyDeviceToHost, st));
        NCHK(nd, hipEventRecord(nd->ev[2], st));
    }
    if (fused && nd->p.map_capacity > 0) {
        // the frame is in flight; its filter's count lands in pinned memory ~25 us in (k_range_scan's own store), the loop runs for another ~230: the down-sample
        // shuffle (simpleMapMaker.cpp:147-158: std::shuffle over exactly that many indices with the node's RNG stream) runs here, on this thread, beside it
        volatile int32_t* hc = nd->h_nkept;
        for (long spins = 1; *hc < 0; spins++)
            if ((spins & 4095) == 0 && hipStreamQuery(st) != hipErrorNotReady) break;             // (finished or failed without a count: decided below)
        (void)hipGetLastError();
        if (*hc < 0) NCHK(nd, hipStreamSynchronize(st));
        if (*hc < 0) { nd->err = "the range filter's row count did not arrive"; if (nd->kw) (void)kf_wait_idle(nd); return ICET_ERR_HIP; }
        m_map = draw_downsample(nd, (int64_t)*hc);                                                // into pinned h_idx: the map kernel reads it in place (no copy command)
    }
    if (shuffle.valid()) {
        m_map = shuffle.get();
        if (m_map) NCHK(nd, hipMemcpyAsync(nd->d_idx, nd->h_idx, sizeof(int32_t) * m_map, hipMemcpyHostToDevice, st));
    }
    ICET_TR(3);
    if (so != st) NCHK(nd, hipStreamSynchronize(so));
    if (fused) {
        // the frame's last kernel stores its 48 result floats and then a 1 into pinned host memory: this thread watches that word -- hipStreamSynchronize answers
        // several microseconds after the queue has drained -- and asks the stream only now and then (a frame that failed never writes the word)
        volatile int32_t* hd = nd->h_done;
        for (long spins = 1; *hd == 0; spins++)
            if ((spins & 8191) == 0 && hipStreamQuery(st) != hipErrorNotReady) break;
        (void)hipGetLastError();
        if (*hd == 0) NCHK(nd, hipStreamSynchronize(st));
        if (*hd == 0) { nd->err = "the frame's result did not arrive"; if (nd->kw) (void)kf_wait_idle(nd); return ICET_ERR_HIP; }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    } else NCHK(nd, hipStreamSynchronize(st));
    ICET_TR(4);
    if (fused) {                                                  // the caller's frame may be released when this returns: the other stream's filter has read it (long done: it started beside this stream's)
        const icet_status hs = kf_wait_idle(nd); if (hs != ICET_OK) return hs;      // (the helper has recorded the event)
        NCHK(nd, hipEventSynchronize(nd->ev_f2));
    }
    if (dev_count) { nk = *nd->h_nkept; nd->n_scan[cur] = nk; }   // the filter's count has arrived with everything else
    float X[6];
    std::memcpy(X, nd->h_out, sizeof(X)); std::memcpy(res->pred_stds, nd->h_out + 6, sizeof(float) * 6);
    // seed for the next frame (odometry.cpp:82 / simpleMapMaker.cpp:124), then the guard (simpleMapMaker.cpp:129-137)
    for (int k = 0; k < 6; k++) nd->X0[k] = nd->p.seed_x0 ? X[k] : 0.f;
    {
        // each group is guarded only when ITS threshold is set (0 = off, include/icet_nodes.h): a caller who sets one of the two
        // must not have the other group compared against 0
        const float tt = nd->p.trans_thresh, rt = nd->p.rot_thresh;
        if ((tt > 0.f && (std::fabs(X[0]) > tt || std::fabs(X[1]) > tt || std::fabs(X[2]) > tt)) ||
            (rt > 0.f && (std::fabs(X[3]) > rt || std::fabs(X[4]) > rt || std::fabs(X[5]) > rt))) {
            for (int k = 0; k < 6; k++) X[k] = 0.f;
            res->diverged = 1;
        }
    }
    float R[9]; euler_R_host(X[3], X[4], X[5], R);
    // ---- map queue (simpleMapMaker.cpp:147-158, 34-41) ----
    nd->timed_map = false;
    if (nd->p.map_capacity > 0) {
        const int m = m_map;
        float Ri[9]; inverse3_lu(R, Ri);
        const int cap = nd->p.map_capacity;
        float* q = nd->d_map; const float* sc = nd->d_scan[cur];
        const int blocks = std::min((cap + 255) / 256, 256 * 8);
        if (!fused) NCHK(nd, hipEventRecord(nd->ev[4], st));
        k_map_add_scan<<<blocks, 256, 0, st>>>(q, q + cap, q + 2 * (size_t)cap, cap, (int)nd->map_pos, m, sc, sc + lcur, sc + 2 * lcur, fused ? nd->h_idx : nd->d_idx,
                                               X[0], X[1], X[2], Ri[0], Ri[1], Ri[2], Ri[3], Ri[4], Ri[5], Ri[6], Ri[7], Ri[8]);
        NCHK(nd, hipGetLastError());
        if (!fused) NCHK(nd, hipEventRecord(nd->ev[3], st));      // not waited for: the next push (or icet_node_map) synchronises the stream
        if (nd->map_pos + m >= cap) nd->map_filled = true;
        nd->map_pos = (nd->map_pos + m) % cap;
        nd->timed_map = !fused;
    }
    if (nd->p.flags & (ICET_NODE_ALIGNED_CLOUD | ICET_NODE_SNAIL_TRAIL)) {
        float Ri[9]; inverse3_lu(R, Ri);
        if (nd->p.flags & ICET_NODE_ALIGNED_CLOUD) {
            if (lcur > nd->cap_aligned) {
                NCHK(nd, hipStreamSynchronize(st));
                if (nd->d_aligned) { NCHK(nd, hipFree(nd->d_aligned)); nd->d_aligned = nullptr; }
                NCHK(nd, hipMalloc(reinterpret_cast<void**>(&nd->d_aligned), sizeof(float) * 3 * (size_t)lcur));
                nd->cap_aligned = lcur;
            }
            const float* sc = nd->d_scan[cur]; float* o = nd->d_aligned; const int64_t la = nd->cap_aligned;
            if (nk) k_align_cloud<<<(int)std::min<int64_t>((nk + 255) / 256, 2048), 256, 0, st>>>(sc, sc + lcur, sc + 2 * lcur, (int)nk, o, o + la, o + 2 * la, X[0], X[1], X[2],
                                                                                               Ri[0], Ri[1], Ri[2], Ri[3], Ri[4], Ri[5], Ri[6], Ri[7], Ri[8]);
            NCHK(nd, hipGetLastError());
            nd->n_aligned = nk; nd->ld_aligned = la;
        }
        if (nd->p.flags & ICET_NODE_SNAIL_TRAIL) {               // snailTrail = (snailTrail * rot_mat.inverse()).rowwise() - trans; append the origin
            for (size_t i = 0; i + 2 < nd->snail.size(); i += 3) {
                const float a = nd->snail[i], b = nd->snail[i + 1], c = nd->snail[i + 2];
                nd->snail[i] = ((a * Ri[0] + b * Ri[3]) + c * Ri[6]) - X[0];
                nd->snail[i + 1] = ((a * Ri[1] + b * Ri[4]) + c * Ri[7]) - X[1];
                nd->snail[i + 2] = ((a * Ri[2] + b * Ri[5]) + c * Ri[8]) - X[2];
            }
            nd->snail.insert(nd->snail.end(), {0.f, 0.f, 0.f});
        }
    }
    nd->prev = cur;                                               // prev_pcl_matrix = pcl_matrix (odometry.cpp:88)
    if (flip_owner) nd->owner ^= 1;                               // ... and the keyframe parked for it becomes the one the next frame registers against
    // X_homo = X_homo * X_homo_i (odometry.cpp:91-98)
    const float Hi[16] = {R[0], R[1], R[2], X[0], R[3], R[4], R[5], X[1], R[6], R[7], R[8], X[2], 0, 0, 0, 1};
    float P[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { float acc = 0.f; for (int k = 0; k < 4; k++) acc += nd->pose[r * 4 + k] * Hi[k * 4 + c]; P[r * 4 + c] = acc; }
    std::memcpy(nd->pose, P, sizeof(P));
    res->solved = 1; res->n_kept = nk;
    std::memcpy(res->X, X, sizeof(X)); std::memcpy(res->pose, P, sizeof(P)); quat_of(P, res->quat);
    res->map_rows = nd->map_filled ? nd->p.map_capacity : nd->map_pos;
    nd->timing_valid = !fused;
#ifdef ICET_DIAG_ENV
    if (trace_on && tq[1] > 0) { tq[5] = now_us(); if (++nd->tr_seen > 6) { for (int k = 0; k < 5; k++) nd->tr[k] += tq[k + 1] - tq[k]; nd->tr_n++; } }      // (steady state: the first frames capture their graphs)
#endif
    return ICET_OK;
}
#undef ICET_TR


// No exception may cross the C ABI (std::async can throw std::system_error, the shuffle's vector bad_alloc -- rethrown by get()), and
// a frame that fails half way must not leave the node's idea of "previous scan" and the parked keyframe disagreeing: on ANY failure
// the device is drained and the node drops back to "no previous scan" -- the next cloud is stored like the first one
// (odometry.cpp:46-52) and the pose chain continues from where it was.
icet_status push_device(icet_node* nd, const float* d_scan, int64_t n, int64_t ld, icet_node_result* res) {
    icet_status s;
    const bool was_initialized = nd->initialized;
    try {
        s = push_frame(nd, d_scan, n, ld, res);
    } catch (const std::bad_alloc&) { nd->err = "out of host memory"; s = ICET_ERR_NOMEM;
    } catch (const std::exception& e) { nd->err = std::string("host error: ") + e.what(); s = ICET_ERR_NOMEM;
    } catch (...) { nd->err = "host error"; s = ICET_ERR_NOMEM; }
    if (s != ICET_OK && was_initialized) {
        (void)kf_wait_idle(nd);
        (void)hipDeviceSynchronize();
        nd->initialized = false; nd->timing_valid = false;
    }
    return s;
}

}  // namespace

extern "C" {

icet_status icet_node_create(icet_ctx* ctx, const icet_node_params* p, icet_node** out) {
    if (!out) return ICET_ERR_BAD_ARG;
    *out = nullptr;
    if (!ctx || !p || p->map_capacity < 0 || p->map_downsample < 0 || (p->map_capacity > 0 && p->map_downsample > p->map_capacity) ||
        p->solve.bins_phi <= 0 || p->solve.bins_theta <= 0 || p->solve.n < 1 || p->solve.runlen < 0) return ICET_ERR_BAD_ARG;
    icet_node* nd = new (std::nothrow) icet_node();
    if (!nd) return ICET_ERR_NOMEM;
    nd->ctx = ctx; nd->p = *p; nd->stream = reinterpret_cast<hipStream_t>(icet_stream(ctx)); nd->device = icet_device(ctx);
    auto fail = [&](icet_status s) { icet_node_destroy(nd); return s; };
    if (hipSetDevice(nd->device) != hipSuccess) return fail(ICET_ERR_NO_DEVICE);
    if (hipMalloc(reinterpret_cast<void**>(&nd->d_nkept), 2 * sizeof(int32_t)) != hipSuccess || hipHostMalloc(reinterpret_cast<void**>(&nd->h_nkept), sizeof(int32_t), hipHostMallocCoherent) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&nd->d_x0), sizeof(float) * 6) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&nd->d_out), sizeof(float) * 48) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&nd->h_out), sizeof(float) * 48, hipHostMallocCoherent) != hipSuccess || hipHostMalloc(reinterpret_cast<void**>(&nd->h_x0), sizeof(float) * 6) != hipSuccess)
        return fail(ICET_ERR_NOMEM);
    if (p->map_capacity > 0) {                                    // (a few milliseconds, once per process: the written-out generator against this C++ library's std::shuffle)
        static const bool fast_ok = icet_shuffle::matches_std_shuffle() && icet_shuffle::fast_matches_std_shuffle();
        nd->fast_shuffle = fast_ok;
    }
    for (hipEvent_t& e : nd->ev) if (hipEventCreate(&e) != hipSuccess) return fail(ICET_ERR_HIP);
    for (hipEvent_t& e : nd->ev_kfdone) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(ICET_ERR_HIP);
    if (!(p->flags & ICET_NODE_NO_PIPELINE)) {
        for (icet_ctx*& k : nd->kf) { icet_status cs = icet_create(&k, nd->device, nullptr); if (cs != ICET_OK) return fail(cs); }
        nd->pipelined = true;
    }
    if (p->flags & ICET_NODE_SNAIL_TRAIL) nd->snail.assign(3, 0.f);      // scanMatcher.cpp:27-28: one row at the origin
    if (p->map_capacity > 0) {
        if (hipMalloc(reinterpret_cast<void**>(&nd->d_map), sizeof(float) * 3 * (size_t)p->map_capacity) != hipSuccess) return fail(ICET_ERR_NOMEM);
        if (hipMemset(nd->d_map, 0, sizeof(float) * 3 * (size_t)p->map_capacity) != hipSuccess) return fail(ICET_ERR_HIP);      // Eigen leaves MatrixXf(maxSize, 3) uninitialised; unfilled rows are never returned by getQueue
        const size_t m = p->map_downsample > 0 ? p->map_downsample : 1;
        if (hipMalloc(reinterpret_cast<void**>(&nd->d_idx), sizeof(int32_t) * m) != hipSuccess || hipHostMalloc(reinterpret_cast<void**>(&nd->h_idx), sizeof(int32_t) * m) != hipSuccess)
            return fail(ICET_ERR_NOMEM);
    }
    *out = nd;
    return ICET_OK;
}

icet_status icet_node_destroy(icet_node* nd) {
    if (!nd) return ICET_ERR_BAD_ARG;
#ifdef ICET_DIAG_ENV
    if (nd->tr_n) std::fprintf(stderr, "push_frame host us over %ld frames: to-register %.1f | register (loop graph launch) %.1f | keyframe side enqueue %.1f | synchronize %.1f | tail %.1f\n", nd->tr_n,
                               nd->tr[0] / nd->tr_n, nd->tr[1] / nd->tr_n, nd->tr[2] / nd->tr_n, nd->tr[3] / nd->tr_n, nd->tr[4] / nd->tr_n);
#endif
    kf_worker_stop(nd);                      // (drains what it still has to enqueue, then joins)
    (void)hipSetDevice(nd->device);
    (void)hipDeviceSynchronize();            // not the borrowed stream: the context may already be gone
    void* dp[] = {nd->d_scan[0], nd->d_scan[1], nd->d_stage, nd->d_counts, nd->d_bases, nd->d_nkept, nd->d_x0, nd->d_out, nd->d_map, nd->d_idx, nd->d_aligned,
                  nd->d_scan_kf[0], nd->d_scan_kf[1], nd->d_counts_kf, nd->d_bases_kf, nd->d_nkept_kf};
    for (void* q : dp) if (q) (void)hipFree(q);
    void* hp[] = {nd->h_nkept, nd->h_out, nd->h_x0, nd->h_idx, nd->h_frame, nd->h_done};
    for (void* q : hp) if (q) (void)hipHostFree(q);
    for (hipEvent_t e : nd->ev) if (e) (void)hipEventDestroy(e);
    if (nd->ev_f2) (void)hipEventDestroy(nd->ev_f2);
    for (hipEvent_t e : nd->ev_kfdone) if (e) (void)hipEventDestroy(e);
    for (icet_ctx* k : nd->kf) if (k) (void)icet_destroy(k);
    delete nd;
    return ICET_OK;
}

const char* icet_node_last_error(const icet_node* nd) { return nd ? nd->err.c_str() : ""; }

icet_status icet_node_push_device(icet_node* nd, const float* d_scan, int64_t n, int64_t ld, icet_node_result* res) {
    if (!nd || !res || n < 0 || ld < n || (n > 0 && !d_scan) || n >= ((int64_t)1 << 30)) return ICET_ERR_BAD_ARG;
    if (hipSetDevice(nd->device) != hipSuccess) return ICET_ERR_NO_DEVICE;
    return push_device(nd, d_scan, n, ld, res);
}

icet_status icet_node_push_many_device(icet_node* nd, const icet_dev_scan* frames, int32_t n_frames, icet_node_result* results) {
    if (!nd || n_frames < 0 || (n_frames > 0 && (!frames || !results))) return ICET_ERR_BAD_ARG;
    for (int k = 0; k < n_frames; k++)
        if (frames[k].n < 0 || frames[k].ld < frames[k].n || (frames[k].n > 0 && !frames[k].ptr) || frames[k].n >= ((int64_t)1 << 30)) return ICET_ERR_BAD_ARG;
    if (hipSetDevice(nd->device) != hipSuccess) return ICET_ERR_NO_DEVICE;
    { const icet_status hs = kf_wait_idle(nd); if (hs != ICET_OK) { (void)hipDeviceSynchronize(); nd->initialized = false; return hs; } }
    // Rounds 4-5 chained a burst's frames on the device (X0 <- X device to device, results parked in HBM, one copy at the end).  That needs two hand-overs between
    // streams per frame -- the build of frame k waits for the loop of frame k - 1 to let go of the context's tables, the loop of frame k + 1 for that build -- at 30 to
    // 60 us each on this part, and ran at 4.0 - 4.7 k frames/s; since a frame is one graph launch (push_frame, round 6) the host in the loop costs ~25 us and frame by
    // frame runs at 5.0 - 5.2 k.  The burst entry is therefore a loop over frames (no Python / FFI call per frame for the caller: that is what it still saves).
    for (int k = 0; k < n_frames; k++) { const icet_status s = push_device(nd, frames[k].ptr, frames[k].n, frames[k].ld, &results[k]); if (s != ICET_OK) return s; }
    return ICET_OK;
}

icet_status icet_node_push(icet_node* nd, const float* scan, int64_t n, int64_t ld, icet_node_result* res) {
    if (!nd || !res || n < 0 || ld < n || (n > 0 && !scan) || n >= ((int64_t)1 << 30)) return ICET_ERR_BAD_ARG;
    if (hipSetDevice(nd->device) != hipSuccess) return ICET_ERR_NO_DEVICE;
    const int64_t l = (n + 63) / 64 * 64;
    if (3 * l > nd->cap_stage) {
        NCHK(nd, hipStreamSynchronize(nd->stream));
        if (nd->d_stage) { NCHK(nd, hipFree(nd->d_stage)); nd->d_stage = nullptr; }
        NCHK(nd, hipMalloc(reinterpret_cast<void**>(&nd->d_stage), sizeof(float) * 3 * (size_t)(l + l / 8)));
        nd->cap_stage = 3 * (l + l / 8);
    }
    if (n) NCHK(nd, hipMemcpy2DAsync(nd->d_stage, l * sizeof(float), scan, ld * sizeof(float), n * sizeof(float), 3, hipMemcpyHostToDevice, nd->stream));
    return push_device(nd, nd->d_stage, n, l, res);
}

icet_status icet_node_map(icet_node* nd, float* out, int64_t ld, int64_t* rows_out) {
    if (!nd || !rows_out) return ICET_ERR_BAD_ARG;
    (void)kf_wait_idle(nd);
    const int64_t rows = nd->map_filled ? nd->p.map_capacity : nd->map_pos;
    *rows_out = rows;
    if (!out || rows == 0) return ICET_OK;
    if (ld < rows) return ICET_ERR_BAD_ARG;
    if (hipSetDevice(nd->device) != hipSuccess) return ICET_ERR_NO_DEVICE;
    float* tmp = nullptr;
    NCHK(nd, hipMalloc(reinterpret_cast<void**>(&tmp), sizeof(float) * 3 * (size_t)rows));
    const int cap = nd->p.map_capacity;
    k_map_unroll<<<std::min((int)((rows + 255) / 256), 2048), 256, 0, nd->stream>>>(nd->d_map, nd->d_map + cap, nd->d_map + 2 * (size_t)cap, cap, (int)nd->map_pos,
                                                                                     nd->map_filled ? 1 : 0, (int)rows, tmp, (int)rows);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy2DAsync(out, ld * sizeof(float), tmp, rows * sizeof(float), rows * sizeof(float), 3, hipMemcpyDeviceToHost, nd->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(nd->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) { nd->err = hipGetErrorString(e); return ICET_ERR_HIP; }
    return ICET_OK;
}

icet_status icet_node_prev_scan(icet_node* nd, float* out, int64_t ld, int64_t* rows_out) {
    if (!nd || !rows_out) return ICET_ERR_BAD_ARG;
    (void)kf_wait_idle(nd);
    const int64_t rows = nd->initialized ? nd->n_scan[nd->prev] : 0;
    *rows_out = rows;
    if (!out || rows == 0) return ICET_OK;
    if (ld < rows) return ICET_ERR_BAD_ARG;
    if (hipSetDevice(nd->device) != hipSuccess) return ICET_ERR_NO_DEVICE;
    NCHK(nd, hipMemcpy2DAsync(out, ld * sizeof(float), nd->d_scan[nd->prev], nd->ld_scan[nd->prev] * sizeof(float), rows * sizeof(float), 3, hipMemcpyDeviceToHost, nd->stream));
    NCHK(nd, hipStreamSynchronize(nd->stream));
    return ICET_OK;
}

icet_status icet_node_aligned(icet_node* nd, float* out, int64_t ld, int64_t* rows_out) {
    if (!nd || !rows_out) return ICET_ERR_BAD_ARG;
    (void)kf_wait_idle(nd);
    const int64_t rows = nd->n_aligned;
    *rows_out = rows;
    if (!out || rows == 0) return ICET_OK;
    if (ld < rows) return ICET_ERR_BAD_ARG;
    if (hipSetDevice(nd->device) != hipSuccess) return ICET_ERR_NO_DEVICE;
    NCHK(nd, hipMemcpy2DAsync(out, ld * sizeof(float), nd->d_aligned, nd->ld_aligned * sizeof(float), rows * sizeof(float), 3, hipMemcpyDeviceToHost, nd->stream));
    NCHK(nd, hipStreamSynchronize(nd->stream));
    return ICET_OK;
}

icet_status icet_node_snail_trail(icet_node* nd, float* out, int64_t ld, int64_t* rows_out) {
    if (!nd || !rows_out) return ICET_ERR_BAD_ARG;
    const int64_t rows = (int64_t)(nd->snail.size() / 3);
    *rows_out = rows;
    if (!out || rows == 0) return ICET_OK;
    if (ld < rows) return ICET_ERR_BAD_ARG;
    for (int64_t i = 0; i < rows; i++) { out[i] = nd->snail[3 * i]; out[ld + i] = nd->snail[3 * i + 1]; out[2 * ld + i] = nd->snail[3 * i + 2]; }
    return ICET_OK;
}

icet_status icet_node_last_timing(icet_node* nd, float out_ms[3]) {
    if (!nd || !out_ms) return ICET_ERR_BAD_ARG;
    if (!nd->timing_valid) return ICET_ERR_BAD_ARG;
    float a = 0, b = 0, c = 0;
    NCHK(nd, hipEventElapsedTime(&a, nd->ev[0], nd->ev[1]));
    NCHK(nd, hipEventElapsedTime(&b, nd->pipelined ? nd->ev[5] : nd->ev[1], nd->ev[2]));
    if (nd->timed_map) { NCHK(nd, hipEventSynchronize(nd->ev[3])); NCHK(nd, hipEventElapsedTime(&c, nd->ev[4], nd->ev[3])); }
    out_ms[0] = a; out_ms[1] = b; out_ms[2] = c;
    return ICET_OK;
}

}  // extern "C"
