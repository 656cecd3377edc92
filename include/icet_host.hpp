// include/icet_host.hpp -- Eigen-free C++ host class over the C ABI (include/icet_hip.h).
//
// Mirrors the reference's `class ICET` (/root/reference/include/icet.h:36-116): the constructor IS the solve
// (src/icet.cpp:29-63) and callers read public members afterwards -- `X` and `pred_stds` in odometry_node /
// map_maker_node (src/odometry.cpp:76-79,126-131; src/simpleMapMaker.cpp:119-122), plus `points1`, `points2`,
// `clusterBounds` and the ellipsoid lists in the demos (src/icet_cpp_demo.cpp:47-57).  Same argument order,
// names, defaults and meaning; scans are passed as raw COLUMN-MAJOR N x 3 float buffers (what
// Eigen::MatrixXf::data() is), so this header needs no Eigen and is what tests/cpp exercises.
// include/icet.h is the thin Eigen adapter with the reference's exact constructor signature.
//
// Errors: the reference signals none (degenerate input yields zeros/NaN in X).  Here `status` holds the
// icet_status of the call and `error` its message; on failure X = X0 and pred_stds = 0, nothing throws.
// There is NO CPU fallback: without a usable MI355X the status is ICET_ERR_NO_DEVICE / ICET_ERR_HIP.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "icet_hip.h"

namespace icet_amd {

// One context (device + stream + workspace) per host thread and device, created on first use.
inline icet_ctx* thread_context(int device, icet_status* st_out = nullptr) {
    struct Holder {
        std::vector<std::pair<int, icet_ctx*>> ctxs;
        ~Holder() { for (auto& c : ctxs) icet_destroy(c.second); }
    };
    static thread_local Holder h;
    for (auto& c : h.ctxs) if (c.first == device) { if (st_out) *st_out = ICET_OK; return c.second; }
    icet_ctx* ctx = nullptr;
    icet_status st = icet_create(&ctx, device, nullptr);
    if (st_out) *st_out = st;
    if (st != ICET_OK) return nullptr;
    h.ctxs.emplace_back(device, ctx);
    return ctx;
}

class ICET {
public:
    struct Deferred {};      // tag: the constructor returns once the solve is enqueued (icet_solve_begin); finish() completes the object

    // scan1 / scan2: column-major N x 3 (x[0..n) | y[0..n) | z[0..n)), leading dimension ld >= n.
    // side_tables: 0 = X / pred_stds / cov only; 1 = every member a caller in the reference reads (clusterBounds, points2, the mu1 / sigma1 / U / L
    //   tables, testPoints, HTWH_i, ellipsoid1*); 2 = also the per-point members nobody in the reference reads (points1Spherical, pointIndices1,
    //   points2Spherical, pointIndices2: include/icet.h:79,82,95-96) -- several MB more per solve.
    // points2_out (optional): n2 x 3 column-major buffer of the caller that receives `points2` instead of the member vector.
    ICET(const float* scan1, int64_t n1, int64_t ld1, const float* scan2, int64_t n2, int64_t ld2, int runlen,
         const float X0[6], int num_bins_phi, int num_bins_theta, int n = 25, float thresh = 0.1f, float buff = 0.1f,
         int device = 0, int side_tables = 1, float* points2_out = nullptr)
        : ICET(Deferred{}, scan1, n1, ld1, scan2, n2, ld2, runlen, X0, num_bins_phi, num_bins_theta, n, thresh, buff, device, side_tables, points2_out) {
        finish();
    }

    // First half: both scans are read completely (the caller may reuse them), the registration runs on the device while the caller
    // does other host work -- include/icet.h makes the reference's member copies of the scans here -- until finish().
    ICET(Deferred, const float* scan1, int64_t n1, int64_t ld1, const float* scan2, int64_t n2, int64_t ld2, int runlen,
         const float X0[6], int num_bins_phi, int num_bins_theta, int n = 25, float thresh = 0.1f, float buff = 0.1f,
         int device = 0, int side_tables = 1, float* points2_out = nullptr)
        : rl(runlen), numBinsPhi(num_bins_phi), numBinsTheta(num_bins_theta), n(n), thresh(thresh), buff(buff), side_tables_(side_tables != 0) {
        for (int k = 0; k < 6; k++) { X[k] = X0 ? X0[k] : 0.f; x0_[k] = X[k]; pred_stds[k] = 0.f; dx[k] = 0.f; }
        cov.fill(0.f); HTWH_i.fill(0.f); HTWdz_i.fill(0.f);
        ctx_ = thread_context(device, &status);
        if (!ctx_) { error = "icet_create failed (no usable HIP device; this path has no CPU fallback)"; return; }
        icet_params p{runlen, num_bins_phi, num_bins_theta, n, thresh, buff, ICET_FLAG_NONE};
        const int64_t V = (int64_t)num_bins_phi * num_bins_theta;
        const int64_t RL = runlen > 0 ? runlen : 1;
        icet_aux aux{};
        if (side_tables && V > 0) {
            clusterBounds.assign((size_t)V * 6, 0.f); has_fit.assign(V, 0); mu1.assign((size_t)V * 3, 0.f); sigma1.assign((size_t)V * 9, 0.f);
            evecs1.assign((size_t)V * 9, 0.f); l_diag.assign((size_t)V * 3, 0.f); aux.evecs1 = evecs1.data(); aux.l_diag = l_diag.data();
            x_hist_.assign((size_t)RL * 6, 0.f); htwh_.assign((size_t)RL * 36, 0.f); htwdz_.assign((size_t)RL * 6, 0.f);
            testPoints.assign((size_t)V * 18, 0.f); aux.test_points = testPoints.data();
            aux.cluster_bounds = clusterBounds.data(); aux.has_fit = has_fit.data(); aux.mu1 = mu1.data(); aux.sigma1 = sigma1.data();
            aux.x_hist = x_hist_.data(); aux.htwh = htwh_.data(); aux.htwdz = htwdz_.data();
            // `points2` = scan 2 under the transform of the LAST iteration, i.e. X before the final update (src/icet.cpp:375-378 precede
            // :433): (p + t) * R(angles), column-major n2 x 3 -- computed on the device with the loop's own transform
            if (n2 > 0) {
                if (points2_out) aux.points2 = points2_out;
                else { points2.resize((size_t)n2 * 3); aux.points2 = points2.data(); }      // (value-initialised once; overwritten by the solve)
            }
            if (side_tables >= 2 && runlen > 0) {
                if (n1 > 0) { points1Spherical.resize((size_t)n1 * 3); pointIndex1.resize((size_t)n1); binStart1.assign((size_t)V + 1, 0);
                              aux.points1_spherical = points1Spherical.data(); aux.point_index1 = pointIndex1.data(); aux.bin_start1 = binStart1.data(); }
                if (n2 > 0) { points2Spherical.resize((size_t)n2 * 3); voxel2.resize((size_t)n2); aux.points2_spherical = points2Spherical.data(); aux.voxel2 = voxel2.data(); }
            }
        }
        status = icet_solve_begin(ctx_, &p, scan1, n1, ld1, scan2, n2, ld2, x0_, X.data(), pred_stds.data(), cov.data(), side_tables ? &aux : nullptr);
        if (status != ICET_OK) { error = icet_last_error(ctx_); for (int k = 0; k < 6; k++) { X[k] = x0_[k]; pred_stds[k] = 0.f; } return; }
        begun_ = true;
    }

    // Optional, between the Deferred constructor and finish(): returns when everything fitScan1 produces (clusterBounds, has_fit, mu1, sigma1,
    // evecs1, l_diag, testPoints, the ellipsoid1* lists) is in the members, while the Gauss-Newton loop still iterates on the device.
    void finish_keyframe() {
        if (!begun_ || kf_done_ || !side_tables_ || rl <= 0) return;
        if (icet_solve_keyframe_tables(ctx_) != ICET_OK) return;              // finish() reports the error
        kf_done_ = true;
        const int64_t V = (int64_t)numBinsPhi * numBinsTheta;
        // ellipsoid1* : one entry per fitted scan-1 voxel, phi-major order (src/icet.cpp:95-102,236-239); ellipsoid2* stay empty
        size_t fitted = 0;
        for (int64_t v = 0; v < V; v++) fitted += has_fit[v] ? 1 : 0;
        ellipsoid1Means.reserve(fitted); ellipsoid1Covariances.reserve(fitted); ellipsoid1Alphas.reserve(fitted);
        for (int64_t v = 0; v < V; v++) if (has_fit[v]) {
            ellipsoid1Means.push_back({mu1[3 * v], mu1[3 * v + 1], mu1[3 * v + 2]});
            std::array<float, 9> c; for (int k = 0; k < 9; k++) c[k] = sigma1[9 * v + k];
            ellipsoid1Covariances.push_back(c); ellipsoid1Alphas.push_back(0.3f);
        }
    }

    // Second half: waits for the device and fills the members.  Idempotent; called by the one-shot constructor.
    void finish() {
        if (!begun_) return;
        finish_keyframe();
        begun_ = false;
        status = icet_solve_end(ctx_);
        if (status != ICET_OK) { error = icet_last_error(ctx_); for (int k = 0; k < 6; k++) { X[k] = x0_[k]; pred_stds[k] = 0.f; } return; }
        const int runlen = rl;
        if (!side_tables_ || runlen <= 0) return;
        for (int k = 0; k < 36; k++) HTWH_i[k] = htwh_[(size_t)(runlen - 1) * 36 + k];
        for (int k = 0; k < 6; k++) HTWdz_i[k] = htwdz_[(size_t)(runlen - 1) * 6 + k];
        const float* xp = (runlen == 1) ? x0_ : &x_hist_[(size_t)(runlen - 2) * 6];
        for (int k = 0; k < 6; k++) dx[k] = x_hist_[(size_t)(runlen - 1) * 6 + k] - xp[k];
    }

    ~ICET() { if (begun_) (void)icet_solve_end(ctx_); }      // a Deferred object that was never finished: the context must not stay pending
    ICET(const ICET&) = delete;
    ICET& operator=(const ICET&) = delete;

    // utils::R (src/utils.cpp:144-152), row-major
    static void euler_R(float phi, float theta, float psi, float R[9]) {
        using std::cos; using std::sin;
        R[0] = cos(theta) * cos(psi); R[1] = sin(psi) * cos(phi) + sin(phi) * sin(theta) * cos(psi); R[2] = sin(phi) * sin(psi) - sin(theta) * cos(phi) * cos(psi);
        R[3] = -sin(psi) * cos(theta); R[4] = cos(phi) * cos(psi) - sin(phi) * sin(theta) * sin(psi); R[5] = sin(phi) * cos(psi) + sin(theta) * sin(psi) * cos(phi);
        R[6] = sin(theta); R[7] = -sin(phi) * cos(theta); R[8] = cos(phi) * cos(theta);
    }

    void step() { rl--; }     // the reference's step() is a stub that only decrements rl (src/icet.cpp:438-441)

    // algorithm params (include/icet.h:71-76)
    int rl, numBinsPhi, numBinsTheta, n; float thresh, buff;
    // results
    std::array<float, 6> X{}, pred_stds{}, dx{};
    std::array<float, 36> cov{};                 // 6x6 noise_mat, row-major (a local in the reference, src/icet.cpp:410-411)
    std::array<float, 36> HTWH_i{}; std::array<float, 6> HTWdz_i{};
    std::vector<float> clusterBounds;            // V x 6 row-major
    std::vector<float> testPoints;               // (V * 6) x 3 row-major: sigma points of the pruned axes (src/icet.cpp:213-231), zeros elsewhere
    std::vector<float> points2;                  // n2 x 3 column-major
    // side_tables == 2 only (include/icet.h:79,82,95-96 of the reference):
    std::vector<float> points1Spherical;         // n1 x 3 column-major (r | theta | phi): the scan-1 rows in the order the reference's sort + swap loop leaves them (src/icet.cpp:69-83)
    std::vector<int32_t> pointIndex1, binStart1; // pointIndices1[theta][phi] = pointIndex1[binStart1[v] .. binStart1[v + 1]), v = numBinsTheta * phi + theta: rows of points1Spherical, ascending
    std::vector<float> points2Spherical;         // n2 x 3 column-major: points2 in spherical coordinates (src/icet.cpp:387), caller's row order
    std::vector<int32_t> voxel2;                 // n2: the voxel of every row of points2 (src/icet.cpp:388): pointIndices2[theta][phi] = the rows i with voxel2[i] == v, ascending
    // the scan-1 voxel table behind the reference's std::map members mu1 / sigma1 / U / L (include/icet.h:89-94), dense over the V voxels
    // (row v = numBinsTheta * phi + theta, src/icet.cpp:149); only rows with has_fit[v] == 1 are map entries in the reference
    std::vector<int32_t> has_fit;                // V
    std::vector<float> mu1, sigma1;              // V x 3, V x 9 (row-major 3x3)
    std::vector<float> evecs1;                   // V x 9: eigenvectors of sigma1 as COLUMNS, ascending eigenvalues (the reference's U is the transpose, src/icet.cpp:184)
    std::vector<float> l_diag;                   // V x 3: diagonal of L (1 = axis kept, src/icet.cpp:205-232)
    std::vector<std::array<float, 3>> ellipsoid1Means, ellipsoid2Means;
    std::vector<std::array<float, 9>> ellipsoid1Covariances, ellipsoid2Covariances;
    std::vector<float> ellipsoid1Alphas, ellipsoid2Alphas;
    icet_status status = ICET_OK;
    std::string error;

private:
    icet_ctx* ctx_ = nullptr; bool begun_ = false, kf_done_ = false, side_tables_ = true;
    float x0_[6] = {0, 0, 0, 0, 0, 0};
    std::vector<float> x_hist_, htwh_, htwdz_;
};

}  // namespace icet_amd
