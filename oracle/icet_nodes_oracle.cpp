// oracle/icet_nodes_oracle.cpp -- TEST INFRASTRUCTURE.  CPU restatement of what the reference's two ROS nodes do
// around the ICET constructor.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use it; the product (libicet_hip.so) never does.
//
// Follows, line by line (ROS / PCL plumbing left out):
//   OdometryNode::pointCloudCallback   /root/reference/src/odometry.cpp:46-98,113-118
//   MapMakerNode::pointCloudCallback   /root/reference/src/simpleMapMaker.cpp:86-172
//   EigenQueue                         /root/reference/src/simpleMapMaker.cpp:18-59
//   ScanRegistrationNode::pointcloudCallback  /root/reference/src/scanMatcher.cpp:30-110
// PARITY UNPINNED against the real reference for the same reason as icet_oracle.cpp (Eigen, PCL and ROS are absent
// from this image and the reference has no tests or golden data for these nodes).  Under-specified points and the
// choice made here:
//   * `row(i).norm()`  = sqrt((x*x + y*y) + z*z) in float, no fused multiply-add;
//   * `rot_mat.inverse()` on a dynamic MatrixXf = Eigen's PartialPivLU route: LU with row pivoting, then L and U
//     triangular solves against the permuted identity (Eigen 3.3.7, LU/InverseImpl.h + PartialPivLU.h);
//   * `Matrix4f * Matrix4f` accumulates k = 0..3 in order;
//   * `Quaternionf(Matrix3f)` = Eigen/src/Geometry/Quaternion.h `quaternionbase_assign_impl<Other,3,3>`;
//   * `std::shuffle` / `std::mt19937` are the host C++ library's own (libstdc++ here, as in the reference's build);
//   * the down-sample copies min(downsampleSize, rows) rows (the reference reads past the index vector when the
//     scan is shorter than downsampleSize -- undefined behaviour, simpleMapMaker.cpp:155-157).
#include "icet_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

namespace {

struct Scan { std::vector<float> x, y, z; int64_t n() const { return (int64_t)x.size(); } };

struct Node {
    icet_oracle_node_params p;
    bool initialized = false;
    Scan prev;
    float X0[6] = {0, 0, 0, 0, 0, 0};
    float pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::mt19937 gen;                     // default seed, simpleMapMaker.cpp:258
    // EigenQueue (simpleMapMaker.cpp:18-59)
    std::vector<float> qx, qy, qz; int64_t pos = 0; bool filled = false;
    // scanMatcher.cpp: scan2_in_scan1_frame (:76) and snailTrail (:27-28, 79-84)
    Scan aligned;
    Scan snail;
};

// Eigen's PartialPivLU inverse of a 3 x 3 (row-major in, row-major out)
void inverse3_partial_piv_lu(const float* A, float* inv) {
    float lu[9]; std::memcpy(lu, A, sizeof(lu));
    int perm[3] = {0, 1, 2};
    for (int k = 0; k < 3; k++) {
        int piv = k; float best = std::fabs(lu[k * 3 + k]);
        for (int r = k + 1; r < 3; r++) { const float v = std::fabs(lu[r * 3 + k]); if (v > best) { best = v; piv = r; } }
        if (piv != k) { for (int c = 0; c < 3; c++) std::swap(lu[k * 3 + c], lu[piv * 3 + c]); std::swap(perm[k], perm[piv]); }
        if (lu[k * 3 + k] != 0.f)
            for (int r = k + 1; r < 3; r++) {
                lu[r * 3 + k] /= lu[k * 3 + k];
                for (int c = k + 1; c < 3; c++) lu[r * 3 + c] -= lu[r * 3 + k] * lu[k * 3 + c];
            }
    }
    for (int col = 0; col < 3; col++) {
        float b[3];
        for (int r = 0; r < 3; r++) b[r] = (perm[r] == col) ? 1.f : 0.f;      // P * e_col
        for (int r = 1; r < 3; r++) for (int c = 0; c < r; c++) b[r] -= lu[r * 3 + c] * b[c];         // unit-lower solve
        for (int r = 2; r >= 0; r--) { for (int c = r + 1; c < 3; c++) b[r] -= lu[r * 3 + c] * b[c]; b[r] /= lu[r * 3 + r]; }
        for (int r = 0; r < 3; r++) inv[r * 3 + col] = b[r];
    }
}

void quat_from_matrix(const float* m /* 3x3 row-major */, float q[4] /* x y z w */) {
    float t = m[0] + m[4] + m[8];
    if (t > 0.f) {
        t = std::sqrt(t + 1.0f);
        q[3] = 0.5f * t; t = 0.5f / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0f);
        q[i] = 0.5f * t; t = 0.5f / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}

}  // namespace

extern "C" {

void* icet_oracle_node_create(const icet_oracle_node_params* p) {
    if (!p) return nullptr;
    Node* nd = new Node(); nd->p = *p;
    if (p->map_capacity > 0) { nd->qx.assign(p->map_capacity, 0.f); nd->qy.assign(p->map_capacity, 0.f); nd->qz.assign(p->map_capacity, 0.f); }
    if (p->flags & 4) { nd->snail.x.assign(1, 0.f); nd->snail.y.assign(1, 0.f); nd->snail.z.assign(1, 0.f); }      // scanMatcher.cpp:27-28
    return nd;
}

void icet_oracle_node_destroy(void* h) { delete static_cast<Node*>(h); }

int icet_oracle_node_push(void* h, const float* scan, int64_t n, int64_t ld, icet_oracle_node_result* res) {
    Node* nd = static_cast<Node*>(h);
    if (!nd || !res || n < 0 || ld < n || (n > 0 && !scan)) return 1;
    std::memset(res, 0, sizeof(*res));
    const float *sx = scan, *sy = scan + ld, *sz = scan + 2 * ld;
    if (!nd->initialized) {                                      // odometry.cpp:46-52: stored unfiltered
        nd->prev.x.assign(sx, sx + n); nd->prev.y.assign(sy, sy + n); nd->prev.z.assign(sz, sz + n);
        nd->initialized = true;
        res->solved = 0; res->n_kept = n; std::memcpy(res->pose, nd->pose, sizeof(nd->pose));
        { const float* P = nd->pose; const float m[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]}; quat_from_matrix(m, res->quat); }
        res->map_rows = nd->filled ? (int64_t)nd->qx.size() : nd->pos;
        return 0;
    }
    // odometry.cpp:57-70 / simpleMapMaker.cpp:97-110
    Scan cur;
    for (int64_t i = 0; i < n; i++) {
        const float d = std::sqrt((sx[i] * sx[i] + sy[i] * sy[i]) + sz[i] * sz[i]);
        if ((nd->p.flags & 1) || d > nd->p.min_range) { cur.x.push_back(sx[i]); cur.y.push_back(sy[i]); cur.z.push_back(sz[i]); }
    }
    auto pack = [](const Scan& s) { std::vector<float> m; m.reserve(3 * s.x.size()); m.insert(m.end(), s.x.begin(), s.x.end()); m.insert(m.end(), s.y.begin(), s.y.end()); m.insert(m.end(), s.z.begin(), s.z.end()); return m; };
    const std::vector<float> m1 = pack(nd->prev), m2 = pack(cur);
    float X[6], stds[6], cov[36];
    static const float none[3] = {0.f, 0.f, 0.f};                // an empty scan (every row inside min_range, an empty cloud): the solve wants a pointer, reads nothing
    const int rc = icet_oracle_solve(&nd->p.solve, m1.empty() ? none : m1.data(), nd->prev.n(), nd->prev.n(), m2.empty() ? none : m2.data(), cur.n(), cur.n(), nd->X0, X, stds, cov, nullptr);
    if (rc != 0) return rc;
    // seed for the next frame: odometry.cpp:82 / simpleMapMaker.cpp:124
    for (int k = 0; k < 6; k++) nd->X0[k] = nd->p.seed_x0 ? X[k] : 0.f;
    // simpleMapMaker.cpp:129-137
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
    float R[9]; { const float a[3] = {X[3], X[4], X[5]}; icet_oracle_R(a, R); }
    // map queue: simpleMapMaker.cpp:147-158 then EigenQueue::add_new_scan :34-41
    if (nd->p.map_capacity > 0) {
        const int64_t rows = cur.n();
        std::vector<std::size_t> indices(rows);
        std::iota(indices.begin(), indices.end(), 0);
        std::shuffle(indices.begin(), indices.end(), nd->gen);
        const int64_t m = std::min<int64_t>(nd->p.map_downsample, rows);
        const int64_t cap = nd->p.map_capacity;
        for (int64_t i = 0; i < m; i++) {
            nd->qx[nd->pos] = cur.x[indices[i]]; nd->qy[nd->pos] = cur.y[indices[i]]; nd->qz[nd->pos] = cur.z[indices[i]];
            nd->pos = (nd->pos + 1) % cap;
            if (nd->pos == 0) nd->filled = true;
        }
        float Rinv[9]; inverse3_partial_piv_lu(R, Rinv);
        for (int64_t i = 0; i < cap; i++) {                       // matrix = (matrix.rowwise() - trans) * rot_mat.inverse()
            const float a = nd->qx[i] - X[0], b = nd->qy[i] - X[1], c = nd->qz[i] - X[2];
            nd->qx[i] = (a * Rinv[0] + b * Rinv[3]) + c * Rinv[6];
            nd->qy[i] = (a * Rinv[1] + b * Rinv[4]) + c * Rinv[7];
            nd->qz[i] = (a * Rinv[2] + b * Rinv[5]) + c * Rinv[8];
        }
    }
    if (nd->p.flags & 6) {
        // (M * rot_mat.inverse()).rowwise() - trans : rotate first, then subtract (scanMatcher.cpp:76, 80)
        float Rinv[9]; inverse3_partial_piv_lu(R, Rinv);
        auto xf = [&](Scan& sc) {
            for (int64_t i = 0; i < sc.n(); i++) {
                const float a = sc.x[i], b = sc.y[i], c = sc.z[i];
                sc.x[i] = ((a * Rinv[0] + b * Rinv[3]) + c * Rinv[6]) - X[0];
                sc.y[i] = ((a * Rinv[1] + b * Rinv[4]) + c * Rinv[7]) - X[1];
                sc.z[i] = ((a * Rinv[2] + b * Rinv[5]) + c * Rinv[8]) - X[2];
            }
        };
        if (nd->p.flags & 2) { nd->aligned = cur; xf(nd->aligned); }
        if (nd->p.flags & 4) { xf(nd->snail); nd->snail.x.push_back(0.f); nd->snail.y.push_back(0.f); nd->snail.z.push_back(0.f); }
    }
    nd->prev = cur;                                               // odometry.cpp:88
    // X_homo = X_homo * X_homo_i  (odometry.cpp:91-98)
    const float Hi[16] = {R[0], R[1], R[2], X[0], R[3], R[4], R[5], X[1], R[6], R[7], R[8], X[2], 0, 0, 0, 1};
    float P[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { float s = 0.f; for (int k = 0; k < 4; k++) s += nd->pose[r * 4 + k] * Hi[k * 4 + c]; P[r * 4 + c] = s; }
    std::memcpy(nd->pose, P, sizeof(P));
    res->solved = 1; res->n_kept = cur.n();
    std::memcpy(res->X, X, sizeof(X)); std::memcpy(res->pred_stds, stds, sizeof(stds)); std::memcpy(res->pose, P, sizeof(P));
    { const float m[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]}; quat_from_matrix(m, res->quat); }
    res->map_rows = nd->filled ? (int64_t)nd->qx.size() : nd->pos;
    return 0;
}

// EigenQueue::getQueue (simpleMapMaker.cpp:43-50) -> rows x 3 column-major, leading dimension ld
int64_t icet_oracle_node_map(void* h, float* out, int64_t ld) {
    Node* nd = static_cast<Node*>(h);
    if (!nd) return -1;
    const int64_t cap = (int64_t)nd->qx.size();
    const int64_t rows = nd->filled ? cap : nd->pos;
    if (!out) return rows;
    if (ld < rows) return -1;
    for (int64_t i = 0; i < rows; i++) {
        const int64_t s = nd->filled ? (nd->pos + i) % cap : i;    // bottomRows(maxSize - pos) first, then topRows(pos)
        out[i] = nd->qx[s]; out[ld + i] = nd->qy[s]; out[2 * ld + i] = nd->qz[s];
    }
    return rows;
}

static int64_t copy_scan(const Scan& sc, float* out, int64_t ld) {
    const int64_t rows = sc.n();
    if (!out) return rows;
    if (ld < rows) return -1;
    for (int64_t i = 0; i < rows; i++) { out[i] = sc.x[i]; out[ld + i] = sc.y[i]; out[2 * ld + i] = sc.z[i]; }
    return rows;
}
int64_t icet_oracle_node_aligned(void* h, float* out, int64_t ld) { Node* nd = static_cast<Node*>(h); return nd ? copy_scan(nd->aligned, out, ld) : -1; }
int64_t icet_oracle_node_snail_trail(void* h, float* out, int64_t ld) { Node* nd = static_cast<Node*>(h); return nd ? copy_scan(nd->snail, out, ld) : -1; }

}  // extern "C"
