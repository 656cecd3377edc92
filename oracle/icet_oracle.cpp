// oracle/icet_oracle.cpp -- TEST INFRASTRUCTURE: CPU restatement of the reference hot path.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  It is
// the parity checker (and the timed CPU "port" baseline); the product path (icet_amd/csrc) never
// calls into it.
//
// What it restates (all citations relative to /root/reference):
//   ICET::ICET                 src/icet.cpp:29-63      -> Solver::run
//   ICET::fitScan1             src/icet.cpp:68-107     -> Solver::fitScan1
//   ICET::fitCells1            src/icet.cpp:109-252    -> Solver::fitCells1
//   ICET::prepScan2            src/icet.cpp:254-277    -> Solver::prepScan2
//   ICET::fitCells2            src/icet.cpp:279-344    -> Solver::fitCells2
//   ICET::parallelFitCells2    src/icet.cpp:346-370    -> Solver::voxelLoopPool   (mode 1, 4 workers, src/icet.cpp:31)
//   ICET::fitScan2             src/icet.cpp:372-436    -> Solver::fitScan2
//   ICET::checkCondition       src/icet.cpp:443-492    -> Solver::checkCondition
//   ICET::get_H                src/icet.cpp:494-532    -> get_H
//   ICET::sortSphericalCoordinates src/icet.cpp:534-554 -> Solver::binPoints
//   ICET::findCluster          src/icet.cpp:557-607    -> findCluster
//   ICET::filterPointsInsideCluster src/icet.cpp:609-652 -> insideBounds (predicate form)
//   ICET::testSigmaPoints      src/icet.cpp:654-696    -> inline in fitCells1
//   utils::cartesianToSpherical src/utils.cpp:93-119   -> c2s
//   utils::sphericalToCartesian src/utils.cpp:121-142  -> s2c
//   utils::R                   src/utils.cpp:144-152   -> eulerR
//   ThreadPool                 include/ThreadPool.h, ThreadPool.tpp, src/ThreadPool.cpp -> Pool
//
// Quirks Q1-Q14 of SURVEY.md section 8(a) are reproduced, not fixed.  Choices where the reference is
// under-specified (documented deviations, cannot be checked against the real binary here):
//   * std::sort(std::execution::par) tie order (src/icet.cpp:75,266) -> stable by original index.
//   * Eigen is absent: decompositions restated in smalllinalg.h from Eigen 3.3.7's algorithms.
//   * fitCells2 reading sigma1/U/L of a voxel whose scan-1 fit never ran (std::map default-inserts
//     an uninitialised Matrix3f, src/icet.cpp:315-336) is undefined behaviour in the reference; here
//     such a voxel contributes nothing and is counted in Trace::n_ub_voxels.
//   * float expression contraction: built with -ffp-contract=off so results do not depend on the
//     host's FMA support (the reference's -O3 -march=native build is free to fuse).
//   * THE SHARED ARITHMETIC RULE (round 2).  Two things in the reference are not pinned by its source: which libm its
//     float atan2/acos/sin/cos come from (glibc's are accurate to an ulp, not correctly rounded, and change between
//     versions), and the order in which Eigen adds up the rows for mean / covariance (vectorised, unspecified).  Both
//     reach the per-voxel covariance in its last bits, and the reference's result depends on those bits through the
//     SIGNS of the scan-1 eigenvectors (Q8/Q9).  So that device and oracle can be compared WITHOUT aligning signs,
//     both follow one rule, stated mathematically rather than by shared code:
//       - every transcendental is the correctly rounded float of the exact value: evaluated in double, rounded once
//         (here glibc's double functions; on the device ocml's double atan2/acos and an algebraic double evaluation of
//         sin/cos -- independent implementations that agree unless a double result lies within ~1e-16 of a float
//         rounding boundary, probability ~1e-8 per call);
//       - the per-voxel sums (mean, centred products) are exact: accumulated in double from float addends, rounded to
//         float once, then divided in float -- the value every summation order approximates;
//       - hypot in the QR step is Eigen 3.3's own formula p * sqrt(1 + (q/p)^2) (numext::hypot), not libm's.
//       (round 5: the sines / cosines of the rotation matrix and of get_H's Jacobians follow the rule as well -- they were glibc's float functions in
//       every mode until then, the device used ocml's, and the two matrices differed in a last bit now and then.)
//     mode | ICET_ORACLE_LIBMF switches all three back to the literal expression types of the reference source (glibc float
//     functions, sequential float sums, std::hypot): tests use it to show that the choice moves X by less than the
//     algorithm's own 1-ulp sensitivity.
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
// (SURVEY.md section 4), and neither its C++ (needs Eigen) nor its Python (needs TensorFlow)
// can run in this image.
#include "icet_oracle.h"
#include "smalllinalg.h"

#include <vector>
#include <numeric>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <future>
#include <queue>
#include <functional>
#include <atomic>
#include <memory>
#include <chrono>

namespace ico {

// ---------------------------------------------------------------- utils.cpp:93-152
static inline float atan2_cr(float y, float x) { return (float)std::atan2((double)y, (double)x); }
static inline float acos_cr(float q) { return (float)std::acos((double)q); }
static inline float sin_cr(float a) { return (float)std::sin((double)a); }
static inline float cos_cr(float a) { return (float)std::cos((double)a); }

static inline void c2s_one(float x, float y, float z, float& r, float& th, float& ph, bool libmf = false) {
    r = std::sqrt(x * x + y * y + z * z);               // rowwise().norm()
    th = libmf ? std::atan2(y, x) : atan2_cr(y, x);     // float overload in the reference (src/utils.cpp:103)
    if (th < 0.0) th = (float)((double)th + 2.0 * M_PI);
    ph = libmf ? std::acos(z / r) : acos_cr(z / r);
    if (std::isnan(r)) r = 1000.0f;                     // (isNaN).select(1000.0, .)
    if (std::isnan(th)) th = 1000.0f;
    if (std::isnan(ph)) ph = 1000.0f;
}
static inline void s2c_one(float r, float th, float ph, float& x, float& y, float& z, bool libmf = false) {
    const float sp = libmf ? std::sin(ph) : sin_cr(ph), cp = libmf ? std::cos(ph) : cos_cr(ph);
    const float st = libmf ? std::sin(th) : sin_cr(th), ct = libmf ? std::cos(th) : cos_cr(th);
    x = r * sp * ct;                                    // src/utils.cpp:134-136, left to right in float
    y = r * sp * st;
    z = r * cp;
}
// mean and covariance of m Cartesian rows (src/icet.cpp:160-162, 304-306): unbiased two-pass form.  Shared rule: exact sums
// (double accumulators over float addends), rounded once; libmf: sequential float sums.
// ICET_ORACLE_PINV3_DOUBLE (attribution experiment, icet_oracle.h): Moore-Penrose pseudo-inverse of a symmetric 3 x 3 by cyclic Jacobi in double,
// eigenvalues <= 3 eps x the largest treated as rank deficiency -- the rule of the HIP path's pinv3_sym, written independently.
static inline Mat pinv3_double(const Mat& Rp) {
    double A[3][3], Vv[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] = 0.5 * ((double)Rp(i, j) + (double)Rp(j, i));
    for (int sweep = 0; sweep < 30; sweep++) {
        const double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]), dg = std::fabs(A[0][0]) + std::fabs(A[1][1]) + std::fabs(A[2][2]);
        if (off <= 1e-17 * dg) break;
        for (int p = 0; p < 3; p++) for (int q = p + 1; q < 3; q++) {
            if (A[p][q] == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0)), c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; k++) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
            for (int k = 0; k < 3; k++) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
            for (int k = 0; k < 3; k++) { const double vkp = Vv[k][p], vkq = Vv[k][q]; Vv[k][p] = c * vkp - s * vkq; Vv[k][q] = s * vkp + c * vkq; }
        }
    }
    const double lmax = std::max(std::max(std::fabs(A[0][0]), std::fabs(A[1][1])), std::fabs(A[2][2])), thr = (double)(3.0f * FLT_EPSILON) * lmax;
    Mat W(3, 3);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double t = 0.0;
        for (int k = 0; k < 3; k++) if (std::fabs(A[k][k]) > thr) t += Vv[i][k] * Vv[j][k] / A[k][k];
        W(i, j) = (float)t;
    }
    return W;
}

static inline void mean_cov(const std::vector<float>& cx, const std::vector<float>& cy, const std::vector<float>& cz, bool libmf, float mean[3], Mat& cov) {
    const long rows = (long)cx.size();
    cov = Mat(3, 3);
    if (libmf) {
        mean[0] = mean[1] = mean[2] = 0.f;
        for (long k = 0; k < rows; k++) { mean[0] += cx[k]; mean[1] += cy[k]; mean[2] += cz[k]; }
        for (int a = 0; a < 3; a++) mean[a] /= (float)rows;
        for (long k = 0; k < rows; k++) {
            float d[3] = {cx[k] - mean[0], cy[k] - mean[1], cz[k] - mean[2]};
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) cov(a, b) += d[a] * d[b];
        }
        for (int a = 0; a < 9; a++) cov.a[a] = cov.a[a] / static_cast<float>(rows - 1);
        return;
    }
    double s[3] = {0.0, 0.0, 0.0};
    for (long k = 0; k < rows; k++) { s[0] += (double)cx[k]; s[1] += (double)cy[k]; s[2] += (double)cz[k]; }
    for (int a = 0; a < 3; a++) mean[a] = (float)s[a] / (float)rows;
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (long k = 0; k < rows; k++) {
        const float d[3] = {cx[k] - mean[0], cy[k] - mean[1], cz[k] - mean[2]};
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) c[3 * a + b] += (double)(d[a] * d[b]);     // float product, exact sum
    }
    for (int a = 0; a < 9; a++) cov.a[a] = (float)c[a] / static_cast<float>(rows - 1);
}
// Shared rule: the six sines / cosines are the correctly rounded floats (sin_cr / cos_cr), like every other transcendental of the path -- the device's
// write_xf evaluates them the same way, so both sides transform scan 2 with the SAME matrix bits whenever they hold the same X; libmf: glibc's float functions.
struct Trig6 { float sph, cph, sth, cth, sps, cps; };
static inline Trig6 trig6(float phi, float theta, float psi, bool libmf) {
    Trig6 t;
    if (libmf) { t.sph = std::sin(phi); t.cph = std::cos(phi); t.sth = std::sin(theta); t.cth = std::cos(theta); t.sps = std::sin(psi); t.cps = std::cos(psi); }
    else { t.sph = sin_cr(phi); t.cph = cos_cr(phi); t.sth = sin_cr(theta); t.cth = cos_cr(theta); t.sps = sin_cr(psi); t.cps = cos_cr(psi); }
    return t;
}
static Mat eulerR(float phi_, float theta_, float psi_, bool libmf = false) {
    const Trig6 t = trig6(phi_, theta_, psi_, libmf);
    Mat m(3, 3);
    m(0,0) = t.cth*t.cps; m(0,1) = t.sps*t.cph+t.sph*t.sth*t.cps; m(0,2) = t.sph*t.sps-t.sth*t.cph*t.cps;
    m(1,0) = -t.sps*t.cth; m(1,1) = t.cph*t.cps-t.sph*t.sth*t.sps; m(1,2) = t.sph*t.cps+t.sth*t.sps*t.cph;
    m(2,0) = t.sth; m(2,1) = -t.sph*t.cth; m(2,2) = t.cph*t.cth;
    return m;
}

// ---------------------------------------------------------------- icet.cpp:494-532
static Mat get_H(const float mu[3], const float angs[3], bool libmf = false) {
    const Trig6 t = trig6(angs[0], angs[1], angs[2], libmf);
    const float sph = t.sph, cph = t.cph, sth = t.sth, cth = t.cth, sps = t.sps, cps = t.cps;
    Mat H(3, 6);
    H(0,0) = -1.f; H(1,1) = -1.f; H(2,2) = -1.f;
    float Jx[9] = {0.f, (-sps*sph + cph*sth*cps), (cph*sps + sth*sph*cps),
                   0.f, (-sph*cps - cph*sth*sps), (cph*cps - sth*sps*sph),
                   0.f, (-cph*cth), (-sph*cth)};
    float Jy[9] = {(-sth*cps), (cth*sph*cps), (-cth*cph*cps),
                   (sps*sth), (-cth*sph*sps), (cth*sps*cph),
                   (cth), (sph*sth), (-sth*cph)};
    float Jz[9] = {(-cth*sps), (cps*cph - sph*sth*sps), (cps*sph + sth*cph*sps),
                   (-cps*cth), (-sps*cph - sph*sth*cps), (-sph*sps + sth*cps*cph),
                   0.f, 0.f, 0.f};
    for (int i = 0; i < 3; i++) {
        H(i, 3) = Jx[3*i] * mu[0] + Jx[3*i+1] * mu[1] + Jx[3*i+2] * mu[2];
        H(i, 4) = Jy[3*i] * mu[0] + Jy[3*i+1] * mu[1] + Jy[3*i+2] * mu[2];
        H(i, 5) = Jz[3*i] * mu[0] + Jz[3*i+1] * mu[1] + Jz[3*i+2] * mu[2];
    }
    return H;
}

// ---------------------------------------------------------------- ThreadPool.{h,tpp,cpp}
class Pool {
public:
    explicit Pool(size_t n) : stop(false) { for (size_t i = 0; i < n; i++) workers.emplace_back(&Pool::work, this); }
    ~Pool() { stop.store(true); cv.notify_all(); for (auto& w : workers) w.join(); }
    template <class F> auto enqueue(F&& f) -> std::future<decltype(f())> {
        using R = decltype(f());
        auto task = std::make_shared<std::packaged_task<R()>>(std::forward<F>(f));
        std::future<R> res = task->get_future();
        { std::unique_lock<std::mutex> lk(m); tasks.emplace([task]() { (*task)(); }); }
        cv.notify_one();
        return res;
    }
private:
    void work() {
        while (!stop) {
            std::function<void()> t;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [this] { return stop.load() || !tasks.empty(); });
                if (stop.load() && tasks.empty()) return;
                t = std::move(tasks.front()); tasks.pop();
            }
            t();
        }
    }
    std::vector<std::thread> workers; std::queue<std::function<void()>> tasks;
    std::mutex m; std::condition_variable cv; std::atomic<bool> stop;
};

struct Sph { std::vector<float> r, th, ph; void resize(size_t n) { r.resize(n); th.resize(n); ph.resize(n); } };

struct VoxelFit {            // what the reference keeps in sigma1/mu1/U/L std::maps (icet.h:89-94)
    bool has_fit = false;
    float mu[3] = {0, 0, 0};
    Mat sigma{3, 3};
    Mat V{3, 3};             // eigenvectors as columns; the reference stores U = V^T (icet.cpp:184)
    float Ldiag[3] = {0, 0, 0};
};

struct Contribution { Mat HTWH{6, 6}; Mat HTWdz{6, 1}; int n_in = 0; float mu2[3] = {0, 0, 0}; Mat sigma2{3, 3}; bool used = false; };

// ---------------------------------------------------------------- icet.cpp:557-607
// half_gap (NON-PARITY EXTENSION, ICET_ORACLE_HALF_GAP; the rule of python/utils.py:92-119 as the paper states it): the bounds reach half
// way to the nearest point outside the cluster -- the point walked just before its first one, the point that ended it -- at most buff;
// buff where there is no such point.
static std::pair<float, float> findCluster(const Sph& s, const int* idx, int numPoints, int n, float thresh, float buff, bool half_gap = false) {
    float innerDistance = 0.0f, outerDistance = 0.0f;
    long count = 0; float front = 0.f, back = 0.f;      // localPoints.size(), .front()(0), .back()(0)
    bool has_before = false; float before = 0.f;         // the point walked just before `front`
    auto in_buff = [&]() { return (half_gap && has_before) ? std::min(buff, 0.5f * std::abs(front - before)) : buff; };
    for (int i = 0; i < numPoints; i++) {
        float pr = s.r[idx[i]];
        if (count != 0 && std::abs(back - pr) <= thresh) {
            back = pr; count++;
        } else {
            if (count >= n) {
                innerDistance = front - in_buff();
                outerDistance = back + (half_gap ? std::min(buff, 0.5f * std::abs(pr - back)) : buff);
                return {innerDistance, outerDistance};
            } else {
                has_before = i > 0; if (i > 0) before = s.r[idx[i - 1]];
                count = 1; front = pr; back = pr;
            }
        }
    }
    if (count >= n) {
        if (count != 0 && front != 0) {
            innerDistance = front - in_buff();
            outerDistance = back + buff;
            return {innerDistance, outerDistance};
        } else {
            return {0.0f, 0.0f};
        }
    }
    return {innerDistance, outerDistance};
}

// icet.cpp:632-634 / 675-677 (closed float comparisons)
static inline bool insideBounds(float r, float azim, float elev, const float* lims) {
    return azim >= lims[0] && azim <= lims[1] && elev >= lims[2] && elev <= lims[3] && r >= lims[4] && r <= lims[5];
}

// In-place "sort" by r with the reference's one-step swap loop (icet.cpp:72-83, 264-274; quirk Q3).
static void sortAndScramble(Sph& s, bool true_sort = false, std::vector<float>* cx = nullptr, std::vector<float>* cy = nullptr, std::vector<float>* cz = nullptr) {
    const int N = (int)s.r.size();
    std::vector<int> index(N);
    std::iota(index.begin(), index.end(), 0);
    std::stable_sort(index.begin(), index.end(), [&](int a, int b) { return s.r[a] < s.r[b]; });
    if (true_sort) {                      // NON-PARITY EXTENSION (ICET_ORACLE_TRUE_SORT): apply the permutation properly
        Sph t; t.resize(N);
        for (int i = 0; i < N; i++) { t.r[i] = s.r[index[i]]; t.th[i] = s.th[index[i]]; t.ph[i] = s.ph[index[i]]; }
        s = t;
        if (cx) { std::vector<float> a(N), b(N), c(N); for (int i = 0; i < N; i++) { a[i] = (*cx)[index[i]]; b[i] = (*cy)[index[i]]; c[i] = (*cz)[index[i]]; } *cx = a; *cy = b; *cz = c; }
        return;
    }
    for (int i = 0; i < N; i++) {
        if (index[i] != i) {
            int j = index[i];
            std::swap(s.r[i], s.r[j]); std::swap(s.th[i], s.th[j]); std::swap(s.ph[i], s.ph[j]);
            if (cx) { std::swap((*cx)[i], (*cx)[j]); std::swap((*cy)[i], (*cy)[j]); std::swap((*cz)[i], (*cz)[j]); }
            std::swap(index[i], index[j]);
        }
    }
}

// ---------------------------------------------------------------- icet.cpp:443-492
// checkCondition: eigen-decomposition of H^T W H (ascending), then while |lambda_5 / lambda_k| > 1e6: pred_stds += U2.col(k) (Q12),
// drop the top row of L2.  (The reference's loop has no upper bound on eyecount -- with six axes pruned it would read eigenvalues(6);
// bounded here.)  Returns the number of pruned axes.
static int check_condition(const Mat& HTWH, bool libmf, float pred_stds[6], Mat& L2, Mat& lam, Mat& U2, float ev[6]) {
    const float cutoff = 1e6f;
    selfadjoint_eigen(HTWH, /*fixed3=*/false, ev, U2, libmf);
    float condition = ev[5] / ev[0];
    int keep_from = 0;
    int eyecount = 1;
    while (std::abs(condition) > cutoff && eyecount < 6) {
        for (int k = 0; k < 6; k++) pred_stds[k] += U2(k, eyecount - 1);     // icet.cpp:479 (Q12)
        keep_from++;
        condition = ev[5] / ev[eyecount];
        eyecount++;
    }
    L2 = Mat(6 - keep_from, 6);
    for (int i = 0; i < 6 - keep_from; i++) L2(i, keep_from + i) = 1.f;
    lam = Mat(6, 6); for (int i = 0; i < 6; i++) lam(i, i) = ev[i];
    return keep_from;
}

// The 6x6 tail of fitScan2, icet.cpp:410-430: noise_mat = pinv(HTWH), pred_stds = sqrt|diag|, checkCondition, dx = pinv(L2 lam U2^T) L2 U2^T HTWdz.
// A function of (HTWH, HTWdz) alone, so that the device's restatement of it can be compared bit for bit on the same inputs
// (icet_oracle_gn_tail / icet_debug_gn_tail, tests/test_gpu_parity.py).
static void gn_tail(const Mat& HTWH, const Mat& HTWdz, bool libmf, Mat& noise_mat, float pred_stds[6], Mat& d, float ev[6], int* pruned, int* rank) {
    noise_mat = cod_pinv(HTWH, rank);                                         // icet.cpp:410-411
    for (int k = 0; k < 6; k++) pred_stds[k] = std::sqrt(std::abs(noise_mat(k, k)));
    Mat L2, lam, U2;
    const int k0 = check_condition(HTWH, libmf, pred_stds, L2, lam, U2, ev);
    if (pruned) *pruned = k0;
    Mat U2t = transpose(U2);
    Mat innards = matmul(matmul(L2, lam), U2t);                               // icet.cpp:427
    Mat inv = cod_pinv(innards);
    Mat lhs = matmul(matmul(inv, L2), U2t);
    d = matmul(lhs, HTWdz);                                                   // icet.cpp:430
}

struct Solver {
    icet_oracle_params prm;
    int T, P, V, n;
    std::vector<float> p1x, p1y, p1z;       // points1
    std::vector<float> ogx, ogy, ogz;       // points2_OG
    std::vector<float> p2x, p2y, p2z;       // points2
    Sph sph1, sph2;
    std::vector<int> bin1_start, bin1_idx;  // pointIndices1 as CSR over v = T*phi + theta
    std::vector<int> bin2_start, bin2_idx;  // pointIndices2
    std::vector<float> clusterBounds;       // V x 6
    std::vector<VoxelFit> fit;
    Mat HTWH_i{6, 6}, HTWdz_i{6, 1};
    float X[6], dx[6], pred_stds[6];
    Mat noise_mat{6, 6};
    icet_oracle_trace* tr = nullptr;
    int iter_no = 0;
    std::unique_ptr<Pool> pool;

    // icet.cpp:534-554.  Bins in double from float angles; lists hold ascending point index.
    void binPoints(const Sph& s, std::vector<int>& start, std::vector<int>& idx) {
        const int N = (int)s.r.size();
        std::vector<int> b(N);
        start.assign(V + 1, 0);
        for (int i = 0; i < N; ++i) {
            float theta = s.th[i], phi = s.ph[i];
            int binTheta = static_cast<int>((theta / (2 * M_PI)) * T) % T;
            int binPhi = static_cast<int>((phi / M_PI) * P) % P;
            b[i] = T * binPhi + binTheta;
            start[b[i] + 1]++;
        }
        for (int v = 0; v < V; v++) start[v + 1] += start[v];
        idx.resize(N);
        std::vector<int> cur(start.begin(), start.end() - 1);
        for (int i = 0; i < N; i++) idx[cur[b[i]]++] = i;
    }

    bool libmf = false;                // ICET_ORACLE_LIBMF: float libm + sequential float sums instead of the shared rule
    bool skip_rt2 = false;             // ICET_ORACLE_SKIP_RT2: scan 2 is never round-tripped through spherical coordinates (the device's deviation)
    bool rt2_thin_only = false;        // ICET_ORACLE_RT2_THIN: ... except for the points of voxels whose scan-1 Gaussian is thin (lambda_min < rt_tau)
    bool dev_arith = false;            // ICET_ORACLE_DEVICE_ARITH: the HIP path's documented deviations (icet_oracle.h)
    float rt_tau = 1e-5f;
    std::vector<float> lam_min1;       // smallest eigenvalue of sigma1 per voxel
    std::vector<float> rawx, rawy, rawz;   // scan 2 as given (before prepScan2's round trip), in scrambled order
    std::vector<float> rtx, rty, rtz;      // its transform by the current X (skip_rt2 modes)
    Sph sphr;                              // and the spherical coordinates of that
    const float* sign_ref = nullptr;   // optional V x 9 eigenvectors (columns, row-major) to align signs with; see fitCells1
    int n_sign_flips = 0;

    void boundsRow(int theta, int phi, float inner, float outer) {
        float azimMin_i = (static_cast<float>(theta) / T) * (2 * M_PI);
        float azimMax_i = (static_cast<float>(theta + 1) / T) * (2 * M_PI);
        float elevMin_i = (static_cast<float>(phi) / P) * (M_PI);
        float elevMax_i = (static_cast<float>(phi + 1) / P) * (M_PI);
        float* row = &clusterBounds[6 * (T * phi + theta)];
        row[0] = azimMin_i; row[1] = azimMax_i; row[2] = elevMin_i; row[3] = elevMax_i; row[4] = inner; row[5] = outer;
    }

    // icet.cpp:109-252
    void fitCells1(const int* indices, int cnt, int theta, int phi) {
        const int v = T * phi + theta;
        if ((size_t)cnt >= (size_t)n) {
            auto cd = findCluster(sph1, indices, cnt, n, prm.thresh, prm.buff, (prm.mode & ICET_ORACLE_HALF_GAP) != 0);
            float innerDistance = cd.first, outerDistance = cd.second;
            boundsRow(theta, phi, innerDistance, outerDistance);
            const float* lims = &clusterBounds[6 * v];
            std::vector<float> cx, cy, cz;
            for (int k = 0; k < cnt; k++) {
                int i = indices[k];
                if (insideBounds(sph1.r[i], sph1.th[i], sph1.ph[i], lims)) {
                    float x, y, z; s2c_one(sph1.r[i], sph1.th[i], sph1.ph[i], x, y, z, libmf);
                    cx.push_back(x); cy.push_back(y); cz.push_back(z);
                }
            }
            long rows = (long)cx.size();
            if (outerDistance > 0.1 && rows * 3 >= n) {
                VoxelFit& f = fit[v];
                float mean[3]; Mat cov;
                mean_cov(cx, cy, cz, libmf, mean, cov);
                f.has_fit = true; f.sigma = cov; for (int a = 0; a < 3; a++) f.mu[a] = mean[a];
                float ev[3]; Mat evec(3, 3);
                selfadjoint_eigen(cov, /*fixed3=*/true, ev, evec, libmf);
                // Eigenvector signs are implementation-defined, and the reference's result DEPENDS on them: through the
                // rows-of-V sigma points (Q9) and through L*U^T with U = V^T, which applies V and is therefore not
                // invariant under column sign flips (Q8).  On near-degenerate voxels the QR iteration's signs flip with
                // the last bits of the covariance, so two correct builds of the reference disagree there.  A test may pass
                // the eigenvectors another implementation obtained: columns are then flipped to agree with them, which
                // separates "different signs" from "different algebra".  NULL (the default) keeps the natural signs.
                if (sign_ref) {
                    const float* ref = sign_ref + (size_t)v * 9;
                    for (int k = 0; k < 3; k++) {
                        const float dot = evec(0, k) * ref[k] + evec(1, k) * ref[3 + k] + evec(2, k) * ref[6 + k];
                        if (dot < 0.f) { for (int r = 0; r < 3; r++) evec(r, k) = -evec(r, k); n_sign_flips++; }
                    }
                }
                f.V = evec; lam_min1[v] = ev[0];
                // axislen = 2*sqrt(lambda);  rotated = diag(axislen) * U^T = diag(axislen) * V  (rows!)  icet.cpp:187-193
                float sp[6][3];
                for (int k = 0; k < 3; k++) {
                    float al = 2.0f * std::sqrt(ev[k]);
                    for (int c = 0; c < 3; c++) {
                        float rot = al * evec(k, c);
                        sp[2 * k][c] = f.mu[c] + rot;
                        sp[2 * k + 1][c] = f.mu[c] - rot;
                    }
                }
                bool inside[6] = {false, false, false, false, false, false};
                for (int j = 0; j < 6; j++) {       // icet.cpp:669-686 incl. the early break
                    float r, az, el; c2s_one(sp[j][0], sp[j][1], sp[j][2], r, az, el, libmf);
                    if (insideBounds(r, az, el, lims)) inside[j] = true;
                    if (r > lims[5]) break;
                }
                for (int k = 0; k < 3; k++) f.Ldiag[k] = (inside[2 * k] || inside[2 * k + 1]) ? 1.f : 0.f;
                if (tr) for (int j = 0; j < 6; j++) for (int c = 0; c < 3; c++) tr->sigma_points[(size_t)(6 * v + j) * 3 + c] = sp[j][c];
            }
        } else {
            boundsRow(theta, phi, 0.f, 0.f);
        }
    }

    // icet.cpp:68-107
    void fitScan1() {
        const int N = (int)p1x.size();
        sph1.resize(N);
        for (int i = 0; i < N; i++) c2s_one(p1x[i], p1y[i], p1z[i], sph1.r[i], sph1.th[i], sph1.ph[i], libmf);
        sortAndScramble(sph1, (prm.mode & (ICET_ORACLE_TRUE_SORT | ICET_ORACLE_HALF_GAP)) != 0);
        binPoints(sph1, bin1_start, bin1_idx);
        for (int phi = 0; phi < P; phi++)
            for (int theta = 0; theta < T; theta++) {
                int v = T * phi + theta;
                fitCells1(&bin1_idx[bin1_start[v]], bin1_start[v + 1] - bin1_start[v], theta, phi);
            }
    }

    // icet.cpp:254-277
    void prepScan2() {
        const int N = (int)p2x.size();
        sph2.resize(N);
        for (int i = 0; i < N; i++) c2s_one(p2x[i], p2y[i], p2z[i], sph2.r[i], sph2.th[i], sph2.ph[i], libmf);
        if (skip_rt2) { rawx = p2x; rawy = p2y; rawz = p2z; }          // the rows as given, carried through the same scramble
        sortAndScramble(sph2, (prm.mode & (ICET_ORACLE_TRUE_SORT | ICET_ORACLE_HALF_GAP)) != 0, skip_rt2 ? &rawx : nullptr, &rawy, &rawz);
        ogx.resize(N); ogy.resize(N); ogz.resize(N);
        for (int i = 0; i < N; i++) s2c_one(sph2.r[i], sph2.th[i], sph2.ph[i], ogx[i], ogy[i], ogz[i], libmf);
    }

    // icet.cpp:279-344
    Contribution fitCells2(int theta, int phi) const {
        Contribution out;
        const int v = T * phi + theta;
        const size_t n1 = bin1_start[v + 1] - bin1_start[v];
        const size_t n2 = bin2_start[v + 1] - bin2_start[v];
        const float* lims = &clusterBounds[6 * v];
        if ((n2 > (size_t)n) && (n1 > (size_t)n) && (lims[5] > 1)) {
            const int* idx2 = &bin2_idx[bin2_start[v]];
            std::vector<float> cx, cy, cz;
            // experimental modes (skip_rt2): membership comes from the transform of the RAW rows; the Gaussian is fitted to those
            // Cartesian rows directly -- what the device does -- unless the voxel is thin and rt2_thin_only asks for the reference's
            // round-tripped rows there
            const bool raw_here = skip_rt2 && !(rt2_thin_only && fit[v].has_fit && lam_min1[v] < rt_tau);
            for (size_t k = 0; k < n2; k++) {
                int i = idx2[k];
                if (raw_here) {
                    if (insideBounds(sphr.r[i], sphr.th[i], sphr.ph[i], lims)) { cx.push_back(rtx[i]); cy.push_back(rty[i]); cz.push_back(rtz[i]); }
                } else if (insideBounds(sph2.r[i], sph2.th[i], sph2.ph[i], lims)) {
                    float x, y, z; s2c_one(sph2.r[i], sph2.th[i], sph2.ph[i], x, y, z, libmf);
                    cx.push_back(x); cy.push_back(y); cz.push_back(z);
                }
            }
            long rows = (long)cx.size();
            out.n_in = (int)rows;
            if (rows > n) {                                  // filteredPoints2.size()/3 > n, icet.cpp:302
                const VoxelFit& f = fit[v];
                if (!f.has_fit) { out.used = false; out.n_in = -(int)rows - 1; return out; }   // reference UB, see header
                float mean[3]; Mat cov;
                float db[3] = {0.f, 0.f, 0.f};
                if (dev_arith) {
                    // icet_solve_body.h: sums of d = q - mu1 (float) and of d d^T, exact; mean - mu1, mean and covariance from them in double, rounded once
                    double sd[3] = {0, 0, 0}, sdd[6] = {0, 0, 0, 0, 0, 0};
                    for (long k = 0; k < rows; k++) {
                        const float d[3] = {cx[k] - f.mu[0], cy[k] - f.mu[1], cz[k] - f.mu[2]};
                        for (int a = 0; a < 3; a++) sd[a] += (double)d[a];
                        sdd[0] += (double)d[0] * d[0]; sdd[1] += (double)d[0] * d[1]; sdd[2] += (double)d[0] * d[2];
                        sdd[3] += (double)d[1] * d[1]; sdd[4] += (double)d[1] * d[2]; sdd[5] += (double)d[2] * d[2];
                    }
                    const double fm = (double)rows, dbD[3] = {sd[0] / fm, sd[1] / fm, sd[2] / fm}, den = 1.0 / (double)(rows - 1);
                    for (int a = 0; a < 3; a++) { db[a] = (float)dbD[a]; mean[a] = (float)((double)f.mu[a] + dbD[a]); }
                    cov = Mat(3, 3);
                    const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
                    for (int q = 0; q < 6; q++) { const float c = (float)((sdd[q] - fm * dbD[ia[q]] * dbD[ib[q]]) * den); cov(ia[q], ib[q]) = c; cov(ib[q], ia[q]) = c; }
                } else
                mean_cov(cx, cy, cz, libmf, mean, cov);
                // R_noise = sigma1/(n1-1) + cov/(n2-1)            icet.cpp:315 (raw bin counts, Q10)
                Mat Rn(3, 3);
                const float d1 = (float)(n1 - 1), d2 = (float)(n2 - 1);
                for (int a = 0; a < 9; a++) Rn.a[a] = f.sigma.a[a] / d1 + cov.a[a] / d2;
                // L * U^T = L * V  (Q8)
                Mat Lm(3, 3); for (int k = 0; k < 3; k++) Lm(k, k) = f.Ldiag[k];
                Mat LUt = matmul(Lm, f.V);
                Mat Rp = matmul(matmul(matmul(LUt, Rn), transpose(f.V)), transpose(Lm));   // icet.cpp:317
                Mat W = (prm.mode & ICET_ORACLE_PINV3_DOUBLE) ? pinv3_double(Rp) : cod_pinv(Rp);   // icet.cpp:320-321
                float angs[3] = {X[3], X[4], X[5]};
                Mat H_j = get_H(mean, angs, libmf);
                Mat H_z = matmul(LUt, H_j);                                                  // icet.cpp:329
                Mat HzT = transpose(H_z);
                Mat HzTW = matmul(HzT, W);
                out.HTWH = matmul(HzTW, H_z);                                                // icet.cpp:332
                Mat mu1m(3, 1), mu2m(3, 1);
                for (int a = 0; a < 3; a++) { mu1m(a, 0) = f.mu[a]; mu2m(a, 0) = mean[a]; }
                Mat z1 = matmul(LUt, mu1m), z2 = matmul(LUt, mu2m);
                Mat dz(3, 1); for (int a = 0; a < 3; a++) dz(a, 0) = z2(a, 0) - z1(a, 0);
                if (dev_arith) { Mat dbm(3, 1); for (int a = 0; a < 3; a++) dbm(a, 0) = db[a]; dz = matmul(LUt, dbm); }
                if ((prm.mode & ICET_ORACLE_REJECT_MOVING) && iter_no >= 4 &&
                    (std::fabs(dz(0, 0)) > 0.3f || std::fabs(dz(1, 0)) > 0.3f || std::fabs(dz(2, 0)) > 0.3f)) { out.HTWH = Mat(6, 6); return out; }   // extension: a moving object
                out.HTWdz = matmul(HzTW, dz);                                                // icet.cpp:338
                out.used = true; out.sigma2 = cov; for (int a = 0; a < 3; a++) out.mu2[a] = mean[a];
            }
        }
        return out;
    }

    void record(int v, const Contribution& c) {
        if (!tr || iter_no >= tr->max_iters) return;
        size_t base = (size_t)iter_no * V + v;
        tr->n2_raw[base] = bin2_start[v + 1] - bin2_start[v];
        tr->n2_in[base] = c.n_in;
        tr->used[base] = c.used ? 1 : 0;
        for (int a = 0; a < 3; a++) tr->mu2[base * 3 + a] = c.mu2[a];
        for (int a = 0; a < 9; a++) tr->sigma2[base * 9 + a] = c.sigma2.a[a];
        if (c.n_in < 0) tr->n_ub_voxels++;
    }

    // icet.cpp:346-370 (the ThreadPool variant whose call is commented out at icet.cpp:407)
    void voxelLoopPool() {
        std::vector<std::future<Contribution>> futures;
        futures.reserve(V);
        for (int phi = 0; phi < P; phi++)
            for (int theta = 0; theta < T; theta++)
                futures.push_back(pool->enqueue([this, theta, phi]() { return fitCells2(theta, phi); }));
        int v = 0;
        for (auto& fu : futures) {
            Contribution c = fu.get();
            for (int a = 0; a < 36; a++) HTWH_i.a[a] += c.HTWH.a[a];
            for (int a = 0; a < 6; a++) HTWdz_i.a[a] += c.HTWdz.a[a];
            record(v++, c);
        }
    }

    // icet.cpp:443-492: now the free function ico::check_condition (below the class it is used through gn_tail)
    // icet.cpp:372-436
    void fitScan2() {
        const int N = (int)ogx.size();
        Mat rot = eulerR(X[3], X[4], X[5], libmf);
        p2x.resize(N); p2y.resize(N); p2z.resize(N);
        for (int i = 0; i < N; i++) {
            float a = ogx[i] + X[0], b = ogy[i] + X[1], c = ogz[i] + X[2];        // rowwise() + trans
            p2x[i] = a * rot(0, 0) + b * rot(1, 0) + c * rot(2, 0);              // points2 * rot_mat
            p2y[i] = a * rot(0, 1) + b * rot(1, 1) + c * rot(2, 1);
            p2z[i] = a * rot(0, 2) + b * rot(1, 2) + c * rot(2, 2);
        }
        HTWH_i = Mat(6, 6); HTWdz_i = Mat(6, 1);
        sph2.resize(N);
        for (int i = 0; i < N; i++) c2s_one(p2x[i], p2y[i], p2z[i], sph2.r[i], sph2.th[i], sph2.ph[i], libmf);
        if (skip_rt2) {
            rtx.resize(N); rty.resize(N); rtz.resize(N); sphr.resize(N);
            for (int i = 0; i < N; i++) {
                float a = rawx[i] + X[0], b = rawy[i] + X[1], c = rawz[i] + X[2];
                if (dev_arith) {
                    rtx[i] = std::fmaf(c, rot(2, 0), std::fmaf(b, rot(1, 0), a * rot(0, 0)));
                    rty[i] = std::fmaf(c, rot(2, 1), std::fmaf(b, rot(1, 1), a * rot(0, 1)));
                    rtz[i] = std::fmaf(c, rot(2, 2), std::fmaf(b, rot(1, 2), a * rot(0, 2)));
                } else {
                rtx[i] = a * rot(0, 0) + b * rot(1, 0) + c * rot(2, 0);
                rty[i] = a * rot(0, 1) + b * rot(1, 1) + c * rot(2, 1);
                rtz[i] = a * rot(0, 2) + b * rot(1, 2) + c * rot(2, 2);
                }
                c2s_one(rtx[i], rty[i], rtz[i], sphr.r[i], sphr.th[i], sphr.ph[i], libmf);
            }
            binPoints(sphr, bin2_start, bin2_idx);
        } else
        binPoints(sph2, bin2_start, bin2_idx);
        if ((prm.mode & ICET_ORACLE_POOL4)) {
            voxelLoopPool();
        } else {
            for (int phi = 0; phi < P; phi++)
                for (int theta = 0; theta < T; theta++) {
                    Contribution c = fitCells2(theta, phi);
                    for (int a = 0; a < 36; a++) HTWH_i.a[a] += c.HTWH.a[a];
                    for (int a = 0; a < 6; a++) HTWdz_i.a[a] += c.HTWdz.a[a];
                    record(T * phi + theta, c);
                }
        }
        Mat d; float ev[6]; int pruned = 0;
        gn_tail(HTWH_i, HTWdz_i, libmf, noise_mat, pred_stds, d, ev, &pruned, nullptr);       // icet.cpp:410-430
        if (tr && iter_no < tr->max_iters) { for (int i = 0; i < 6; i++) tr->eigvals[iter_no * 6 + i] = ev[i]; tr->pruned[iter_no] = pruned; }
        if (tr && iter_no < tr->max_iters) {
            for (int a = 0; a < 36; a++) tr->HTWH[iter_no * 36 + a] = HTWH_i.a[a];
            for (int a = 0; a < 6; a++) { tr->HTWdz[iter_no * 6 + a] = HTWdz_i.a[a]; tr->dx[iter_no * 6 + a] = d(a, 0); }
        }
        for (int k = 0; k < 6; k++) { dx[k] = d(k, 0); X[k] += dx[k]; }
        if (tr && iter_no < tr->max_iters) for (int a = 0; a < 6; a++) tr->X[iter_no * 6 + a] = X[a];
        iter_no++;
    }

    // icet.cpp:29-63
    int run(const float* s1, int64_t n1, int64_t ld1, const float* s2, int64_t n2, int64_t ld2, const float* x0) {
        T = prm.bins_theta; P = prm.bins_phi; V = T * P; n = prm.n;
        libmf = (prm.mode & ICET_ORACLE_LIBMF) != 0;
        dev_arith = (prm.mode & ICET_ORACLE_DEVICE_ARITH) != 0;
        skip_rt2 = (prm.mode & (ICET_ORACLE_SKIP_RT2 | ICET_ORACLE_RT2_THIN | ICET_ORACLE_DEVICE_ARITH)) != 0; rt2_thin_only = (prm.mode & ICET_ORACLE_RT2_THIN) != 0;
        lam_min1.assign((size_t)prm.bins_theta * prm.bins_phi, 0.f);
        p1x.assign(s1, s1 + n1); p1y.assign(s1 + ld1, s1 + ld1 + n1); p1z.assign(s1 + 2 * ld1, s1 + 2 * ld1 + n1);
        p2x.assign(s2, s2 + n2); p2y.assign(s2 + ld2, s2 + ld2 + n2); p2z.assign(s2 + 2 * ld2, s2 + 2 * ld2 + n2);
        for (int k = 0; k < 6; k++) { X[k] = x0[k]; pred_stds[k] = 0.f; dx[k] = 0.f; }
        clusterBounds.assign((size_t)V * 6, 0.f);
        fit.assign(V, VoxelFit());
        if ((prm.mode & ICET_ORACLE_POOL4)) pool.reset(new Pool(4));               // icet.cpp:31
        fitScan1();
        prepScan2();
        for (int iter = 0; iter < prm.runlen; iter++) fitScan2();
        if (tr) {
            for (int v = 0; v < V; v++) {
                tr->n1_raw[v] = bin1_start[v + 1] - bin1_start[v];
                tr->has_fit[v] = fit[v].has_fit ? 1 : 0;
                for (int a = 0; a < 6; a++) tr->bounds[v * 6 + a] = clusterBounds[v * 6 + a];
                for (int a = 0; a < 3; a++) { tr->mu1[v * 3 + a] = fit[v].mu[a]; tr->Ldiag[v * 3 + a] = fit[v].Ldiag[a]; }
                for (int a = 0; a < 9; a++) { tr->sigma1[v * 9 + a] = fit[v].sigma.a[a]; tr->evecs1[v * 9 + a] = fit[v].V.a[a]; }
            }
        }
        return 0;
    }
};

}  // namespace ico

extern "C" {

int icet_oracle_solve(const icet_oracle_params* p, const float* scan1, int64_t n1, int64_t ld1,
                      const float* scan2, int64_t n2, int64_t ld2, const float x0[6],
                      float x_out[6], float pred_stds_out[6], float cov_out[36], icet_oracle_trace* trace) {
    if (!p || !scan1 || !scan2 || !x0 || n1 < 0 || n2 < 0 || ld1 < n1 || ld2 < n2 || p->bins_phi <= 0 || p->bins_theta <= 0 || p->runlen < 0) return 1;
    ico::Solver s; s.prm = *p; s.tr = trace;
    if (trace) trace->n_ub_voxels = 0;
    s.run(scan1, n1, ld1, scan2, n2, ld2, x0);
    for (int k = 0; k < 6; k++) { x_out[k] = s.X[k]; pred_stds_out[k] = s.pred_stds[k]; }
    if (cov_out) for (int a = 0; a < 36; a++) cov_out[a] = s.noise_mat.a[a];
    return 0;
}

// Same solve with the eigenvector signs of every fitted voxel aligned to `evecs_ref` (V x 9, columns = eigenvectors, row-major;
// all-zero rows are ignored).  Returns the number of columns that had to be flipped in *n_flips.  TESTS ONLY.
int icet_oracle_solve_signed(const icet_oracle_params* p, const float* scan1, int64_t n1, int64_t ld1,
                             const float* scan2, int64_t n2, int64_t ld2, const float x0[6], const float* evecs_ref,
                             float x_out[6], float pred_stds_out[6], float cov_out[36], icet_oracle_trace* trace, int32_t* n_flips) {
    if (!p || !scan1 || !scan2 || !x0 || n1 < 0 || n2 < 0 || ld1 < n1 || ld2 < n2 || p->bins_phi <= 0 || p->bins_theta <= 0 || p->runlen < 0) return 1;
    ico::Solver s; s.prm = *p; s.tr = trace; s.sign_ref = evecs_ref;
    if (trace) trace->n_ub_voxels = 0;
    s.run(scan1, n1, ld1, scan2, n2, ld2, x0);
    for (int k = 0; k < 6; k++) { x_out[k] = s.X[k]; pred_stds_out[k] = s.pred_stds[k]; }
    if (cov_out) for (int a = 0; a < 36; a++) cov_out[a] = s.noise_mat.a[a];
    if (n_flips) *n_flips = s.n_sign_flips;
    return 0;
}

// One pair per host thread (the batched CPU baseline of BASELINE.md section 3).
int icet_oracle_solve_batch(const icet_oracle_params* p, int n_pairs, const float* const* scan1, const int64_t* n1,
                            const float* const* scan2, const int64_t* n2, const float* x0 /* n_pairs x 6 */,
                            float* x_out, float* pred_stds_out, float* cov_out, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    std::atomic<int> next(0);
    std::atomic<int> err(0);
    auto worker = [&]() {
        for (;;) {
            int k = next.fetch_add(1);
            if (k >= n_pairs) return;
            int rc = icet_oracle_solve(p, scan1[k], n1[k], n1[k], scan2[k], n2[k], n2[k], x0 + 6 * k,
                                       x_out + 6 * k, pred_stds_out + 6 * k, cov_out ? cov_out + 36 * k : nullptr, nullptr);
            if (rc) err.store(rc);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(worker);
    for (auto& t : th) t.join();
    return err.load();
}

// Timed loop for bench.py's cpu_baseline leg: solves the same pair `reps` times, returns mean seconds.
double icet_oracle_time_pair(const icet_oracle_params* p, const float* scan1, int64_t n1, const float* scan2, int64_t n2,
                             const float x0[6], int reps, float x_out[6]) {
    float ps[6], cov[36];
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) icet_oracle_solve(p, scan1, n1, n1, scan2, n2, n2, x0, x_out, ps, cov, nullptr);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count() / (reps > 0 ? reps : 1);
}

// ---- small exported helpers so tests can pin the restated decompositions against numpy -------------
void icet_oracle_eig_sym(const float* A, int n, int fixed3, float* evals, float* evecs) {
    ico::Mat M(n, n); for (int i = 0; i < n * n; i++) M.a[i] = A[i];
    ico::Mat Q; ico::selfadjoint_eigen(M, fixed3 != 0, evals, Q);
    for (int i = 0; i < n * n; i++) evecs[i] = Q.a[i];
}
int icet_oracle_pinv(const float* A, int rows, int cols, float* out) {
    ico::Mat M(rows, cols); for (int i = 0; i < rows * cols; i++) M.a[i] = A[i];
    int rank = 0; ico::Mat Pm = ico::cod_pinv(M, &rank);
    for (int i = 0; i < rows * cols; i++) out[i] = Pm.a[i];
    return rank;
}
void icet_oracle_c2s(const float* xyz, int64_t n, int64_t ld, float* out /* n x 3 col-major r|th|ph */) {
    for (int64_t i = 0; i < n; i++) ico::c2s_one(xyz[i], xyz[ld + i], xyz[2 * ld + i], out[i], out[n + i], out[2 * n + i]);
}
// The scramble on its own: given r[n], returns src[v] = original row that ends at position v.
void icet_oracle_scramble(const float* r, int64_t n, int32_t* src) {
    std::vector<int> index(n);
    std::iota(index.begin(), index.end(), 0);
    std::stable_sort(index.begin(), index.end(), [&](int a, int b) { return r[a] < r[b]; });
    for (int64_t i = 0; i < n; i++) src[i] = (int32_t)i;
    for (int64_t i = 0; i < n; i++) {
        if (index[i] != i) { int j = index[i]; std::swap(src[i], src[j]); std::swap(index[i], index[j]); }
    }
}
// The 6x6 tail of one Gauss-Newton iteration on its own (icet.cpp:410-430): out = cov[36] | pred_stds[6] | dx[6] | eigenvalues[6]; returns pruned axes, *rank = COD rank of HTWH.
int icet_oracle_gn_tail(const float* HTWH, const float* HTWdz, int libmf, float* out54, int32_t* rank) {
    ico::Mat H(6, 6), g(6, 1), cov, d; for (int i = 0; i < 36; i++) H.a[i] = HTWH[i]; for (int i = 0; i < 6; i++) g.a[i] = HTWdz[i];
    float ps[6], ev[6]; int pruned = 0, rk = 0;
    ico::gn_tail(H, g, libmf != 0, cov, ps, d, ev, &pruned, &rk);
    for (int i = 0; i < 36; i++) out54[i] = cov.a[i];
    for (int i = 0; i < 6; i++) { out54[36 + i] = ps[i]; out54[42 + i] = d.a[i]; out54[48 + i] = ev[i]; }
    if (rank) *rank = rk;
    return pruned;
}
void icet_oracle_get_H(const float mu[3], const float angs[3], float* H18) {
    ico::Mat H = ico::get_H(mu, angs); for (int i = 0; i < 18; i++) H18[i] = H.a[i];
}
void icet_oracle_R(const float angs[3], float* R9) {
    ico::Mat R = ico::eulerR(angs[0], angs[1], angs[2]); for (int i = 0; i < 9; i++) R9[i] = R.a[i];
}

}  // extern "C"
