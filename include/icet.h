// include/icet.h -- drop-in replacement for the reference's include/icet.h (class ICET), backed by libicet_hip.so.
//
// Same class name, constructor signature, defaults and public data members as /root/reference/include/icet.h:36-116,
// so src/odometry.cpp:73-79, src/simpleMapMaker.cpp:113-122, src/scanMatcher.cpp:55-64 and the demos compile
// unchanged against it (they only construct the object and read X / pred_stds / points1 / points2 / clusterBounds /
// ellipsoid*).  The whole registration runs on the MI355X inside the constructor through icet_solve()
// (include/icet_hip.h); nothing of the reference's CPU implementation is reproduced here.
//
// Needs Eigen (as the reference does).  Eigen is not present in the build image of this repository, so this adapter
// is deliberately a thin copy-in / copy-out shim over the Eigen-free icet_amd::ICET of include/icet_host.hpp, which
// is what the test-suite compiles and runs (tests/cpp); the adapter itself is compiled and run against a minimal Eigen-API mock
// (tests/cpp/mock_eigen, tests/cpp/adapter_demo.cpp) so that it cannot rot.  The scan-1 voxel table is exposed as the reference
// exposes it: the std::map members mu1 / sigma1 / L / U keyed [theta][phi] (include/icet.h:27-29,89-94), one entry per fitted voxel,
// with U = eigenvectors^T as at src/icet.cpp:184 and L the 0/1 diagonal of src/icet.cpp:205-232; sigma2 / mu2 are declared and stay
// empty (the reference declares them and never fills them).  pointIndices1/2 and points1Spherical / points2Spherical -- indices into, and
// the rows of, the spherical copies the reference keeps for its own loops; no caller reads them -- are filled only when
// ICET::side_tables_level() is set to 2 (extra kernels + several MB of D2H per object).  points1Spherical / pointIndices1 are the reference's
// bit for bit (rows in the order its sort + swap loop leaves them); the scan-2 members are in the CALLER's row order, because the device never
// sorts scan 2 (the reference's are in its scrambled order: the same sets, permuted).  The reference's other public METHODS (fitScan1, prepScan2,
// fitScan2, sortSphericalCoordinates, findCluster, filterPointsInsideCluster, testSigmaPoints, fitCells1/2, parallelFitCells2, get_H,
// checkCondition: include/icet.h:46-68) are the stages of its constructor; here they run fused on the device inside the constructor and are
// not callable one by one -- no caller in the reference calls them from outside the class.  step() is kept (a stub there too).
// Two members differ in content, neither is read by any caller: `points2_OG` holds scan 2 as given (the reference stores it after
// its radial "sort" and a spherical round trip, src/icet.cpp:263-275: same points, permuted, <= 2 ulp away -- the device never
// sorts scan 2); `testPoints` holds the sigma points of the pruned axes like the reference's (src/icet.cpp:213-231) and ZEROS in the
// rows the reference leaves uninitialised.
#ifndef ICET_H
#define ICET_H

#include <Eigen/Dense>
#include <map>
#include <string>
#include <vector>
#include "icet_host.hpp"

using CovarianceMatrix = Eigen::Matrix3f;                                   // == Eigen::Matrix<float, 3, 3> (include/icet.h:27)
using CovarianceMap = std::map<int, std::map<int, CovarianceMatrix>>;      // include/icet.h:28
using MeanMap = std::map<int, std::map<int, Eigen::Vector3f>>;             // include/icet.h:29

class ICET {
public:
    ICET(Eigen::MatrixXf& scan1, Eigen::MatrixXf& scan2, int runlen, Eigen::VectorXf X0, int num_bins_phi, int num_bins_theta,
         int n = 25, float thresh = 0.1, float buff = 0.1)
        : rl(runlen), numBinsPhi(num_bins_phi), numBinsTheta(num_bins_theta), n(n), thresh(thresh), buff(buff), X(X0) {
        // Eigen::MatrixXf is column-major: data() is x[N] | y[N] | z[N] with leading dimension rows()
        float x0[6] = {0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 6 && k < X0.size(); k++) x0[k] = X0[k];
        // `points2` receives scan 2 under the last iteration's transform straight from the solve (no intermediate copy)
        points2.resize(scan2.rows(), 3);
        icet_amd::ICET it(icet_amd::ICET::Deferred{}, scan1.data(), scan1.rows(), scan1.rows(), scan2.data(), scan2.rows(), scan2.rows(), runlen, x0,
                          num_bins_phi, num_bins_theta, n, thresh, buff, 0, side_tables_level() >= 2 ? 2 : 1, scan2.rows() > 0 ? points2.data() : nullptr);
        // the deep copies the reference's constructor makes of both scans (src/icet.cpp:30,33) -- done while the device builds the keyframe
        points1 = scan1; points2_OG = scan2;
        // everything fitScan1 produces arrives while the Gauss-Newton loop still iterates on the device: converted to the reference's members here
        it.finish_keyframe();
        const int V = num_bins_phi * num_bins_theta;
        clusterBounds = Eigen::MatrixXf::Zero(V, 6);
        if ((int)it.clusterBounds.size() == V * 6)
            for (int v = 0; v < V; v++) for (int k = 0; k < 6; k++) clusterBounds(v, k) = it.clusterBounds[v * 6 + k];
        testPoints = Eigen::MatrixXf::Zero(V * 6, 3);                  // src/icet.cpp:41
        if ((int)it.testPoints.size() == V * 18)
            for (int v = 0; v < V; v++) if (it.has_fit[v])             // rows of un-fitted voxels stay zero
                for (int r = 6 * v; r < 6 * v + 6; r++) for (int k = 0; k < 3; k++) testPoints(r, k) = it.testPoints[r * 3 + k];
        for (size_t i = 0; i < it.ellipsoid1Means.size(); i++) {
            ellipsoid1Means.emplace_back(it.ellipsoid1Means[i][0], it.ellipsoid1Means[i][1], it.ellipsoid1Means[i][2]);
            Eigen::Matrix3f c;
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) c(a, b) = it.ellipsoid1Covariances[i][a * 3 + b];
            ellipsoid1Covariances.push_back(c);
            ellipsoid1Alphas.push_back(it.ellipsoid1Alphas[i]);
        }
        // mu1 / sigma1 / U / L keyed [theta][phi], one entry per fitted voxel (src/icet.cpp:177-184, 205-232)
        if ((int)it.has_fit.size() == V) {
            for (int v = 0; v < V; v++) if (it.has_fit[v]) {
                const int theta = v % num_bins_theta, phi = v / num_bins_theta;
                CovarianceMatrix s, u, l;
                for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
                    s(a, b) = it.sigma1[(size_t)v * 9 + a * 3 + b];
                    u(a, b) = it.evecs1[(size_t)v * 9 + b * 3 + a];                 // U = eigenvectors.transpose()
                    l(a, b) = (a == b) ? it.l_diag[(size_t)v * 3 + a] : 0.f;
                }
                sigma1[theta][phi] = s; U[theta][phi] = u; L[theta][phi] = l;
                mu1[theta][phi] = Eigen::Vector3f(it.mu1[(size_t)v * 3], it.mu1[(size_t)v * 3 + 1], it.mu1[(size_t)v * 3 + 2]);
            }
        }
        it.finish();
        if (it.status != ICET_OK || runlen <= 0) points2 = scan2;                 // an object that never iterated keeps its copy of scan 2
        X = Eigen::VectorXf(6); pred_stds = Eigen::VectorXf(6); dx = Eigen::VectorXf(6);
        for (int k = 0; k < 6; k++) { X[k] = it.X[k]; pred_stds[k] = it.pred_stds[k]; dx[k] = it.dx[k]; }
        HTWH_i.resize(6, 6); HTWdz_i.resize(6, 1);
        for (int a = 0; a < 6; a++) { HTWdz_i(a, 0) = it.HTWdz_i[a]; for (int b = 0; b < 6; b++) HTWH_i(a, b) = it.HTWH_i[a * 6 + b]; }
        // the per-point members (include/icet.h:79,82,95-96 of the reference), only at level 2
        if (side_tables_level() >= 2 && it.status == ICET_OK && runlen > 0) {
            const long N1 = scan1.rows(), N2 = scan2.rows();
            if ((long)it.points1Spherical.size() == N1 * 3) points1Spherical = Eigen::Map<const Eigen::MatrixXf>(it.points1Spherical.data(), N1, 3);
            if ((long)it.points2Spherical.size() == N2 * 3) points2Spherical = Eigen::Map<const Eigen::MatrixXf>(it.points2Spherical.data(), N2, 3);
            pointIndices1.assign(num_bins_theta, std::vector<std::vector<int>>(num_bins_phi));
            pointIndices2.assign(num_bins_theta, std::vector<std::vector<int>>(num_bins_phi));
            if ((int)it.binStart1.size() == V + 1)
                for (int v = 0; v < V; v++)
                    pointIndices1[v % num_bins_theta][v / num_bins_theta].assign(it.pointIndex1.begin() + it.binStart1[v], it.pointIndex1.begin() + it.binStart1[v + 1]);
            for (long i = 0; i < (long)it.voxel2.size(); i++) { const int v = it.voxel2[i]; if (v >= 0 && v < V) pointIndices2[v % num_bins_theta][v / num_bins_theta].push_back((int)i); }
        }
        status = it.status; error = it.error;
    }
    ~ICET() {}

    // 1 (default): every member a caller in the reference reads.  2: also points1Spherical / pointIndices1 / points2Spherical / pointIndices2,
    // which the reference keeps for its own loops and no caller reads (several MB more per object).  Process-wide, set before constructing.
    static int& side_tables_level() { static int level = 1; return level; }

    void step() { rl--; }      // the reference's stub prints "step", decrements rl and does nothing else (src/icet.cpp:438-441); silent here

    // algorithm params
    int rl; int numBinsPhi; int numBinsTheta; int n; float thresh; float buff;

    Eigen::MatrixXf points1, points2, points2_OG, clusterBounds, testPoints, HTWH_i, HTWdz_i;
    Eigen::MatrixXf points1Spherical, points2Spherical;                      // filled at side_tables_level() >= 2
    std::vector<std::vector<std::vector<int>>> pointIndices1, pointIndices2; // [theta][phi] -> ascending row indices; level 2
    Eigen::VectorXf pred_stds;
    Eigen::VectorXf X;    // global solution vector (x, y, z, roll, pitch, yaw)
    Eigen::VectorXf dx;   // last linear perturbation

    CovarianceMap sigma1, sigma2, L, U;      // [theta][phi]; sigma2 / mu2 stay empty as in the reference
    MeanMap mu1, mu2;

    // for viz
    std::vector<Eigen::Vector3f> ellipsoid1Means;
    std::vector<Eigen::Matrix3f> ellipsoid1Covariances;
    std::vector<float> ellipsoid1Alphas;
    std::vector<Eigen::Vector3f> ellipsoid2Means;            // always empty, as in the reference (src/icet.cpp:49-61)
    std::vector<Eigen::Matrix3f> ellipsoid2Covariances;
    std::vector<float> ellipsoid2Alphas;

    icet_status status = ICET_OK;   // additions: the reference signals no errors
    std::string error;
};

#endif
