// tests/cpp/adapter_demo.cpp -- compiles and runs include/icet.h (the adapter with the reference's class name, constructor signature
// and member names) through the call pattern of /root/reference/src/odometry.cpp:73-82: construct `ICET it(prev, cur, run_length, X0,
// numBinsPhi, numBinsTheta)`, read it.X and it.pred_stds, seed X0 for the next frame.  Built against tests/cpp/mock_eigen (this image
// has no Eigen; the mock pins no numerics) or against the real Eigen when one is installed (-DICET_TEST_REAL_EIGEN -I<eigen>).
// usage: adapter_demo scan1.f32 scan2.f32 n1 n2 [reps]   (files: column-major N x 3 float32; reps > 0 appends a timing loop: what one
//        constructor call costs a node, pageable Eigen matrices in, members out -- bench.py's "ctor" sub-record)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>
#include "icet.h"

static Eigen::MatrixXf slurp(const char* path, long n) {
    Eigen::MatrixXf m(n, 3);
    FILE* f = std::fopen(path, "rb");
    if (!f || std::fread(m.data(), sizeof(float), (size_t)n * 3, f) != (size_t)n * 3) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
    std::fclose(f);
    return m;
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    Eigen::MatrixXf prev_pcl_matrix = slurp(argv[1], std::atol(argv[3])), pcl_matrix = slurp(argv[2], std::atol(argv[4]));
    Eigen::VectorXf X0(6);
    X0 << 0., 0., 0., 0., 0., 0.;
    for (int frame = 0; frame < 2; frame++) {
        int run_length = 7;
        int numBinsPhi = 24;
        int numBinsTheta = 75;
        ICET it(prev_pcl_matrix, pcl_matrix, run_length, X0, numBinsPhi, numBinsTheta);
        if (it.status != ICET_OK) { std::fprintf(stderr, "status %d: %s\n", (int)it.status, it.error.c_str()); return 1; }
        Eigen::VectorXf X = it.X;
        std::printf("X %.9g %.9g %.9g %.9g %.9g %.9g\n", X[0], X[1], X[2], X[3], X[4], X[5]);
        std::printf("pred_stds %.9g %.9g %.9g %.9g %.9g %.9g\n", it.pred_stds[0], it.pred_stds[1], it.pred_stds[2], it.pred_stds[3], it.pred_stds[4], it.pred_stds[5]);
        std::printf("members %ld %ld %ld %ld %zu %zu %ld\n", it.clusterBounds.rows(), it.clusterBounds.cols(), it.points2.rows(), it.testPoints.rows(),
                    it.ellipsoid1Means.size(), it.ellipsoid2Means.size(), it.HTWH_i.rows());
        // the scan-1 voxel table as the reference exposes it (include/icet.h:89-94): maps keyed [theta][phi]
        size_t entries = 0; double tr = 0, lsum = 0, orth = 0;
        for (const auto& th : it.sigma1) for (const auto& ph : th.second) {
            entries++;
            const CovarianceMatrix& s = ph.second; const CovarianceMatrix& u = it.U[th.first][ph.first]; const CovarianceMatrix& l = it.L[th.first][ph.first];
            tr += s(0, 0) + s(1, 1) + s(2, 2); lsum += l(0, 0) + l(1, 1) + l(2, 2);
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { double d = 0; for (int k = 0; k < 3; k++) d += u(a, k) * u(b, k); orth += (d - (a == b)) * (d - (a == b)); }
        }
        const Eigen::Vector3f m0 = it.mu1.empty() ? Eigen::Vector3f() : it.mu1.begin()->second.begin()->second;
        std::printf("maps %zu %zu %zu %zu %zu %zu trace %.9g lsum %.9g orth %.3g first_mu %.9g %.9g %.9g key %d %d\n", entries, it.mu1.size(), it.U.size(), it.L.size(), it.sigma2.size(),
                    it.mu2.size(), tr, lsum, orth, m0[0], m0[1], m0[2], it.mu1.empty() ? -1 : it.mu1.begin()->first, it.mu1.empty() ? -1 : it.mu1.begin()->second.begin()->first);
        //seed initial estimate for next iteration
        X0 << X[0], X[1], X[2], X[3], X[4], X[5];
    }
    if (argc > 6) {            // the per-point members at side_tables_level 2 (include/icet.h:79,82,95-96 of the reference)
        ICET::side_tables_level() = 2;
        X0 << 0., 0., 0., 0., 0., 0.;
        ICET it(prev_pcl_matrix, pcl_matrix, 7, X0, 24, 75);
        ICET::side_tables_level() = 1;
        size_t e1 = 0, e2 = 0; long bad = 0;
        for (const auto& th : it.pointIndices1) for (const auto& ph : th) { e1 += ph.size(); for (size_t k = 1; k < ph.size(); k++) bad += ph[k] <= ph[k - 1]; }
        for (const auto& th : it.pointIndices2) for (const auto& ph : th) { e2 += ph.size(); for (size_t k = 1; k < ph.size(); k++) bad += ph[k] <= ph[k - 1]; }
        double rs = 0; for (long i = 0; i < it.points1Spherical.rows(); i++) rs += it.points1Spherical(i, 0);
        std::printf("perpoint %zu %zu %ld %ld %ld %.9g X %.9g\n", e1, e2, bad, it.points1Spherical.rows(), it.points2Spherical.rows(), rs, it.X[0]);
    }
    const int reps = argc > 5 ? std::atoi(argv[5]) : 0;
    if (reps > 0) {
        X0 << 0., 0., 0., 0., 0., 0.;
        std::vector<double> ms;
        double sink = 0;
        for (int r = 0; r < reps + 3; r++) {
            const auto t0 = std::chrono::steady_clock::now();
            ICET it(prev_pcl_matrix, pcl_matrix, 7, X0, 24, 75);
            const auto t1 = std::chrono::steady_clock::now();
            sink += it.X[0] + it.pred_stds[0];
            if (r >= 3) ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
        }
        std::sort(ms.begin(), ms.end());
        double mean = 0; for (double v : ms) mean += v;
        std::printf("ctor_timing reps %d mean_ms %.4f median_ms %.4f min_ms %.4f sink %.3g\n", reps, mean / ms.size(), ms[ms.size() / 2], ms[0], sink);
    }
    return 0;
}
