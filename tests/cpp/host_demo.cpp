// tests/cpp/host_demo.cpp -- C++ caller of the drop-in class, in the shape of the reference's demo harness
// (/root/reference/src/icet_cpp_demo.cpp:25-45: load two scans, construct ICET, print the solution and the wall time).
// usage: host_demo scan1.f32 scan2.f32 n1 n2 runlen bins_phi bins_theta      (files: column-major N x 3 float32)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "icet_host.hpp"

static std::vector<float> slurp(const char* path, size_t count) {
    std::vector<float> v(count);
    FILE* f = std::fopen(path, "rb");
    if (!f || std::fread(v.data(), sizeof(float), count, f) != count) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
    std::fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc < 8) { std::fprintf(stderr, "usage\n"); return 2; }
    const long n1 = std::atol(argv[3]), n2 = std::atol(argv[4]);
    const int run_length = std::atoi(argv[5]), numBinsPhi = std::atoi(argv[6]), numBinsTheta = std::atoi(argv[7]);
    std::vector<float> scan1 = slurp(argv[1], (size_t)n1 * 3), scan2 = slurp(argv[2], (size_t)n2 * 3);
    const float X0[6] = {0, 0, 0, 0, 0, 0};
    { icet_amd::ICET warm(scan1.data(), n1, n1, scan2.data(), n2, n2, 1, X0, numBinsPhi, numBinsTheta); if (warm.status != ICET_OK) { std::fprintf(stderr, "status %d: %s\n", (int)warm.status, warm.error.c_str()); return 1; } }
    auto before = std::chrono::steady_clock::now();
    icet_amd::ICET it(scan1.data(), n1, n1, scan2.data(), n2, n2, run_length, X0, numBinsPhi, numBinsTheta);
    auto after = std::chrono::steady_clock::now();
    if (it.status != ICET_OK) { std::fprintf(stderr, "status %d: %s\n", (int)it.status, it.error.c_str()); return 1; }
    std::printf("X %.9g %.9g %.9g %.9g %.9g %.9g\n", it.X[0], it.X[1], it.X[2], it.X[3], it.X[4], it.X[5]);
    std::printf("pred_stds %.9g %.9g %.9g %.9g %.9g %.9g\n", it.pred_stds[0], it.pred_stds[1], it.pred_stds[2], it.pred_stds[3], it.pred_stds[4], it.pred_stds[5]);
    std::printf("ellipsoids %zu bounds %zu points2 %zu\n", it.ellipsoid1Means.size(), it.clusterBounds.size(), it.points2.size());
    std::printf("Took: %.3f ms to register scans using ICET\n", std::chrono::duration<double, std::milli>(after - before).count());
    // errors come back as a status, never as an exception or abort
    icet_amd::ICET bad(scan1.data(), n1, n1, scan2.data(), n2, n2, run_length, X0, 0, numBinsTheta);
    std::printf("bad_status %d\n", (int)bad.status);
    return 0;
}
