// scripts/hip/stage_probe.hip -- where the time of a pageable-host -> HBM upload goes on the GPU box (round 4, constructor path):
// host memcpy rates into pageable / pinned memory by thread count, hipMemcpyAsync from pinned memory by chunk size and the runtime's own
// pageable path.  (The first version also timed a pinned chunk ring filled by copy threads -- the upload path round 4 built first:
// 92 - 258 us per scan against 35 us for ONE hipMemcpy2DAsync from pageable memory; profiles/r04_stage_probe.txt keeps those lines, the
// ring is not in the tree.)
//   hipcc --offload-arch=gfx950 -O2 scripts/hip/stage_probe.hip -o /tmp/stage_probe -lpthread && /tmp/stage_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const size_t n = 120876, col = n * sizeof(float), bytes = 3 * col;          // one 64-channel scan
    std::vector<float> src(3 * n, 1.5f), dst(3 * n);
    float *pin = nullptr, *dev = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void**>(&pin), bytes)); CK(hipMalloc(reinterpret_cast<void**>(&dev), bytes + 4096));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::printf("one scan = %.2f MB; host threads available: %u\n", bytes / 1e6, std::thread::hardware_concurrency());
    auto rate = [&](double us) { return bytes / us / 1e3; };
    auto best = [&](auto fn) { double b = 1e30; for (int r = 0; r < 20; r++) { for (size_t i = 0; i < 3 * n; i += 1024) src[i] += 1.f; const double t0 = now_us(); fn(); const double t = now_us() - t0; if (t < b) b = t; } return b; };
    double t = best([&] { std::memcpy(dst.data(), src.data(), bytes); });
    std::printf("memcpy pageable -> pageable, 1 thread:            %7.1f us  %5.1f GB/s\n", t, rate(t));
    t = best([&] { std::memcpy(pin, src.data(), bytes); });
    std::printf("memcpy pageable -> pinned (hipHostMalloc), 1 thr: %7.1f us  %5.1f GB/s\n", t, rate(t));
    for (int th : {2, 4, 8}) {
        t = best([&] { std::vector<std::thread> ts; const size_t per = bytes / th; for (int k = 0; k < th; k++) ts.emplace_back([&, k] { std::memcpy(reinterpret_cast<char*>(pin) + k * per, reinterpret_cast<char*>(src.data()) + k * per, per); }); for (auto& x : ts) x.join(); });
        std::printf("memcpy pageable -> pinned, %d fresh threads:       %7.1f us  %5.1f GB/s (includes thread start)\n", th, t, rate(t));
    }
    t = best([&] { (void)hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); });
    std::printf("hipMemcpyAsync pinned -> device, one call + sync:  %7.1f us  %5.1f GB/s\n", t, rate(t));
    for (size_t chunk : {(size_t)128 << 10, (size_t)256 << 10, (size_t)512 << 10}) {
        double issue = 0;
        t = best([&] { const double t0 = now_us(); for (size_t o = 0; o < bytes; o += chunk) (void)hipMemcpyAsync(reinterpret_cast<char*>(dev) + o, reinterpret_cast<char*>(pin) + o, bytes - o < chunk ? bytes - o : chunk, hipMemcpyHostToDevice, st); issue = now_us() - t0; (void)hipStreamSynchronize(st); });
        std::printf("hipMemcpyAsync pinned -> device, %3zu KB chunks:     %7.1f us  %5.1f GB/s (issue %.1f us)\n", chunk >> 10, t, rate(t), issue);
    }
    t = best([&] { (void)hipMemcpyAsync(dev, src.data(), bytes, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); });
    std::printf("hipMemcpyAsync PAGEABLE -> device (runtime path):  %7.1f us  %5.1f GB/s\n", t, rate(t));
    t = best([&] { (void)hipMemcpy2DAsync(dev, (n + 60) * 4, src.data(), col, col, 3, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); });
    std::printf("hipMemcpy2DAsync PAGEABLE -> device (round 3):     %7.1f us  %5.1f GB/s\n", t, rate(t));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    { const double t0 = now_us(); for (int r = 0; r < 100; r++) (void)hipEventRecord(ev, st); std::printf("hipEventRecord: %.2f us each\n", (now_us() - t0) / 100); (void)hipStreamSynchronize(st); }
    return 0;
}
