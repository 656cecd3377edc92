// scripts/hip/lds_atomic_order.hip -- in which order does gfx950's LDS serve the lanes of ONE wave64 ds_add_rtn_u32 that hit the same
// address?  (round 4: the stable multi-splits of the keyframe build rank rows among the lanes holding the same class with one ballot per
// id bit -- ~10 VALU instructions per bit and round; if the returned pre-add values come back in ascending lane order, one LDS atomic per
// row gives the same rank.)  Every block of a full-chip launch runs R rounds: each lane draws a class from a per-round pattern (few / many
// distinct classes, runs, all-equal, packed 16-bit halves as the kernels use them), adds 1 to its wave's counter of that class and compares
// the value it got back with the ballot-computed count of earlier occurrences (earlier rounds + lower lanes of this round).
//   hipcc --offload-arch=gfx950 -O2 scripts/hip/lds_atomic_order.hip -o /tmp/lds_atomic_order && /tmp/lds_atomic_order
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int kClasses = 2048, kWaves = 4;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode: number of distinct classes drawn from (1 = every lane the same address)
__global__ __launch_bounds__(64 * kWaves) void k_order(int rounds, int distinct, int packed, unsigned long long* __restrict__ bad, unsigned long long* __restrict__ worst, uint32_t seed) {
    __shared__ uint32_t cnt[kWaves][kClasses];        // the counters the atomics hit (packed: two 16-bit halves per word)
    __shared__ uint16_t ref[kWaves][kClasses];        // what the ballots say
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kWaves * kClasses; i += blockDim.x) { (&cnt[0][0])[i] = 0u; (&ref[0][0])[i] = 0; }
    __syncthreads();
    unsigned long long nbad = 0ull;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int r = 0; r < rounds; r++) {
        uint32_t h = mix(seed ^ (uint32_t)(blockIdx.x * 7919 + r * 104729 + wave * 31));
        uint32_t c;
        if (r % 3 == 0) c = mix(h + lane) % (uint32_t)distinct;                 // independent draws
        else if (r % 3 == 1) c = (mix(h + (lane >> (h & 3))) % (uint32_t)distinct);   // runs of 1 / 2 / 4 / 8 equal neighbours
        else c = (mix(h + (lane & (int)(h >> 8 & 63))) % (uint32_t)distinct);    // lanes far apart share a class
        // reference rank: lanes of this round with the same class, by ballots per bit (11 bits)
        unsigned long long peers = ~0ull;
        for (int q = 0; q < 11; q++) { const bool bit = (c >> q) & 1u; const unsigned long long m = __ballot(bit); peers &= bit ? m : ~m; }
        const uint32_t before = ref[wave][c];
        const uint32_t want = before + (uint32_t)__popcll(peers & lt);
        uint32_t got;
        if (packed) { const uint32_t old = atomicAdd(&cnt[wave][c >> 1], 1u << (16u * (c & 1u))); got = (old >> (16u * (c & 1u))) & 0xFFFFu; }
        else got = atomicAdd(&cnt[wave][c], 1u);
        if (got != want) nbad++;
        if ((peers & lt) == 0ull) ref[wave][c] = (uint16_t)(before + (uint32_t)__popcll(peers));
        // (same wave, LDS in order: the next round's read of ref sees this write)
    }
    if (nbad) { atomicAdd(bad, nbad); atomicMax(worst, (unsigned long long)blockIdx.x); }
}

int main() {
    unsigned long long *d_bad, *d_worst;
    CK(hipMalloc(reinterpret_cast<void**>(&d_bad), 16)); d_worst = d_bad + 1;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    std::printf("%s, %d CUs\n", pr.gcnArchName, pr.multiProcessorCount);
    const int rounds = 15;                            // <= 15 x 64 = 960 per class: the packed halves do not carry
    unsigned long long total_bad = 0, total = 0;
    for (int packed = 0; packed < 2; packed++)
        for (int distinct : {1, 2, 3, 7, 16, 64, 600, 2048}) {
            unsigned long long bad_mode = 0;
            for (int rep = 0; rep < 20; rep++) {
                CK(hipMemset(d_bad, 0, 16));
                k_order<<<8192, 64 * kWaves>>>(rounds, distinct, packed, d_bad, d_worst, 0x9e3779b9u * (uint32_t)(rep + 1) + (uint32_t)distinct);
                CK(hipDeviceSynchronize());
                unsigned long long h[2]; CK(hipMemcpy(h, d_bad, 16, hipMemcpyDeviceToHost));
                bad_mode += h[0]; total += 8192ull * 64 * kWaves * rounds;
            }
            std::printf("packed %d  distinct classes %4d : %llu of %llu returned values differ from the lane-order rank\n", packed, distinct, bad_mode, 20ull * 8192 * 64 * kWaves * rounds);
            total_bad += bad_mode;
        }
    std::printf("TOTAL %llu mismatches in %llu atomics -> %s\n", total_bad, total, total_bad ? "NOT in lane order" : "lane order held everywhere");
    return 0;
}
