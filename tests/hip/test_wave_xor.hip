// Stand-alone check of wave_xor<D> (icet_device_common.h) against __shfl_xor, and of the DPP scans against a serial loop.
// Run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -I icet_amd/csrc -I include tests/hip/test_wave_xor.hip -o /tmp/t && /tmp/t
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "icet_device_common.h"
using namespace icet;
__global__ void k(const uint32_t* in, uint32_t* out) {
    const uint32_t x = in[threadIdx.x];
    uint32_t bad = 0;
    bad |= (wave_xor<1>(x) != (uint32_t)__shfl_xor((int)x, 1)) ? 1u : 0u;
    bad |= (wave_xor<2>(x) != (uint32_t)__shfl_xor((int)x, 2)) ? 2u : 0u;
    bad |= (wave_xor<4>(x) != (uint32_t)__shfl_xor((int)x, 4)) ? 4u : 0u;
    bad |= (wave_xor<8>(x) != (uint32_t)__shfl_xor((int)x, 8)) ? 8u : 0u;
    bad |= (wave_xor<16>(x) != (uint32_t)__shfl_xor((int)x, 16)) ? 16u : 0u;
    bad |= (wave_xor<32>(x) != (uint32_t)__shfl_xor((int)x, 32)) ? 32u : 0u;
    int ref = 0; for (int o = 0; o <= (int)(threadIdx.x & 63); o++) ref += (int)(__shfl((int)x, o) & 1023);
    bad |= (wave_incl_sum((int)(x & 1023)) != ref) ? 64u : 0u;
    bad |= (wave_reduce_max(x) != (uint32_t)__reduce_max_sync(~0ull, x)) ? 128u : 0u;
    // the remaining helpers: totals against a serial sum of exactly representable values, the other reductions against the
    // library's, the one-lane shifts against __shfl_up / __shfl_down with their fill values
    const int lane = threadIdx.x & 63;
    {
        int si = 0; float sf = 0.f; double sd = 0.0;
        for (int o = 0; o < 64; o++) { const int v = (int)(__shfl((int)x, o) & 4095) - 2048; si += v; sf += (float)v; sd += (double)v * 0.5; }   // |sum| < 2^18: exact in float
        const int mine = (int)(x & 4095) - 2048;
        bad |= (wave_total(mine) != si) ? 256u : 0u;
        bad |= (wave_total((float)mine) != sf) ? 512u : 0u;
        bad |= (wave_total((double)mine * 0.5) != sd) ? 1024u : 0u;
    }
    bad |= (wave_reduce_or(x) != (uint32_t)__reduce_or_sync(~0ull, x)) ? 2048u : 0u;
    bad |= (wave_reduce_and(x | 0xFFFF0000u) != (uint32_t)__reduce_and_sync(~0ull, x | 0xFFFF0000u)) ? 4096u : 0u;
    bad |= (wave_reduce_min(x) != (uint32_t)__reduce_min_sync(~0ull, x)) ? 8192u : 0u;
    bad |= (wave_reduce_min_i((int)x) != __reduce_min_sync(~0ull, (int)x)) ? 16384u : 0u;
    bad |= (wave_reduce_max_i((int)x) != __reduce_max_sync(~0ull, (int)x)) ? 32768u : 0u;
    {
        const int up = __shfl_up((int)x, 1), dn = __shfl_down((int)x, 1);
        bad |= (wave_shr1((int)x, -7) != (lane == 0 ? -7 : up)) ? 65536u : 0u;
        bad |= (wave_shr1_zero((int)x) != (lane == 0 ? 0 : up)) ? 131072u : 0u;
        bad |= (wave_shl1((int)x, -2) != (lane == 63 ? -2 : dn)) ? 262144u : 0u;
        bad |= (wave_shr1(__int_as_float((int)x & 0x3FFFFFFF), 1.5f) != (lane == 0 ? 1.5f : __int_as_float(up & 0x3FFFFFFF))) ? 524288u : 0u;
        bad |= (wave_read((int)x, 17) != __shfl((int)x, 17)) ? 1048576u : 0u;
    }
    out[threadIdx.x + blockDim.x * blockIdx.x] = bad;
}
int main() {
    uint32_t h[256], *d_in, *d_out, r[256];
    for (int i = 0; i < 256; i++) h[i] = 2654435761u * (uint32_t)(i + 17);
    hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, sizeof(h));
    hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 256>>>(d_in, d_out);
    hipMemcpy(r, d_out, sizeof(r), hipMemcpyDeviceToHost);
    uint32_t any = 0; for (int i = 0; i < 256; i++) any |= r[i];
    printf("wave_xor / scans: %s (mask %u)\n", any ? "MISMATCH" : "ok", any);
    return any ? 1 : 0;
}
