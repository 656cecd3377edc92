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
