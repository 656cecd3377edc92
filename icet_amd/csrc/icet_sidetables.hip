// icet_amd/csrc/icet_sidetables.hip -- the per-POINT members of the reference object that no caller in the reference reads but that are
// part of `class ICET`'s public surface (/root/reference/include/icet.h:79,82,95-96): points1Spherical, pointIndices1, points2Spherical,
// pointIndices2.  Produced on request only (icet_aux::points1_spherical ...), after the solve, from what the keyframe build left in
// the workspace; the default constructor path does not pay for them.
//     k_side_scan1    points1Spherical (src/icet.cpp:69-83): (r, theta, phi) of the row that sits at every position after the
//                     reference's sort + swap loop, under the shared arithmetic rule (bit for bit the oracle's c2s); and the inverse
//                     of src[] (position of every original row)
//     k_side_index1   pointIndices1 (src/icet.cpp:86, 534-554): the POSITIONS of every voxel's rows, ascending, concatenated in voxel
//                     order (the sorted-row table holds original rows; positions are what the reference stores)
//     k_side_scan2    points2 / points2Spherical of the LAST fitScan2 (src/icet.cpp:375-388) and the voxel
//                     sortSphericalCoordinates assigns to every row -- in the CALLER's row order: the device never sorts scan 2
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include "icet_internal.h"
#include "icet_device_common.h"

namespace icet {
namespace {

__global__ __launch_bounds__(kBlock) void k_side_scan1(const PairDesc* __restrict__ desc, const int32_t* __restrict__ src, uint32_t* __restrict__ inv,
                                                       float* __restrict__ sph) {
    const PairDesc d = desc[0];
    const float* px = d.s1; const float* py = px + d.ld1; const float* pz = px + 2 * (size_t)d.ld1;
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < d.n1; p += gridDim.x * kBlock) {
        const int row = src[(size_t)d.off1 + p];
        inv[(size_t)d.off1 + row] = (uint32_t)p;
        float r, th, ph;
        c2s_cr(px[row], py[row], pz[row], r, th, ph);
        sph[p] = r; sph[(size_t)d.n1 + p] = th; sph[2 * (size_t)d.n1 + p] = ph;
    }
}

__global__ __launch_bounds__(kBlock) void k_side_index1(const PairDesc* __restrict__ desc, const uint32_t* __restrict__ sorted_row, const uint32_t* __restrict__ inv,
                                                        int32_t* __restrict__ out) {
    const PairDesc d = desc[0];
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < d.n1; k += gridDim.x * kBlock)
        out[k] = (int32_t)inv[(size_t)d.off1 + (sorted_row[(size_t)d.off1 + k] & kSortedRowMask)];
}

__global__ __launch_bounds__(kBlock) void k_side_scan2(const PairDesc* __restrict__ desc, const float* __restrict__ xf, float* __restrict__ pts, float* __restrict__ sph,
                                                       int32_t* __restrict__ voxel, int T, int P) {
    __shared__ float sxf[12];
    if (threadIdx.x < 12) sxf[threadIdx.x] = xf[threadIdx.x];
    __syncthreads();
    const PairDesc d = desc[0];
    const float* px = d.s2; const float* py = px + d.ld2; const float* pz = px + 2 * (size_t)d.ld2;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.n2; i += gridDim.x * kBlock) {
        float qx, qy, qz, r, th, ph;
        transform_point(px[i], py[i], pz[i], sxf, qx, qy, qz);
        c2s_cr(qx, qy, qz, r, th, ph);
        pts[i] = qx; pts[(size_t)d.n2 + i] = qy; pts[2 * (size_t)d.n2 + i] = qz;
        if (sph) { sph[i] = r; sph[(size_t)d.n2 + i] = th; sph[2 * (size_t)d.n2 + i] = ph; }
        if (voxel) voxel[i] = voxel_of(th, ph, T, P);
    }
}

}  // namespace

#define ICET_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t launch_side_scan1(const Workspace& w, const LaunchCfg& c, float* sph, int32_t* index, hipStream_t st) {
    if (c.max_n1 <= 0) return hipSuccess;
    const int blocks = std::min(2048, (c.max_n1 + kBlock - 1) / kBlock);
    k_side_scan1<<<blocks, kBlock, 0, st>>>(w.desc, w.src, w.keyB, sph);          // keyB: keyframe scratch, dead once the keyframe is built
    ICET_LAUNCH_CHECK();
    k_side_index1<<<blocks, kBlock, 0, st>>>(w.desc, w.valA, w.keyB, index);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_side_scan2(const Workspace& w, const LaunchCfg& c, const float* xf, float* pts, float* sph, int32_t* voxel, hipStream_t st) {
    if (c.max_n2 <= 0) return hipSuccess;
    const int blocks = std::min(2048, (c.max_n2 + kBlock - 1) / kBlock);
    k_side_scan2<<<blocks, kBlock, 0, st>>>(w.desc, xf, pts, sph, voxel, c.T, c.P);
    ICET_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace icet
