// icet_amd/csrc/icet_device_common.h -- device helpers shared by the keyframe, accumulate and solve translation units.
//
// THE SHARED ARITHMETIC RULE (the CPU checker used by the tests states the same rule independently).  The reference takes its
// float atan2 / acos / sin / cos from whatever glibc it is built against (accurate to an ulp, not correctly rounded) and adds
// up rows in Eigen's unspecified vectorised order.  Both reach the per-voxel covariance in its last bits, and the result
// depends on those bits through the SIGNS of the scan-1 eigenvectors.  Device and oracle therefore follow one
// rule that is a mathematical statement, not shared code:
//   * every transcendental that feeds a stored value is the CORRECTLY ROUNDED float of the exact function value: evaluated
//     in double, rounded once (theta_cr, phi_cr, roundtrip_cr below; glibc's double functions in the CPU checker);
//   * per-voxel sums are exact (double accumulators over float addends), rounded once, then divided in float;
//   * no contraction where the oracle has separate roundings (#pragma clang fp contract(off)).
// Decisions (which voxel, inside the bounds or not) are comparisons; they are taken on cheap monotone stand-ins with guard
// bands and fall back to these formulas near an edge, so they equal the literal decisions everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "icet_internal.h"

namespace icet {

constexpr int kBlock = 256;
constexpr double kTwoPi = 6.283185307179586476925286766559;      // == 2.0 * M_PI as a double, the reference's constant (src/utils.cpp:105)
constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPiTail = 2.4492935982947064e-16;             // (exact 2 pi) - kTwoPi

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// 1-D grid -> (pair, chunk).  With >= 8 pairs every chunk of a pair gets the same blockIdx % 8, i.e. (as the
// dispatcher is observed to deal blocks round-robin over the 8 XCDs) the same XCD and the same 4 MiB L2, and
// consecutive block ids walk through ONE group of 8 pairs before touching the next: the pointer-chasing keyframe
// kernels (rank / scramble / gather, ~1 MB of randomly accessed tables per pair) then find their pair's tables in
// L2 instead of HBM.  Speed only -- nothing depends on where a block actually lands.
__device__ __forceinline__ bool decode_block(int n_pairs, int chunks, int& pair, int& chunk) {
    if (n_pairs >= 8) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        pair = (j / chunks) * 8 + xcd; chunk = j % chunks;
        return pair < n_pairs;
    }
    pair = blockIdx.x / chunks; chunk = blockIdx.x % chunks;
    return true;
}
inline int grid_groups(int n_pairs) { return n_pairs >= 8 ? (n_pairs + 7) / 8 * 8 : n_pairs; }

// r of utils::cartesianToSpherical (src/utils.cpp:99, rowwise().norm()) BEFORE the NaN -> 1000 replacement: no contraction,
// correctly rounded sqrt, so its bits match a plain IEEE evaluation -- the radial sort + swap loop downstream is chaotic in
// the order of r.
__device__ __forceinline__ float radius_raw(float x, float y, float z) {
    float r;
    {
#pragma clang fp contract(off)
        float s = x * x + y * y;
        s = s + z * z;
        r = sqrtf(s);
    }
    return r;
}

// theta and phi of utils::cartesianToSpherical (src/utils.cpp:103-108,116) under the shared rule.
__device__ __forceinline__ float theta_cr(float y, float x) {
    float th = (float)atan2((double)y, (double)x);
    if (th < 0.0f) th = (float)((double)th + kTwoPi);
    return (th != th) ? 1000.0f : th;
}
__device__ __forceinline__ float phi_cr(float z, float r_raw) {
    const float ph = (float)acos((double)(z / r_raw));
    return (ph != ph) ? 1000.0f : ph;
}
__device__ __forceinline__ void c2s_cr(float x, float y, float z, float& r, float& th, float& ph) {
    const float rr = radius_raw(x, y, z);
    th = theta_cr(y, x); ph = phi_cr(z, rr);
    r = (rr != rr) ? 1000.0f : rr;
}

// sortSphericalCoordinates' bin of one (theta, phi) pair: double arithmetic on float angles, truncation, modulo
// (src/icet.cpp:545-546).
__device__ __forceinline__ int voxel_of(float th, float ph, int T, int P) {
    int bt = static_cast<int>(((double)th / kTwoPi) * (double)T) % T;
    int bp = static_cast<int>(((double)ph / kPi) * (double)P) % P;
    return T * bp + bt;
}

__device__ __forceinline__ bool inside_bounds(float r, float az, float el, float az0, float az1, float el0, float el1, float inner, float outer) {
    return az >= az0 && az <= az1 && el >= el0 && el <= el1 && r >= inner && r <= outer;
}

// clusterBounds' angular limits of voxel (theta, phi): (float / int) -> float, times a double constant, stored to float
// (src/icet.cpp:136-139).
__device__ __forceinline__ void voxel_limits(int theta, int phi, int T, int P, float& az0, float& az1, float& el0, float& el1) {
    az0 = (float)((double)((float)theta / (float)T) * kTwoPi);
    az1 = (float)((double)((float)(theta + 1) / (float)T) * kTwoPi);
    el0 = (float)((double)((float)phi / (float)P) * kPi);
    el1 = (float)((double)((float)(phi + 1) / (float)P) * kPi);
}

// cartesianToSpherical followed by sphericalToCartesian (src/utils.cpp:93-142) for one finite point with r > 0, under the
// shared rule: th = fl(atan2), ph = fl(acos(z / r)), out = (r sin ph cos th, r sin ph sin th, r cos ph) with the four
// sines / cosines correctly rounded -- WITHOUT a double sin / cos.  With a_d the double angle and a_f its float rounding
// (plus the wrap of theta), sin a_f = sin(a_d + d) = sin a_d (1 - d^2/2) + cos a_d d, and sin a_d, cos a_d are algebraic in
// the Cartesian inputs (y / rho, x / rho; sqrt(1 - q^2), q).  |d| < 5e-7, so the dropped d^3 term is below 1e-20.
__device__ __forceinline__ void roundtrip_cr(float x, float y, float z, float r_raw, float& th, float& ph, float& ox, float& oy, float& oz) {
    const double xd = (double)x, yd = (double)y;
    const double td = atan2(yd, xd);
    const float t0 = (float)td;
    const bool wrap = t0 < 0.0f;
    th = wrap ? (float)((double)t0 + kTwoPi) : t0;
    const double dth = wrap ? ((((double)th - kTwoPi) - td) - kTwoPiTail) : ((double)th - td);
    const double rho2 = xd * xd + yd * yd;                       // both products exact in double
    const double inv = 1.0 / sqrt(rho2);
    const double s0 = (rho2 > 0.0) ? yd * inv : 0.0;
    const double c0 = (rho2 > 0.0) ? xd * inv : (__builtin_signbit(x) ? -1.0 : 1.0);
    const double hth = 0.5 * dth * dth;
    const double st = s0 + (c0 * dth - s0 * hth);
    const double ct = c0 - (s0 * dth + c0 * hth);
    const double qd = (double)(z / r_raw);                       // float division, as the reference's acos argument
    const double pd = acos(qd);
    ph = (float)pd;
    const double dph = (double)ph - pd;
    const double sp0 = sqrt((1.0 - qd) * (1.0 + qd));            // sin(acos q) >= 0; 1 -+ q exact for float q
    const double hph = 0.5 * dph * dph;
    const double sp = sp0 + (qd * dph - sp0 * hph);
    const double cp = qd - (sp0 * dph + qd * hph);
    const float stf = (float)st, ctf = (float)ct, spf = (float)sp, cpf = (float)cp;
    {
#pragma clang fp contract(off)
        ox = r_raw * spf * ctf;                                  // src/utils.cpp:134-136, left to right in float
        oy = r_raw * spf * stf;
        oz = r_raw * cpf;
    }
}

// The same round trip for ANY row (ICET_FLAG_ROUNDTRIP_SCAN2): an ordinary point takes roundtrip_cr; a zero / non-finite row goes through
// the literal sentinel rules (NaN -> 1000, src/utils.cpp:116) and double-precision sin / cos of the float angles.
__device__ __noinline__ void roundtrip_odd(float x, float y, float z, float& ox, float& oy, float& oz) {
    float r, th, ph;
    c2s_cr(x, y, z, r, th, ph);
    const float spf = (float)sin((double)ph), cpf = (float)cos((double)ph), stf = (float)sin((double)th), ctf = (float)cos((double)th);
    {
#pragma clang fp contract(off)
        ox = r * spf * ctf; oy = r * spf * stf; oz = r * cpf;
    }
}
__device__ __forceinline__ void roundtrip_any(float x, float y, float z, float& ox, float& oy, float& oz) {
    const float rr = radius_raw(x, y, z);
    if (rr > 0.f && rr < INFINITY) { float th, ph; roundtrip_cr(x, y, z, rr, th, ph, ox, oy, oz); }
    else roundtrip_odd(x, y, z, ox, oy, oz);
}

// ---- fast angular classification -------------------------------------------------------------------------------------
// Every decision the reference takes on a point's angles is a comparison against a voxel edge: the azimuth bin
// int(theta / 2pi * T) in double (src/icet.cpp:545), the polar bin, and the f32 azimuth / polar bounds (:632-633, Q6).  Two
// monotone, transcendental-free coordinates stand in for the angles --
//     polar   : w  = -z / |q|                      (monotone in phi   = acos(z/|q|))
//     azimuth : pa = "diamond angle" of (x, y)     (monotone in theta = atan2(y, x)), unfolded to [0, 4]
// -- and are looked up in tables whose cells are narrower than half a bin: a cell names the single edge a point in it can be
// near, one compare picks the side.  `near` is set when the point lies within a guard band of that edge (a few float ulps:
// the summed worst-case rounding of this evaluation and of the literal one), when the cell is marked ambiguous (edge = NaN,
// near the poles), or when the coordinates are not ordinary numbers; such a point must be classified with the literal
// formulas.  Away from the edges the azimuth / polar bounds of filterPointsInsideCluster hold by construction.  Invariant,
// tested bitwise (test_fast_classification_equals_literal_evaluation): the fast path never decides differently.
struct LutCell { float edge; int32_t idx; };       // nearest edge (in pa / w units) and its index (polar: T * index)

__device__ __forceinline__ void classify_angular_fast(float qx, float qy, float qz, float rs /* 1/|q| */, const LutCell* lut_t, const LutCell* lut_p,
                                                      float cell_t, float cell_p, int T, float guard_t, float guard_p, int& bt, int& prow, bool& near) {
    const float w = -qz * rs;                                            // -cos(phi)
    const float q1 = qy * __builtin_amdgcn_rcpf(fabsf(qx) + fabsf(qy));  // y / (|x| + |y|) in [-1, 1]
    const float pa = (qx >= 0.f) ? ((qy >= 0.f) ? q1 : 4.f + q1) : 2.f - q1;      // diamond angle in [0, 4]
    const LutCell et = lut_t[static_cast<int>(pa * cell_t)];             // NaN converts to 0; pa in [0,4] -> cell in [0, Mt]
    const LutCell ep = lut_p[static_cast<int>((w + 1.f) * cell_p)];
    bt = et.idx - ((pa < et.edge) ? 1 : 0);                              // in [0, T]
    prow = ep.idx - ((w < ep.edge) ? T : 0);                             // T * polar bin, polar bin in [0, P]
    near = !(fabsf(pa - et.edge) >= guard_t) | !(fabsf(w - ep.edge) >= guard_p);
}
// ---- cross-lane moves on the DPP path (no LDS crossbar: __shfl_* compiles to ds_bpermute_b32, a trip through the LDS pipeline) --------
// lane i receives x of lane i - 1; lane 0 receives `fill`
__device__ __forceinline__ int wave_shr1(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ float wave_shr1(float x, float fill) { return __int_as_float(wave_shr1(__float_as_int(x), __float_as_int(fill))); }
// the same with 0 for lane 0, through DPP's bound control: the destination needs no prior value, i.e. no v_mov per call
__device__ __forceinline__ int wave_shr1_zero(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true); }
__device__ __forceinline__ float wave_shr1_zero(float x) { return __int_as_float(wave_shr1_zero(__float_as_int(x))); }
// lane i receives x of lane i + 1; lane 63 receives `fill`
__device__ __forceinline__ int wave_shl1(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }
// Sum over the 64 lanes on the DPP path (the __shfl_xor butterflies above are ds_bpermute_b32: two trips through the LDS pipeline per
// step of a double): an inclusive scan -- four row_shr steps inside the rows of 16, the last lane of rows 0 / 2 into rows 1 / 3, the last
// lane of row 1 into rows 2 and 3 -- whose lane 63 holds the total, read back with v_readlane.  Every lane gets the same bits.
template <int kCtrl, int kRowMask> __device__ __forceinline__ int dpp_zero_fill(int v) { return __builtin_amdgcn_update_dpp(0, v, kCtrl, kRowMask, 0xf, false); }
__device__ __forceinline__ float wave_total(float v) {
    v += __int_as_float(dpp_zero_fill<0x111 /* row_shr:1 */, 0xf>(__float_as_int(v)));
    v += __int_as_float(dpp_zero_fill<0x112 /* row_shr:2 */, 0xf>(__float_as_int(v)));
    v += __int_as_float(dpp_zero_fill<0x114 /* row_shr:4 */, 0xf>(__float_as_int(v)));
    v += __int_as_float(dpp_zero_fill<0x118 /* row_shr:8 */, 0xf>(__float_as_int(v)));
    v += __int_as_float(dpp_zero_fill<0x142 /* row_bcast:15 */, 0xa>(__float_as_int(v)));
    v += __int_as_float(dpp_zero_fill<0x143 /* row_bcast:31 */, 0xc>(__float_as_int(v)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_total(int v) {
    v += dpp_zero_fill<0x111, 0xf>(v); v += dpp_zero_fill<0x112, 0xf>(v); v += dpp_zero_fill<0x114, 0xf>(v); v += dpp_zero_fill<0x118, 0xf>(v);
    v += dpp_zero_fill<0x142, 0xa>(v); v += dpp_zero_fill<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}
template <int kCtrl, int kRowMask> __device__ __forceinline__ double dpp_zero_fill_d(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)dpp_zero_fill<kCtrl, kRowMask>((int)(unsigned)b), hi = (unsigned)dpp_zero_fill<kCtrl, kRowMask>((int)(unsigned)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_total(double v) {
    v += dpp_zero_fill_d<0x111, 0xf>(v); v += dpp_zero_fill_d<0x112, 0xf>(v); v += dpp_zero_fill_d<0x114, 0xf>(v); v += dpp_zero_fill_d<0x118, 0xf>(v);
    v += dpp_zero_fill_d<0x142, 0xa>(v); v += dpp_zero_fill_d<0x143, 0xc>(v);
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The same scan for any associative operation on 32-bit words (`id` = its identity, what a lane without a source contributes):
// every lane ends with the inclusive prefix of lanes 0 .. its own; wave_reduce_* read the total back from lane 63.
template <typename Op> __device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t id, Op op) {
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, false));
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, false));
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, false));
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
    return v;
}
__device__ __forceinline__ int wave_incl_sum(int v) { return (int)wave_incl_scan((uint32_t)v, 0u, [](uint32_t a, uint32_t b) { return a + b; }); }
__device__ __forceinline__ uint32_t wave_reduce_or(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0u, [](uint32_t a, uint32_t b) { return a | b; }), 63); }
__device__ __forceinline__ uint32_t wave_reduce_and(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0xFFFFFFFFu, [](uint32_t a, uint32_t b) { return a & b; }), 63); }
__device__ __forceinline__ uint32_t wave_reduce_min(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0xFFFFFFFFu, [](uint32_t a, uint32_t b) { return a < b ? a : b; }), 63); }
__device__ __forceinline__ uint32_t wave_reduce_max(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0u, [](uint32_t a, uint32_t b) { return a > b ? a : b; }), 63); }
__device__ __forceinline__ int wave_reduce_min_i(int v) { return __builtin_amdgcn_readlane((int)wave_incl_scan((uint32_t)v, 0x7FFFFFFFu, [](uint32_t a, uint32_t b) { return (uint32_t)min((int)a, (int)b); }), 63); }
__device__ __forceinline__ int wave_reduce_max_i(int v) { return __builtin_amdgcn_readlane((int)wave_incl_scan((uint32_t)v, 0x80000000u, [](uint32_t a, uint32_t b) { return (uint32_t)max((int)a, (int)b); }), 63); }
// x of lane `src`, src wave-uniform: v_readlane_b32 instead of a ds_bpermute_b32
__device__ __forceinline__ int wave_read(int x, int src) { return __builtin_amdgcn_readlane(x, __builtin_amdgcn_readfirstlane(src)); }
__device__ __forceinline__ float wave_read(float x, int src) { return __int_as_float(wave_read(__float_as_int(x), src)); }

// x of lane (lane ^ kD), kD a power of two below 64, without the LDS crossbar: quad permutes for 1 and 2, a pair of row shifts for
// 4 and 8 (the partner is kD lanes up or down inside the row of 16), gfx950's v_permlane16_swap / v_permlane32_swap for 16 and 32
// (with both operands = x they return [x0 x0 x2 x2] and [x1 x1 x3 x3] by rows of 16, resp. halves: the partner row is in one of them).
template <int kD> __device__ __forceinline__ uint32_t wave_xor(uint32_t x) {
    static_assert(kD == 1 || kD == 2 || kD == 4 || kD == 8 || kD == 16 || kD == 32, "partner distance");
    const int lane = (int)(threadIdx.x & 63u);
    if constexpr (kD == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true);
    else if constexpr (kD == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, true);
    else if constexpr (kD == 4 || kD == 8) {
        const int up = __builtin_amdgcn_update_dpp(0, (int)x, 0x100 + kD /* row_shl: lane i <- i + kD */, 0xf, 0xf, true);
        const int dn = __builtin_amdgcn_update_dpp(0, (int)x, 0x110 + kD /* row_shr: lane i <- i - kD */, 0xf, 0xf, true);
        return (uint32_t)((lane & kD) ? dn : up);
    } else if constexpr (kD == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return (lane & 32) ? r[0] : r[1];
    }
}

// Wave totals of N doubles at once (N = 4 or 8), as a reduce-scatter: in the step with partner lane ^ D a lane keeps half of its list and
// hands the other half over, so the list halves while the lanes it spans double -- N / 2 + N / 4 + .. + 1 exchanges and then one value over the
// remaining distances, instead of N x 6 (a double costs two DPP moves per exchange: wave_total of nine doubles was 3/4 of k_fit_moments).
// Afterwards the total of v[j] sits in every lane whose low log2(N) bits are j's bits reversed: wave_total_scatter_get reads it from there.
template <int kD> __device__ __forceinline__ double wave_xor_d(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const uint32_t lo = wave_xor<kD>((uint32_t)b), hi = wave_xor<kD>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int N> __device__ __forceinline__ double wave_total_scatter(double (&v)[N]) {
    static_assert(N == 4 || N == 8, "list length");
    const int lane = (int)(threadIdx.x & 63u);
    {
        const bool up = lane & 1;
#pragma unroll
        for (int i = 0; i < N / 2; i++) { const double keep = up ? v[i + N / 2] : v[i], send = up ? v[i] : v[i + N / 2]; v[i] = keep + wave_xor_d<1>(send); }
    }
    {
        const bool up = lane & 2;
#pragma unroll
        for (int i = 0; i < N / 4; i++) { const double keep = up ? v[i + N / 4] : v[i], send = up ? v[i] : v[i + N / 4]; v[i] = keep + wave_xor_d<2>(send); }
    }
    if constexpr (N == 8) {
        const bool up = lane & 4;
        const double keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
        v[0] = keep + wave_xor_d<4>(send);
    } else {
        v[0] += wave_xor_d<4>(v[0]);
    }
    v[0] += wave_xor_d<8>(v[0]); v[0] += wave_xor_d<16>(v[0]); v[0] += wave_xor_d<32>(v[0]);
    return v[0];
}
template <int N> __device__ __forceinline__ double wave_total_scatter_get(double mine, int j) {      // j: compile-time index into the list handed to wave_total_scatter<N>
    const int src = (N == 8) ? (((j >> 2) & 1) | (((j >> 1) & 1) << 1) | ((j & 1) << 2)) : (((j >> 1) & 1) | ((j & 1) << 1));
    const unsigned long long b = (unsigned long long)__double_as_longlong(mine);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// inclusive prefix maximum over the 64 lanes, for values >= -1 (-1 = "nothing"): four row_shr steps inside the rows of 16, then the
// last lane of row 0 / 2 into rows 1 / 3 and the last lane of row 1 into rows 2 and 3
__device__ __forceinline__ int wave_incl_max(int v) {
    // on v + 1 >= 0 as UNSIGNED values: a lane without a source then contributes 0, the identity, and every step is ONE v_max_u32_dpp
    // (with the signed identity -1 the compiler keeps v_mov -1 / v_mov_dpp / v_max per step: 18 instructions instead of 6)
    return (int)wave_incl_scan((uint32_t)(v + 1), 0u, [](uint32_t a, uint32_t b) { return a > b ? a : b; }) - 1;
}

// ---- the literal path for points the fast classification cannot decide ------------------------------------------------
// sortSphericalCoordinates' bin index WITHOUT the double divide: thr[k] is the smallest float whose reference bin (double
// arithmetic, src/icet.cpp:545-546) is >= k, built on the host with exactly that arithmetic, so "largest k with
// thr[k] <= a" is bit-for-bit the reference's truncation.  The float product only proposes a candidate (off by at most
// one).  Returns nb when a lies beyond the last edge (a == float(2 pi), float(pi) or the 1000 sentinel): the caller then
// takes the literal formula.
__device__ __forceinline__ int bin_from_table(float a, const float* __restrict__ thr, int nb, float scale) {
    int k = static_cast<int>(a * scale);
    k = min(k, nb - 1);
    const float lo = thr[k], hi = thr[k + 1];
    k += (a >= hi) ? 1 : 0;
    k -= (a < lo) ? 1 : 0;
    return k;
}

// points2 = (points2_OG.rowwise() + t) * R (src/icet.cpp:375-378) for one point.  xf = t[3] | R[9] row-major.  The fused
// multiply-adds are written out so that every place that transforms a point (the hot loop, the deferred literal path) gets
// the same bits whatever the compiler's contraction choices.
__device__ __forceinline__ void transform_point(float x, float y, float z, const float* __restrict__ xf, float& qx, float& qy, float& qz) {
    const float a = x + xf[0], b = y + xf[1], c = z + xf[2];
    qx = fmaf(c, xf[9], fmaf(b, xf[6], a * xf[3]));
    qy = fmaf(c, xf[10], fmaf(b, xf[7], a * xf[4]));
    qz = fmaf(c, xf[11], fmaf(b, xf[8], a * xf[5]));
}

struct PointClass { int s; bool inb; float dx, dy, dz; };

// Literal evaluation of one transformed point: cartesianToSpherical under the shared rule (correctly rounded theta / phi),
// bin, slot look-up, 6-sided bounds test (src/icet.cpp:387-388, 299).  map: voxel -> slot (int16, -1 = inactive).
__device__ __forceinline__ void classify_literal(float qx, float qy, float qz, const int16_t* map, const float* __restrict__ thr, int T, int P,
                                                 const SlotHot* __restrict__ hs, PointClass& out, bool rt2 = false) {
    float r, th, ph;
    c2s_cr(qx, qy, qz, r, th, ph);
    const float scale_t = (float)((double)T / kTwoPi), scale_p = (float)((double)P / kPi);
    int bt = bin_from_table(th, thr, T, scale_t);
    int bp = bin_from_table(ph, thr + T + 1, P, scale_p);
    if (bt >= T) bt = static_cast<int>(((double)th / kTwoPi) * (double)T) % T;
    if (bp >= P) bp = static_cast<int>(((double)ph / kPi) * (double)P) % P;
    const int s = map[T * bp + bt];
    out.s = s; out.inb = false; out.dx = out.dy = out.dz = 0.f;
    if (s >= 0) {
        const SlotHot h = hs[s];
        out.inb = inside_bounds(r, th, ph, h.az0, h.az1, h.el0, h.el1, h.inner, h.outer);
        if (rt2 && out.inb) roundtrip_any(qx, qy, qz, qx, qy, qz);          // src/icet.cpp:303: the Gaussian is fitted to the round-tripped rows
        out.dx = qx - h.mu[0]; out.dy = qy - h.mu[1]; out.dz = qz - h.mu[2];
    }
}

// float -> 64-bit fixed point: v * 2^36 rounded to the nearest integer (ties to even), two's complement.
// (double)v + 1.5 * 2^16 lies in [2^16, 2^17) for |v| < 2^15, where a double's ulp is 2^-36: the addition itself rounds v to the
// grid and leaves 2^51 + round(v * 2^36) in the mantissa field, i.e. the integer plus the bit pattern of 1.5 * 2^16 (kFixBias).
// Any fixed rounding rule would do; what matters is that EVERY conversion on the path uses this one (a slot's sums may arrive
// through LDS or straight in HBM) and that integer addition is associative.
// to_fix_biased leaves the bias in: a sum of n such words is n * kFixBias + the sum of the integers (mod 2^64), so a block that
// counts its conversions takes the bias out once per slot instead of once per value (k_gn_accumulate): two VALU instructions
// per value (v_cvt_f64_f32, v_add_f64 with a scalar constant).
// RANGE.  The trick holds for |v| < 2^15 m^2 only: beyond it the sum leaves the binade and the mantissa no longer holds the integer.
// A partial sum of <= 4 squared distances to mu1 stays far below that on an ordinary grid (a 75 x 24 voxel at 100 m is ~13 m wide),
// but a coarse grid (4 x 2 bins) with long ranges reaches it (d ~ 100 m: 4 d^2 = 40 000).  Such values take to_fix_wide: for
// |v| >= 2^15 a float is a multiple of 2^-8, so v * 2^36 is an integer and the conversion is exact -- the same round-to-nearest
// rule, trivially.  k_gn_accumulate tests a whole flush at once (kFixFastMax on the three squared sums, which bound the other six)
// in a wave-uniform branch; the rare paths (to_fix below) select per value.  Totals have 2^27 m^2 of headroom; past that the
// two's-complement sum wraps (a reference in float has lost every digit of such a voxel long before).
static_assert(kFixScale == 68719476736.0f, "kFixMagic / kFixBias below are written for a scale of 2^36");
constexpr double kFixMagic = 98304.0;                               // 1.5 * 2^(52 - 36)
constexpr unsigned long long kFixBias = 0x40F8000000000000ULL;      // its bit pattern
constexpr float kFixFastMax = 16384.0f;                             // squared sums below 2^14 keep every one of the 9 values of a flush below 2^15 (|cross| <= max square, rounding included)
__device__ __forceinline__ unsigned long long to_fix_biased(float v) {
    return (unsigned long long)__double_as_longlong((double)v + kFixMagic);
}
// any finite float (saturating far outside the accumulator's range); carries the same bias so that a block's conversion count stays right
__device__ __forceinline__ unsigned long long to_fix_wide_biased(float v) {
    return (unsigned long long)__double2ll_rn((double)v * 68719476736.0) + kFixBias;
}
__device__ __forceinline__ unsigned long long to_fix(float v) {
    return ((fabsf(v) < 32768.0f) ? to_fix_biased(v) : to_fix_wide_biased(v)) - kFixBias;
}

// One run's partial sums into a slot's HBM accumulator record (kAccWords words: [raw | in << 32], then 9 fixed-point sums).
__device__ __forceinline__ void acc_add_hbm(uint32_t* A, uint32_t nraw, uint32_t nin, float S0, float S1, float S2, float S3, float S4,
                                            float S5, float S6, float S7, float S8) {
    unsigned long long* F = reinterpret_cast<unsigned long long*>(A + 2);
    atomicAdd(reinterpret_cast<unsigned long long*>(A), (unsigned long long)nraw | ((unsigned long long)nin << 32));   // A[0] raw, A[1] in: one 64-bit add
    if (nin) {
        atomicAdd(&F[0], to_fix(S0)); atomicAdd(&F[1], to_fix(S1)); atomicAdd(&F[2], to_fix(S2)); atomicAdd(&F[3], to_fix(S3)); atomicAdd(&F[4], to_fix(S4));
        atomicAdd(&F[5], to_fix(S5)); atomicAdd(&F[6], to_fix(S6)); atomicAdd(&F[7], to_fix(S7)); atomicAdd(&F[8], to_fix(S8));
    }
}

// |q|^2 outside this range over- or underflows the stand-in coordinates (the literal formulas stay well defined): literal path.
constexpr float kR2Min = 1e-30f, kR2Max = 1e30f;

}  // namespace icet
