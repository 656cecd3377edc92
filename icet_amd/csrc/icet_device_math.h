// icet_amd/csrc/icet_device_math.h -- per-lane small dense algebra for the gfx950 kernels.
//
// The reference delegates these to Eigen (absent here): SelfAdjointEigenSolver<Matrix3f>
// (/root/reference/src/icet.cpp:181-183), SelfAdjointEigenSolver<MatrixXf> 6x6 (:455-458) and
// CompleteOrthogonalDecomposition::pseudoInverse (:320-321, :410-411, :428-429).  On the device one
// lane owns one matrix, everything lives in registers / scratch, and there are no MFMA-shaped
// contractions (3x3, 3x6, 6x6 only).
//
// The eigen-solvers follow the same published scheme Eigen 3.3 uses (closed-form 3x3 / Householder
// tridiagonalisation, then implicit symmetric QR with Wilkinson shift) because the reference's
// sigma-point test (src/icet.cpp:187-232) and its pred_stds inflation (:479) depend on eigenvector
// SIGNS; a Jacobi sweep would give the same eigenpairs with different signs.  Contraction is
// switched off inside these routines so host-side checks see the same rounding sequence.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>

namespace icetdev {

__device__ __forceinline__ void make_givens(float p, float q, float& c, float& s) {
#pragma clang fp contract(off)
    if (q == 0.f) { c = p < 0.f ? -1.f : 1.f; s = 0.f; }
    else if (p == 0.f) { c = 0.f; s = q < 0.f ? 1.f : -1.f; }
    else if (fabsf(p) > fabsf(q)) {
        float t = q / p; float u = sqrtf(1.f + t * t); if (p < 0.f) u = -u;
        c = 1.f / u; s = -t * c;
    } else {
        float t = p / q; float u = sqrtf(1.f + t * t); if (q < 0.f) u = -u;
        s = -1.f / u; c = -t * s;
    }
}

// Eigen 3.3's numext::hypot (MathFunctions.h, hypot_impl): plain IEEE operations, reproduced bit for bit by the CPU checker of the tests;
// libm's hypotf is not specified to the last bit.
__device__ __forceinline__ float eigen_hypot(float x, float y) {
#pragma clang fp contract(off)
    const float ax = fabsf(x), ay = fabsf(y);
    float p, qp;
    if (ax > ay) { p = ax; qp = ay / p; } else { p = ay; qp = ax / p; }
    if (p == 0.f) return 0.f;
    return p * sqrtf(1.f + qp * qp);
}

// One implicit-shift QR sweep on the unreduced block [start, end] of a symmetric tridiagonal matrix;
// the rotations are accumulated into Q (N x N row-major) on the right.
template <int N, bool kVec = true>
__device__ inline void tridiag_qr_step(float* diag, float* sub, int start, int end, float* Q) {
#pragma clang fp contract(off)
    float td = (diag[end - 1] - diag[end]) * 0.5f;
    float e = sub[end - 1];
    float mu = diag[end];
    if (td == 0.f) {
        mu -= fabsf(e);
    } else if (e != 0.f) {
        const float e2 = e * e;
        const float h = eigen_hypot(td, e);
        if (e2 == 0.f) mu -= e / ((td + (td > 0.f ? h : -h)) / e);
        else           mu -= e2 / (td + (td > 0.f ? h : -h));
    }
    float x = diag[start] - mu;
    float z = sub[start];
    for (int k = start; k < end && z != 0.f; ++k) {
        float c, s; make_givens(x, z, c, s);
        float sdk  = s * diag[k] + c * sub[k];
        float dkp1 = s * sub[k] + c * diag[k + 1];
        diag[k]     = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
        diag[k + 1] = s * sdk + c * dkp1;
        sub[k]      = c * sdk - s * dkp1;
        if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
        x = sub[k];
        if (k < end - 1) { z = -s * sub[k + 1]; sub[k + 1] = c * sub[k + 1]; }
        if (kVec) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                float xi = Q[i * N + k], yi = Q[i * N + k + 1];
                Q[i * N + k]     = c * xi - s * yi;
                Q[i * N + k + 1] = s * xi + c * yi;
            }
        }
    }
}

// Deflate / iterate / sort ascending (eigenvector columns follow).  Returns false on no convergence.
template <int N, bool kVec = true>
__device__ inline bool tridiag_eigen(float* diag, float* sub, float* Q) {
#pragma clang fp contract(off)
    int end = N - 1, start = 0, iter = 0;
    const float precision = 2.f * FLT_EPSILON;
    while (end > 0) {
        for (int i = start; i < end; ++i)
            if (fabsf(sub[i]) <= (fabsf(diag[i]) + fabsf(diag[i + 1])) * precision || fabsf(sub[i]) <= FLT_MIN)
                sub[i] = 0.f;
        while (end > 0 && sub[end - 1] == 0.f) end--;
        if (end <= 0) break;
        iter++;
        if (iter > 30 * N) break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0.f) start--;
        tridiag_qr_step<N, kVec>(diag, sub, start, end, Q);
    }
    bool ok = iter <= 30 * N;
    if (ok) {
        for (int i = 0; i < N - 1; ++i) {
            int k = i; float mn = diag[i];
            for (int j = i + 1; j < N; j++) if (diag[j] < mn) { mn = diag[j]; k = j; }
            if (k != i) {
                float t = diag[i]; diag[i] = diag[k]; diag[k] = t;
                if (kVec) for (int r = 0; r < N; r++) { float q = Q[r * N + i]; Q[r * N + i] = Q[r * N + k]; Q[r * N + k] = q; }
            }
        }
    }
    return ok;
}

// Symmetric 3x3 (lower triangle a00 a10 a11 a20 a21 a22): eigenvalues ascending, eigenvectors = columns
// of V (row-major).  Closed-form tridiagonalisation as for a fixed-size 3x3, then QR iteration.
__device__ inline bool eig3_sym(float a00, float a10, float a11, float a20, float a21, float a22, float ev[3], float V[9]) {
#pragma clang fp contract(off)
    float scale = fmaxf(fmaxf(fmaxf(fabsf(a00), fabsf(a10)), fmaxf(fabsf(a11), fabsf(a20))), fmaxf(fabsf(a21), fabsf(a22)));
    if (scale == 0.f) scale = 1.f;
    a00 /= scale; a10 /= scale; a11 /= scale; a20 /= scale; a21 /= scale; a22 /= scale;
    float diag[3], sub[3] = {0.f, 0.f, 0.f};
    V[0] = 1.f; V[1] = 0.f; V[2] = 0.f; V[3] = 0.f; V[4] = 1.f; V[5] = 0.f; V[6] = 0.f; V[7] = 0.f; V[8] = 1.f;
    diag[0] = a00;
    float v1norm2 = a20 * a20;
    if (v1norm2 <= FLT_MIN) {
        diag[1] = a11; diag[2] = a22; sub[0] = a10; sub[1] = a21;
    } else {
        float beta = sqrtf(a10 * a10 + v1norm2);
        float invBeta = 1.f / beta;
        float m01 = a10 * invBeta, m02 = a20 * invBeta;
        float q = 2.f * m01 * a21 + m02 * (a22 - a11);
        diag[1] = a11 + m02 * q;
        diag[2] = a22 - m02 * q;
        sub[0] = beta;
        sub[1] = a21 - m01 * q;
        V[4] = m01; V[5] = m02; V[7] = m02; V[8] = -m01;
    }
    bool ok = tridiag_eigen<3>(diag, sub, V);
    ev[0] = diag[0] * scale; ev[1] = diag[1] * scale; ev[2] = diag[2] * scale;
    return ok;
}

// SelfAdjointEigenSolver<MatrixXf> on a symmetric 6x6 (row-major, lower triangle read; src/icet.cpp:455-458): eigenvalues ascending, eigenvectors = columns of Q --
// Householder tridiagonalisation, implicit symmetric QR with Wilkinson shift, selection sort -- on REGISTERS, by a whole wave (round 6; rounds 4-5 ran it on one
// lane through LDS arrays with run-time indices: 39 us of the literal route's 67; here 16).
// Every lane evaluates the scalar recurrences -- scaling, the five Householder steps of the tridiagonalisation, the implicit-QR sweeps on (diag, sub), deflation,
// the final selection sort -- on the same values in the same order of float operations as the one-lane form (all loops unrolled, run-time positions turned into
// predicates and selects, so that nothing is indexed dynamically and nothing leaves the register file), and the one part whose work is per column / per row is
// spread over lanes: lane c < 6 accumulates column c of Q = H0 H1 ... H4, a 6 x 6 transpose through v_readlane hands lane r < 6 ROW r, and every Givens rotation of a sweep
// touches two registers per lane.  Out: ev[6] on every lane, Qrow[j] = Q[lane][j] on lanes 0..5 (other lanes: undefined).  Bit-identical to the CPU checker's
// one-thread form (tests/test_gpu_parity.py::test_gn_tail_literal_bits).  What it costs is the serial chain: ~60 rotations of ~100 dependent instructions.
__device__ __forceinline__ void swap_if(bool c, float& a, float& b) { const float t = a; a = c ? b : a; b = c ? t : b; }
__device__ __forceinline__ void swap_if(bool c, int& a, int& b) { const int t = a; a = c ? b : a; b = c ? t : b; }
__device__ __forceinline__ float sel6(const float a[6], int i) { float r = a[0]; r = i == 1 ? a[1] : r; r = i == 2 ? a[2] : r; r = i == 3 ? a[3] : r; r = i == 4 ? a[4] : r; r = i == 5 ? a[5] : r; return r; }
__device__ inline bool eig6_sym_wave(const float* Ain, float ev[6], float Qrow[6], int lane) {
#pragma clang fp contract(off)
    constexpr int n = 6;
    float A[36];
    float scale = 0.f;
#pragma unroll
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) scale = fmaxf(scale, fabsf(Ain[i * n + j]));
    }
    if (scale == 0.f) scale = 1.f;
#pragma unroll
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < n; j++) A[i * n + j] = (j <= i) ? Ain[i * n + j] / scale : 0.f;
    }
    float h[5];
#pragma unroll
    for (int i = 0; i < n - 1; i++) {
        const int rem = n - i - 1;
        // the Householder vector of column i below the diagonal: essential part in place, tau / beta
        float tau, beta;
        {
            float tailSq = 0.f;
#pragma unroll
            for (int t = 1; t < rem; t++) tailSq += A[(i + 1 + t) * n + i] * A[(i + 1 + t) * n + i];
            const float c0 = A[(i + 1) * n + i];
            const bool flat = tailSq <= FLT_MIN;
            float b2 = sqrtf(c0 * c0 + tailSq);
            if (c0 >= 0.f) b2 = -b2;
            beta = flat ? c0 : b2;
            tau = flat ? 0.f : (b2 - c0) / b2;
#pragma unroll
            for (int t = 1; t < rem; t++) A[(i + 1 + t) * n + i] = flat ? 0.f : A[(i + 1 + t) * n + i] / (c0 - b2);
        }
        A[(i + 1) * n + i] = 1.f;
        float v[5], p[5];
#pragma unroll
        for (int k = 0; k < rem; k++) v[k] = A[(i + 1 + k) * n + i];
#pragma unroll
        for (int a = 0; a < rem; a++) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < rem; b++) {
                const int ra = i + 1 + a, rb = i + 1 + b;
                const float m = (ra >= rb) ? A[ra * n + rb] : A[rb * n + ra];
                s += m * (tau * v[b]);
            }
            p[a] = s;
        }
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < rem; k++) dot += p[k] * v[k];
        const float alpha = tau * -0.5f * dot;
#pragma unroll
        for (int k = 0; k < rem; k++) p[k] += alpha * v[k];
#pragma unroll
        for (int a = 0; a < rem; a++) {
#pragma unroll
            for (int b = 0; b <= a; b++) A[(i + 1 + a) * n + i + 1 + b] -= v[a] * p[b] + p[a] * v[b];
        }
        A[(i + 1) * n + i] = beta;
        h[i] = tau;
    }
    float diag[6], sub[6];
#pragma unroll
    for (int i = 0; i < n; i++) { diag[i] = A[i * n + i]; sub[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < n - 1; i++) sub[i] = A[(i + 1) * n + i];
    // Q = H0 ... H4 applied to I, column `lane` (lanes >= 6 run along on column 5's unit vector: never read)
    float Qc[6];
    const int col = lane < 6 ? lane : 5;
#pragma unroll
    for (int r = 0; r < n; r++) Qc[r] = (r == col) ? 1.f : 0.f;
#pragma unroll
    for (int k = n - 2; k >= 0; k--) {
        const int rem = n - k - 1;
        float v[5];
        v[0] = 1.f;
#pragma unroll
        for (int t = 1; t < rem; t++) v[t] = A[(k + 1 + t) * n + k];
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < rem; t++) s += v[t] * Qc[k + 1 + t];
        s *= h[k];
#pragma unroll
        for (int t = 0; t < rem; t++) Qc[k + 1 + t] -= s * v[t];
    }
#pragma unroll
    for (int j = 0; j < n; j++) {
#pragma unroll
        for (int r = 0; r < n; r++) {
            const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Qc[r]), j));   // Q[r][j]
            if (r == 0) Qrow[j] = t; else Qrow[j] = (lane == r) ? t : Qrow[j];
        }
    }
    // tridiag_eigen<6, true>
    int end = n - 1, start = 0, iter = 0;
    const float precision = 2.f * FLT_EPSILON;
    while (end > 0) {
#pragma unroll
        for (int i = 0; i < n - 1; i++) {
            const bool small = fabsf(sub[i]) <= (fabsf(diag[i]) + fabsf(diag[i + 1])) * precision || fabsf(sub[i]) <= FLT_MIN;
            sub[i] = (i >= start && i < end && small) ? 0.f : sub[i];
        }
#pragma unroll
        for (int c = n - 1; c >= 1; c--) end = (end == c && sub[c - 1] == 0.f) ? c - 1 : end;
        if (end <= 0) break;
        iter++;
        if (iter > 30 * n) break;
        start = end - 1;
#pragma unroll
        for (int c = n - 2; c >= 1; c--) start = (start == c && sub[c - 1] != 0.f) ? c - 1 : start;
        // tridiag_qr_step<6, true>(diag, sub, start, end, Q)
        const float d_e1 = sel6(diag, end - 1), d_e = sel6(diag, end);
        const float td = (d_e1 - d_e) * 0.5f;
        const float e = sel6(sub, end - 1);
        float mu = d_e;
        if (td == 0.f) {
            mu -= fabsf(e);
        } else if (e != 0.f) {
            const float e2 = e * e;
            const float hy = eigen_hypot(td, e);
            if (e2 == 0.f) mu -= e / ((td + (td > 0.f ? hy : -hy)) / e);
            else           mu -= e2 / (td + (td > 0.f ? hy : -hy));
        }
        float x = sel6(diag, start) - mu;
        float z = sel6(sub, start);
#pragma unroll
        for (int k = 0; k < n - 1; k++) {
            if (k >= start && k < end && z != 0.f) {                 // (the same on every lane)
                float c, s; make_givens(x, z, c, s);
                const float sdk  = s * diag[k] + c * sub[k];
                const float dkp1 = s * sub[k] + c * diag[k + 1];
                diag[k]     = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
                diag[k + 1] = s * sdk + c * dkp1;
                sub[k]      = c * sdk - s * dkp1;
                if (k > 0) sub[k > 0 ? k - 1 : 0] = (k > start) ? c * sub[k > 0 ? k - 1 : 0] - s * z : sub[k > 0 ? k - 1 : 0];
                x = sub[k];
                if (k < n - 2) {
                    const bool more = k < end - 1;
                    z = more ? -s * sub[k + 1] : z;
                    sub[k + 1] = more ? c * sub[k + 1] : sub[k + 1];
                }
                const float xi = Qrow[k], yi = Qrow[k + 1];
                Qrow[k]     = c * xi - s * yi;
                Qrow[k + 1] = s * xi + c * yi;
            }
        }
    }
    const bool ok = iter <= 30 * n;
    if (ok) {
#pragma unroll
        for (int i = 0; i < n - 1; i++) {
            int k = i; float mn = diag[i];
#pragma unroll
            for (int j = i + 1; j < n; j++) { const bool lt = diag[j] < mn; mn = lt ? diag[j] : mn; k = lt ? j : k; }
#pragma unroll
            for (int j = i + 1; j < n; j++) { swap_if(k == j, diag[i], diag[j]); swap_if(k == j, Qrow[i], Qrow[j]); }
        }
    }
#pragma unroll
    for (int i = 0; i < n; i++) ev[i] = diag[i] * scale;
    return ok;
}

// Inverse of a symmetric positive-definite 6x6 (row-major) by Cholesky, fully unrolled (registers only).
// Returns false when a pivot is not positive (the caller then takes the eigen-decomposition route).
__device__ inline bool chol6_inverse(const float* A, float* inv) {
    float L[6][6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        float sd = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; k++) sd -= L[j][k] * L[j][k];
        if (!(sd > 0.f)) return false;
        const float dg = sqrtf(sd), rd = 1.f / dg;
        L[j][j] = dg;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            float t = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
            L[i][j] = t * rd;
        }
    }
    float Li[6][6];                                             // inverse of L (lower triangular)
#pragma unroll
    for (int j = 0; j < 6; j++) {
        Li[j][j] = 1.f / L[j][j];
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            float t = 0.f;
#pragma unroll
            for (int k = j; k < i; k++) t -= L[i][k] * Li[k][j];
            Li[i][j] = t / L[i][i];
        }
    }
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b <= a; b++) {
            float t = 0.f;
#pragma unroll
            for (int k = a; k < 6; k++) t += Li[k][a] * Li[k][b];       // (L^-T L^-1)(a,b), k >= max(a,b) = a
            inv[a * 6 + b] = t; inv[b * 6 + a] = t;
        }
    return true;
}

// Moore-Penrose pseudo-inverse of a symmetric PSD 3x3 (xx,xy,xz,yy,yz,zz) by cyclic Jacobi.
// Eigenvalues <= rel_tol * lambda_max are treated as rank deficiency (the reference's COD uses
// eps * 3 relative to its largest pivot).  Rows/columns that are exactly zero (axes masked by L)
// stay exactly zero, so the result is the inverse of the kept principal block embedded in zeros.
// Run in DOUBLE on the float matrix: the projected noise matrix of a line-like voxel has a condition number of 1e4..1e5, and a
// float eigen-solve loses cond * eps = 0.2 .. 0.6 % of the weight such a voxel gets (it can carry 20 % of H^T W H).
template <typename R>
__device__ inline void pinv3_sym(const float a[6], float rel_tol, float w[6]) {
    R A00 = a[0], A01 = a[1], A02 = a[2], A11 = a[3], A12 = a[4], A22 = a[5];
    R V00 = 1, V01 = 0, V02 = 0, V10 = 0, V11 = 1, V12 = 0, V20 = 0, V21 = 0, V22 = 1;
    const R tiny = sizeof(R) == 8 ? (R)1e-17 : (R)1e-9;
#define ICET_JROT(App, Aqq, Apq, Apr, Aqr, Vp0, Vq0, Vp1, Vq1, Vp2, Vq2)                                      \
    if (Apq != (R)0) {                                                                                          \
        R theta = (Aqq - App) / ((R)2 * Apq);                                                                   \
        R t = (theta >= (R)0 ? (R)1 : (R)-1) / (fabs(theta) + sqrt(theta * theta + (R)1));                      \
        R c = (R)1 / sqrt(t * t + (R)1), s = t * c;                                                             \
        R app = App - t * Apq, aqq = Aqq + t * Apq;                                                             \
        R apr = c * Apr - s * Aqr, aqr = s * Apr + c * Aqr;                                                     \
        App = app; Aqq = aqq; Apq = (R)0; Apr = apr; Aqr = aqr;                                                 \
        R v;                                                                                                    \
        v = c * Vp0 - s * Vq0; Vq0 = s * Vp0 + c * Vq0; Vp0 = v;                                                \
        v = c * Vp1 - s * Vq1; Vq1 = s * Vp1 + c * Vq1; Vp1 = v;                                                \
        v = c * Vp2 - s * Vq2; Vq2 = s * Vp2 + c * Vq2; Vp2 = v;                                                \
    }
    // V holds eigenvectors as columns: Vrk = component r of eigenvector k.
    for (int sweep = 0; sweep < (sizeof(R) == 8 ? 10 : 6); sweep++) {
        R off = fabs(A01) + fabs(A02) + fabs(A12);
        R dg = fabs(A00) + fabs(A11) + fabs(A22);
        if (off <= tiny * dg) break;
        ICET_JROT(A00, A11, A01, A02, A12, V00, V01, V10, V11, V20, V21)
        ICET_JROT(A00, A22, A02, A01, A12, V00, V02, V10, V12, V20, V22)
        ICET_JROT(A11, A22, A12, A01, A02, V01, V02, V11, V12, V21, V22)
    }
#undef ICET_JROT
    R lmax = fmax(fmax(fabs(A00), fabs(A11)), fabs(A22));
    R thr = (R)rel_tol * lmax;
    R i0 = (fabs(A00) > thr) ? (R)1 / A00 : (R)0;
    R i1 = (fabs(A11) > thr) ? (R)1 / A11 : (R)0;
    R i2 = (fabs(A22) > thr) ? (R)1 / A22 : (R)0;
    w[0] = (float)(i0 * V00 * V00 + i1 * V01 * V01 + i2 * V02 * V02);
    w[1] = (float)(i0 * V00 * V10 + i1 * V01 * V11 + i2 * V02 * V12);
    w[2] = (float)(i0 * V00 * V20 + i1 * V01 * V21 + i2 * V02 * V22);
    w[3] = (float)(i0 * V10 * V10 + i1 * V11 * V11 + i2 * V12 * V12);
    w[4] = (float)(i0 * V10 * V20 + i1 * V11 * V21 + i2 * V12 * V22);
    w[5] = (float)(i0 * V20 * V20 + i1 * V21 * V21 + i2 * V22 * V22);
}

// The same pseudo-inverse without the eigen-decomposition where that is provably equal: a projected noise matrix M Rn M^T is positive
// semi-definite, exactly zero in the rows / columns of the axes L masks out, and otherwise well conditioned on ordinary voxels.  With
// the masked axes filled by the largest diagonal entry (an eigenvalue between the block's smallest and largest, so the condition number is
// the block's), cond_2 <= |A|_F |A^-1|_F <= 1e6 proves that no eigenvalue of the block falls below rel_tol (3 eps = 3.6e-7) of the largest:
// the pseudo-inverse is then the inverse of the block (adjugate / determinant in double: at a condition number of 1e5 that keeps 11 digits)
// and zero on the masked axes.  Anything else -- ill-conditioned, indefinite by rounding, NaN -- takes the Jacobi route above.
// Why it matters: a double-precision Jacobi rotation is a chain of five dependent divisions / square roots (~1 us per sweep per lane), and
// k_gn_solve is one block of dependent latencies per pair.
__device__ inline void pinv3_sym_fast(const float a[6], float rel_tol, float w[6]) {
    double a00 = a[0], a01 = a[1], a02 = a[2], a11 = a[3], a12 = a[4], a22 = a[5];
    const bool z0 = a[0] == 0.f, z1 = a[3] == 0.f, z2 = a[5] == 0.f;
    const double fill = fmax(fmax(a00, a11), a22);
    if (z0) a00 = fill;
    if (z1) a11 = fill;
    if (z2) a22 = fill;
    const bool clean = (!z0 || (a[1] == 0.f && a[2] == 0.f)) && (!z1 || (a[1] == 0.f && a[4] == 0.f)) && (!z2 || (a[2] == 0.f && a[4] == 0.f));
    const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    const double det = a00 * c00 + a01 * c01 + a02 * c02;
    if (clean && fill > 0.0 && det > 0.0) {
        const double rd = 1.0 / det;
        const double i00 = c00 * rd, i01 = c01 * rd, i02 = c02 * rd, i11 = c11 * rd, i12 = c12 * rd, i22 = c22 * rd;
        const double fa = a00 * a00 + a11 * a11 + a22 * a22 + 2.0 * (a01 * a01 + a02 * a02 + a12 * a12);
        const double fi = i00 * i00 + i11 * i11 + i22 * i22 + 2.0 * (i01 * i01 + i02 * i02 + i12 * i12);
        if (fa * fi <= 1e12) {
            w[0] = z0 ? 0.f : (float)i00; w[1] = (z0 | z1) ? 0.f : (float)i01; w[2] = (z0 | z2) ? 0.f : (float)i02;
            w[3] = z1 ? 0.f : (float)i11; w[4] = (z1 | z2) ? 0.f : (float)i12; w[5] = z2 ? 0.f : (float)i22;
            return;
        }
    }
    pinv3_sym<double>(a, rel_tol, w);
}

// CompleteOrthogonalDecomposition<MatrixXf>(A).pseudoInverse() of ONE 3 x 3 per lane (row-major in, row-major out): the reference's arithmetic for the per-voxel
// weight W = pinv(L U^T R_noise U L^T) (src/icet.cpp:320-321).  Eigen 3.3's algorithm as the CPU restatement states it (column-pivoted Householder QR with norm
// down-dating, rank = pivots above 3 eps x the largest among the non-zero pivots, minimum-norm completion of a rank-deficient R in double), plain IEEE float, no
// contraction, every operation in the restatement's order -- bit-identical to it on the same matrix (tests/test_gpu_parity.py::test_pinv3_reference_bits).
// Written on REGISTERS: every index is a compile-time constant (a column exchange is a pair of selects per entry, a loop up to the run-time rank is unrolled
// under guards), because k_gn_solve has no scratch memory to spare -- a first version with local arrays cost each solve launch 35 us.
struct Col3 { float x, y, z; };
__device__ __forceinline__ void swap_if(bool c, Col3& a, Col3& b) { swap_if(c, a.x, b.x); swap_if(c, a.y, b.y); swap_if(c, a.z, b.z); }
__device__ inline void cod_pinv3_lane(const float Ain[9], float pinv[9]) {
#pragma clang fp contract(off)
    // columns of the working matrix (row i of column k = qr(i, k))
    Col3 c0{Ain[0], Ain[3], Ain[6]}, c1{Ain[1], Ain[4], Ain[7]}, c2{Ain[2], Ain[5], Ain[8]};
    auto norm3 = [](const Col3& c) { float s = 0.f; s += c.x * c.x; s += c.y * c.y; s += c.z * c.z; return sqrtf(s); };
    float nd0 = norm3(c0), nd1 = norm3(c1), nd2 = norm3(c2), nu0 = nd0, nu1 = nd1, nu2 = nd2;
    int p0 = 0, p1 = 1, p2 = 2;
    float maxn = 0.f; maxn = fmaxf(maxn, nu0); maxn = fmaxf(maxn, nu1); maxn = fmaxf(maxn, nu2);
    const float th = maxn * FLT_EPSILON; const float threshold_helper = (th * th) / 3.f;
    const float ndt = 3.4526698300124393e-04f;                                               // sqrt(FLT_EPSILON), correctly rounded
    int nonzero_pivots = 3; float maxpivot = 0.f;
    auto downdate = [&](float rkj, float& nu, float& nd, float tail_sq_below) {             // norm down-dating of one column behind step k
        if (nu != 0.f) {
            float temp = fabsf(rkj) / nu;
            temp = (1.f + temp) * (1.f - temp);
            temp = temp < 0.f ? 0.f : temp;
            const float ratio = nu / nd;
            const float temp2 = temp * ratio * ratio;
            if (temp2 <= ndt) { nd = sqrtf(tail_sq_below); nu = nd; } else nu *= sqrtf(temp);
        }
    };
    // ---- k = 0 ----
    {
        int big = 0; float bn = nu0;
        if (nu1 > bn) { bn = nu1; big = 1; }
        if (nu2 > bn) { bn = nu2; big = 2; }
        if (bn * bn < threshold_helper * 3.f) nonzero_pivots = 0;
        swap_if(big == 1, c0, c1); swap_if(big == 1, nu0, nu1); swap_if(big == 1, nd0, nd1); swap_if(big == 1, p0, p1);
        swap_if(big == 2, c0, c2); swap_if(big == 2, nu0, nu2); swap_if(big == 2, nd0, nd2); swap_if(big == 2, p0, p2);
    }
    float tau0, tau1, tau2;
    {
        float tailSq = 0.f; tailSq += c0.y * c0.y; tailSq += c0.z * c0.z;
        const float h = c0.x; float beta;
        if (tailSq <= FLT_MIN) { tau0 = 0.f; beta = h; c0.y = 0.f; c0.z = 0.f; }
        else { beta = sqrtf(h * h + tailSq); if (h >= 0.f) beta = -beta; c0.y = c0.y / (h - beta); c0.z = c0.z / (h - beta); tau0 = (beta - h) / beta; }
        c0.x = beta;
        if (fabsf(beta) > maxpivot) maxpivot = fabsf(beta);
        auto apply = [&](Col3& c) { float sd = c.x; sd += c0.y * c.y; sd += c0.z * c.z; sd *= tau0; c.x -= sd; c.y -= sd * c0.y; c.z -= sd * c0.z; };
        apply(c1); apply(c2);
        { float t = 0.f; t += c1.y * c1.y; t += c1.z * c1.z; downdate(c1.x, nu1, nd1, t); }
        { float t = 0.f; t += c2.y * c2.y; t += c2.z * c2.z; downdate(c2.x, nu2, nd2, t); }
    }
    // ---- k = 1 ----
    {
        int big = 1; float bn = nu1;
        if (nu2 > bn) { bn = nu2; big = 2; }
        if (nonzero_pivots == 3 && bn * bn < threshold_helper * 2.f) nonzero_pivots = 1;
        swap_if(big == 2, c1, c2); swap_if(big == 2, nu1, nu2); swap_if(big == 2, nd1, nd2); swap_if(big == 2, p1, p2);
        float tailSq = 0.f; tailSq += c1.z * c1.z;
        const float h = c1.y; float beta;
        if (tailSq <= FLT_MIN) { tau1 = 0.f; beta = h; c1.z = 0.f; }
        else { beta = sqrtf(h * h + tailSq); if (h >= 0.f) beta = -beta; c1.z = c1.z / (h - beta); tau1 = (beta - h) / beta; }
        c1.y = beta;
        if (fabsf(beta) > maxpivot) maxpivot = fabsf(beta);
        { float sd = c2.y; sd += c1.z * c2.z; sd *= tau1; c2.y -= sd; c2.z -= sd * c1.z; }
        { float t = 0.f; t += c2.z * c2.z; downdate(c2.y, nu2, nd2, t); }
    }
    // ---- k = 2 ----
    {
        if (nonzero_pivots == 3 && nu2 * nu2 < threshold_helper * 1.f) nonzero_pivots = 2;
        const float h = c2.z;                                        // (no tail: tailSq = 0 <= FLT_MIN)
        tau2 = 0.f;
        if (fabsf(h) > maxpivot) maxpivot = fabsf(h);
    }
    const float premult = fabsf(maxpivot) * (FLT_EPSILON * 3.f);
    int rank = 0;
    rank += (0 < nonzero_pivots && fabsf(c0.x) > premult) ? 1 : 0;
    rank += (1 < nonzero_pivots && fabsf(c1.y) > premult) ? 1 : 0;
    rank += (2 < nonzero_pivots && fabsf(c2.z) > premult) ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 9; k++) pinv[k] = 0.f;
    if (rank == 0) return;
    // R (upper triangle) and the reflectors' essential parts
    const float r00 = c0.x, r01 = c1.x, r02 = c2.x, r11 = c1.y, r12 = c2.y, r22 = c2.z;
    const float v10 = c0.y, v20 = c0.z, v21 = c1.z;
    // C = Q^T restricted to the first `rank` reflectors, applied to I: rows C0, C1, C2 (each a row vector over j)
    float C0[3] = {1.f, 0.f, 0.f}, C1[3] = {0.f, 1.f, 0.f}, C2[3] = {0.f, 0.f, 1.f};
#pragma unroll
    for (int j = 0; j < 3; j++) { float sd = C0[j]; sd += v10 * C1[j]; sd += v20 * C2[j]; sd *= tau0; C0[j] -= sd; C1[j] -= sd * v10; C2[j] -= sd * v20; }
    if (rank > 1) {
#pragma unroll
        for (int j = 0; j < 3; j++) { float sd = C1[j]; sd += v21 * C2[j]; sd *= tau1; C1[j] -= sd; C2[j] -= sd * v21; }
    }
    if (rank > 2) {
#pragma unroll
        for (int j = 0; j < 3; j++) { float sd = C2[j]; sd *= tau2; C2[j] -= sd; }
    }
    float Y0[3], Y1[3], Y2[3];
    if (rank == 3) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            Y2[j] = C2[j] / r22;
            { float sv = C1[j]; sv -= r12 * Y2[j]; Y1[j] = sv / r11; }
            { float sv = C0[j]; sv -= r01 * Y1[j]; sv -= r02 * Y2[j]; Y0[j] = sv / r00; }
        }
    } else {
        // X = [R11 R12] (rank x 3); X^+ = X^T (X X^T)^-1; Y = X^+ C(0:rank, :), in double
        const double x00 = r00, x01 = r01, x02 = r02, x11 = r11, x12 = r12;
        double gi00, gi01 = 0.0, gi10 = 0.0, gi11 = 0.0;
        if (rank == 1) {
            double g = 0; g += x00 * x00; g += x01 * x01; g += x02 * x02;
            gi00 = 1.0 / g;
        } else {
            double g00 = 0, g01 = 0, g10 = 0, g11 = 0;
            g00 += x00 * x00; g00 += x01 * x01; g00 += x02 * x02;
            g01 += x00 * 0.0; g01 += x01 * x11; g01 += x02 * x12;
            g10 += 0.0 * x00; g10 += x11 * x01; g10 += x12 * x02;
            g11 += 0.0 * 0.0; g11 += x11 * x11; g11 += x12 * x12;
            double a00 = g00, a01 = g01, a10 = g10, a11 = g11, b00 = 1.0, b01 = 0.0, b10 = 0.0, b11 = 1.0;
            // Gauss-Jordan with partial pivoting, as the restatement does it
            if (fabs(a10) > fabs(a00)) { double t; t = a00; a00 = a10; a10 = t; t = a01; a01 = a11; a11 = t; t = b00; b00 = b10; b10 = t; t = b01; b01 = b11; b11 = t; }
            { const double dd = a00; a00 /= dd; a01 /= dd; b00 /= dd; b01 /= dd; }
            { const double f = a10; a10 -= f * a00; a11 -= f * a01; b10 -= f * b00; b11 -= f * b01; }
            { const double dd = a11; a10 /= dd; a11 /= dd; b10 /= dd; b11 /= dd; }
            { const double f = a01; a00 -= f * a10; a01 -= f * a11; b00 -= f * b10; b01 -= f * b11; }
            gi00 = b00; gi01 = b01; gi10 = b10; gi11 = b11;
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            // w_i = sum_m Ginv(i, m) C(m, j);  Y(t, j) = sum_i X(i, t) w_i
            double w0 = 0, w1 = 0;
            w0 += gi00 * (double)C0[j]; if (rank > 1) w0 += gi01 * (double)C1[j];
            if (rank > 1) { w1 += gi10 * (double)C0[j]; w1 += gi11 * (double)C1[j]; }
            double y0 = 0, y1 = 0, y2 = 0;
            y0 += x00 * w0; if (rank > 1) y0 += 0.0 * w1;
            y1 += x01 * w0; if (rank > 1) y1 += x11 * w1;
            y2 += x02 * w0; if (rank > 1) y2 += x12 * w1;
            Y0[j] = (float)y0; Y1[j] = (float)y1; Y2[j] = (float)y2;
        }
    }
    // pinv(perm[k], j) = Y(k, j)
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) pinv[r * 3 + j] = (p0 == r) ? Y0[j] : ((p1 == r) ? Y1[j] : Y2[j]);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The LITERAL 6x6 tail of one Gauss-Newton iteration (/root/reference/src/icet.cpp:410-430), for matrices the Cholesky route of
// k_gn_solve cannot prove well conditioned: noise_mat = CompleteOrthogonalDecomposition(HTWH).pseudoInverse() (:410-411), pred_stds
// (:412-417), checkCondition (:443-492: eigenvectors of HTWH, `pred_stds += U2.col(k)` and one row of L2 dropped per pruned axis),
// dx = pinv(L2 lam U2^T) L2 U2^T HTWdz (:427-430) -- statement by statement, with Eigen 3.3's algorithms (column-pivoted Householder QR with
// norm down-dating; rank = pivots above eps * min(rows, cols) * the largest pivot; minimum-norm completion of a rank-deficient R) in plain
// IEEE float operations without contraction, so that on the same (HTWH, HTWdz) bits the result -- the rank decision, the number of pruned
// axes, the SIGN of the eigenvector added to pred_stds -- does not depend on which side evaluated it (checked bit for bit against the CPU
// checker over condition numbers 5e4 .. 3e7, tests/test_gpu_parity.py::test_gn_tail_literal_bits).  Rounds 2-4 used an eigenvalue rule
// (|lambda_k| > 6 eps lambda_max) for the rank here; scripts/rank_rule_study.py finds it disagreeing with the pivot rule on 35 of 8400
// matrices around cond = 1.4e6 = 1 / (6 eps), right above checkCondition's cutoff.
// Rare by construction (a tunnel, a single wall, open ground).  Rounds 4-5 wrote it for fidelity with generic small matrices in an LDS workspace (67 us per
// evaluation); round 6 moved the two decompositions into registers (eig6_sym_wave, cod_pinv_c6_wave: 32 us, the same bits) and the first pseudo-inverse onto a second
// wave beside the eigen-decomposition (gn_tail_literal<true> / gn_tail_literal_helper: 26 us).

// What the phases of the literal tail hand to each other lives in ONE workspace the caller places in LDS: input, results, and the operands of the small products
// (no scratch memory in the kernels that hold this rarely taken branch).
struct GnTailWs {
    float H[36], g[6];                                            // in
    float cov[36], ps[6], dx[6], ev[6]; int pruned, rank;         // out
    float U2[36], L2[36], lam[36], U2t[36], T1[36], innards[36], inv[36], T2[36], lhs[36];
    float H1[36]; int rank1;                                       // the helper wave's copy of H and the rank it found (gn_tail_literal<true>)
#ifdef ICET_TAIL_TIMING
    unsigned long long ts[8];                                     // diagnostic build: wall_clock64 at the phase boundaries of gn_tail_literal
#endif
};

// The wave works TOGETHER (round 5b; one lane alone took 157 us per evaluation, half of it in the six small matrix products): whatever is
// independent per column or per element -- a product's 36 entries, a Householder reflector applied to the columns right of the pivot, the back
// substitution of each right-hand side -- goes to one lane per column / element, each with ITS sequential order of float operations unchanged, so the bits
// are the single lane's; scalar decisions (pivot search, the reflector's norm, rank) are evaluated by every lane on the same values.  Between phases that
// meet in LDS a wave-level fence: LDS operations of one wave complete in program order, the fence keeps the compiler from moving or caching them.
#define ICET_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#ifdef ICET_TAIL_TIMING
#define ICET_TS(k) do { if ((threadIdx.x & 63u) == 0) w.ts[k] = wall_clock64(); } while (0)
#else
#define ICET_TS(k) do { } while (0)
#endif

// C (ar x bc) = A (ar x ac) * B (ac x bc), row-major, each entry a sequential float sum starting from 0 in index order (zero terms included); one lane per entry
__device__ inline void mm_seq(const float* A, int ar, int ac, const float* B, int bc, float* C, int lane) {
#pragma clang fp contract(off)
    if (lane < ar * bc) {
        const int i = lane / bc, j = lane - i * bc;
        float s = 0.f;
        for (int k = 0; k < ac; k++) s += A[i * ac + k] * B[k * bc + j];
        C[i * bc + j] = s;
    }
    ICET_WSYNC();
}

// CompleteOrthogonalDecomposition<MatrixXf>(A).pseudoInverse() for a rows x 6 matrix A (rows <= 6, row-major in LDS: the two calls of the literal route; out is 6 x rows)
// on REGISTERS (round 6; rounds 4-5: the same statements over LDS words, a fence between phases, 12.5 us per call; here 6.5): lane j < 6 owns COLUMN j of the working matrix qr (and of C = Q^T I and of the solution Y), its two column norms and its
// entry of the permutation; what the one-matrix algorithm reads from another column (the pivot column's reflector, R's entries in the back substitution) travels by
// v_readlane; scalar decisions (pivot search, the reflector's norm, the rank) are evaluated by every lane on the same broadcast values.  The rank-deficient branch keeps its
// one-lane-per-entry Gauss-Jordan in double on a fixed 6 x 6 lane grid (rows and pivot row exchanged by ds_bpermute).  All loops unrolled, run-time sizes as predicates: no
// local array is indexed dynamically, nothing goes through LDS but the input and the result.  Returns the rank (wave-uniform); bit-identical to the CPU checker's cod_pinv.
__device__ __forceinline__ float rl(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
__device__ __forceinline__ double rld(double x, int l) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ inline int cod_pinv_c6_wave(const float* Ain, int rows_in, float* pinv, int lane) {
#pragma clang fp contract(off)
    constexpr int cols = 6;
    const int rows = __builtin_amdgcn_readfirstlane(rows_in);
    const int size = rows;                                        // rows <= cols
    const int cj = lane < cols ? lane : cols - 1;                 // (lanes >= 6 run along on column 5: never read, never written out)
    float q[6];
#pragma unroll
    for (int i = 0; i < 6; i++) q[i] = (i < rows) ? Ain[(i < rows ? i : 0) * cols + cj] : 0.f;
    float nDir, nUpd; int perm = cj;
    {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) s = (i < rows) ? s + q[i] * q[i] : s;
        nDir = sqrtf(s); nUpd = nDir;
    }
    float maxn = 0.f;
#pragma unroll
    for (int k = 0; k < cols; k++) maxn = fmaxf(maxn, rl(nUpd, k));
    const float th = maxn * FLT_EPSILON; const float threshold_helper = (th * th) / float(rows);
    const float norm_downdate_threshold = 3.4526698300124393e-04f;
    int nonzero_pivots = size; float maxpivot = 0.f;
    float hc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (k < size) {
            int big = k; float bn = rl(nUpd, k);
#pragma unroll
            for (int j = k + 1; j < cols; j++) { const float nj = rl(nUpd, j); const bool gt = nj > bn; bn = gt ? nj : bn; big = gt ? j : big; }
            big = __builtin_amdgcn_readfirstlane(big);
            const float big_sq = bn * bn;
            if (nonzero_pivots == size && big_sq < threshold_helper * float(rows - k)) nonzero_pivots = k;
            if (k != big) {
#pragma unroll
                for (int i = 0; i < 6; i++) { const float a = rl(q[i], k), b = rl(q[i], big); q[i] = (lane == k) ? b : (lane == big) ? a : q[i]; }
                { const float a = rl(nUpd, k), b = rl(nUpd, big); nUpd = (lane == k) ? b : (lane == big) ? a : nUpd; }
                { const float a = rl(nDir, k), b = rl(nDir, big); nDir = (lane == k) ? b : (lane == big) ? a : nDir; }
                { const int a = __builtin_amdgcn_readlane(perm, k), b = __builtin_amdgcn_readlane(perm, big); perm = (lane == k) ? b : (lane == big) ? a : perm; }
            }
            // make_householder on column k, rows k .. rows - 1
            float x[6];
#pragma unroll
            for (int i = k; i < 6; i++) x[i] = rl(q[i], k);
            float tailSq = 0.f;
#pragma unroll
            for (int i = k + 1; i < 6; i++) tailSq = (i < rows) ? tailSq + x[i] * x[i] : tailSq;
            const float c0 = x[k];
            float tau, beta;
            const bool flat = tailSq <= FLT_MIN;
            if (flat) { tau = 0.f; beta = c0; }
            else { beta = sqrtf(c0 * c0 + tailSq); if (c0 >= 0.f) beta = -beta; tau = (beta - c0) / beta; }
            float e[6];
#pragma unroll
            for (int i = k + 1; i < 6; i++) e[i] = (i < rows) ? (flat ? 0.f : x[i] / (c0 - beta)) : 0.f;
            if (lane == k) {
                q[k] = beta;
#pragma unroll
                for (int i = k + 1; i < 6; i++) q[i] = (i < rows) ? e[i] : q[i];
            }
            hc[k] = tau;
            if (fabsf(beta) > maxpivot) maxpivot = fabsf(beta);
            if (lane > k && lane < cols) {                        // H = I - tau v v^T on the columns right of the pivot, a lane per column
                float sdot = q[k];
#pragma unroll
                for (int i = k + 1; i < 6; i++) sdot = (i < rows) ? sdot + e[i] * q[i] : sdot;
                sdot *= tau;
                q[k] -= sdot;
#pragma unroll
                for (int i = k + 1; i < 6; i++) q[i] = (i < rows) ? q[i] - sdot * e[i] : q[i];
                if (nUpd != 0.f) {                                // norm down-dating of the same column
                    float temp = fabsf(q[k]) / nUpd;
                    temp = (1.f + temp) * (1.f - temp);
                    temp = temp < 0.f ? 0.f : temp;
                    const float ratio = nUpd / nDir;
                    const float temp2 = temp * ratio * ratio;
                    if (temp2 <= norm_downdate_threshold) {
                        float s2 = 0.f;
#pragma unroll
                        for (int i = k + 1; i < 6; i++) s2 = (i < rows) ? s2 + q[i] * q[i] : s2;
                        nDir = sqrtf(s2); nUpd = nDir;
                    } else {
                        nUpd *= sqrtf(temp);
                    }
                }
            }
        }
    }
    const float premult = fabsf(maxpivot) * (FLT_EPSILON * float(size));
    int rank = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) rank += (i < nonzero_pivots && fabsf(rl(q[i], i)) > premult) ? 1 : 0;
    rank = __builtin_amdgcn_readfirstlane(rank);
    if (lane < cols * rows) pinv[lane] = 0.f;
    if (rank == 0) return 0;
    float Cc[6];                                                  // Q^T restricted to the first `rank` reflectors, applied to I (rows x rows): column `lane`
#pragma unroll
    for (int i = 0; i < 6; i++) Cc[i] = (i == cj) ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (k < rank) {
            float v[6];
#pragma unroll
            for (int i = k + 1; i < 6; i++) v[i] = rl(q[i], k);
            float sd = Cc[k];
#pragma unroll
            for (int i = k + 1; i < 6; i++) sd = (i < rows) ? sd + v[i] * Cc[i] : sd;
            sd *= hc[k];
            Cc[k] -= sd;
#pragma unroll
            for (int i = k + 1; i < 6; i++) Cc[i] = (i < rows) ? Cc[i] - sd * v[i] : Cc[i];
        }
    }
    float R[6][6];                                                // R[i][t], t >= i: the upper triangle of qr, on every lane
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int t = i; t < 6; t++) R[i][t] = rl(q[i], t);
    }
    float Yc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                 // permuted solution (cols x rows), column `lane`
    if (rank == cols) {
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            float sd = Cc[i];
#pragma unroll
            for (int t = i + 1; t < 6; t++) sd -= R[i][t] * Yc[t];
            Yc[i] = sd / R[i][i];
        }
    } else {
        // X = [R11 R12] (rank x cols); X^+ = X^T (X X^T)^-1 in double; Y = X^+ C(0:rank, :).  G, Ginv: entry (ii, jj) on lane ii * 6 + jj.
        const int ii = lane / 6, jj = lane - 6 * ii;
        const bool cell = lane < 36 && ii < rank && jj < rank;
        double G = 0.0, Ginv = 0.0;
        {
            double sd = 0;
#pragma unroll
            for (int t = 0; t < cols; t++) {
                float xi = 0.f, xj = 0.f;
#pragma unroll
                for (int i = 0; i <= t; i++) { xi = (ii == i) ? R[i][t] : xi; xj = (jj == i) ? R[i][t] : xj; }
                sd += (double)xi * (double)xj;
            }
            G = sd; Ginv = (ii == jj) ? 1.0 : 0.0;
        }
#pragma unroll
        for (int p = 0; p < 6; p++) {
            if (p < rank) {
                int piv = p; double best = fabs(rld(G, p * 6 + p));
#pragma unroll
                for (int i = p + 1; i < 6; i++) { const double gi = fabs(rld(G, i * 6 + p)); const bool gt = i < rank && gi > best; best = gt ? gi : best; piv = gt ? i : piv; }
                piv = __builtin_amdgcn_readfirstlane(piv);
                if (piv != p) {
                    const int src = (ii == p) ? piv * 6 + jj : (ii == piv) ? p * 6 + jj : lane;
                    G = __shfl(G, src & 63); Ginv = __shfl(Ginv, src & 63);
                }
                const double d = rld(G, p * 7);
                if (cell && ii == p) { G /= d; Ginv /= d; }
                const int rowp = (lane < 36) ? p * 6 + jj : lane, colp = (lane < 36) ? ii * 6 + p : lane;
                const double f = __shfl(G, colp), gp = __shfl(G, rowp), hp = __shfl(Ginv, rowp);
                if (cell && ii != p) { G -= f * gp; Ginv -= f * hp; }
            }
        }
        double sdt[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (i < rank) {
                double ww = 0;
#pragma unroll
                for (int mm = 0; mm < 6; mm++) { const double gim = rld(Ginv, i * 6 + mm); ww = (mm < rank) ? ww + gim * (double)Cc[mm] : ww; }
#pragma unroll
                for (int t = 0; t < 6; t++) { const double xit = (t >= i) ? (double)R[i][t < i ? i : t] : 0.0; sdt[t] += xit * ww; }
            }
        }
#pragma unroll
        for (int t = 0; t < 6; t++) Yc[t] = (float)sdt[t];
    }
#pragma unroll
    for (int k = 0; k < cols; k++) { const int pk = __builtin_amdgcn_readlane(perm, k); if (lane < rows) pinv[pk * rows + lane] = Yc[k]; }
    return rank;
}

// src/icet.cpp:410-430 for the (H, g) in the workspace: cov = noise_mat, ps = pred_stds (after checkCondition's additions), dx, ev = eigenvalues
// ascending, pruned = number of pruned axes, rank = the COD rank of HTWH.  Call from a WHOLE wave (wave-uniform control flow).
// kHelper: a SECOND wave of the block takes pinv(HTWH) (gn_tail_literal_helper) while this one runs the eigen-decomposition -- the two do not depend on each other --
// and raises *pinv_done when cov and the rank are in the workspace (round 6: 6.5 us of the route's 32 per evaluation).
template <bool kHelper = false>
__device__ __noinline__ void gn_tail_literal(GnTailWs& w, volatile int* pinv_done = nullptr) {
#pragma clang fp contract(off)
    const int lane = (int)(threadIdx.x & 63u);
    ICET_TS(0);
    int rank = 0;
    if (!kHelper) { rank = cod_pinv_c6_wave(w.H, 6, w.cov, lane); ICET_WSYNC(); }
    ICET_TS(1);
    float* U2 = w.U2;
    float ev[6], Qrow[6];
    {
        float Hr[36];
#pragma unroll
        for (int k = 0; k < 36; k++) Hr[k] = w.H[k];
        eig6_sym_wave(Hr, ev, Qrow, lane);                            // every lane: the eigenvalues; lane r < 6: row r of the eigenvectors
    }
    ICET_TS(2);
    if (kHelper) {                                                // cov (and the rank) come from the helper wave
        while (*pinv_done == 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        rank = w.rank1;
    }
    int k0 = 0;
    {
        // checkCondition (src/icet.cpp:443-492), one lane per entry of pred_stds: the pruning loop runs on the eigenvalues every lane holds
        float ps = (lane < 6) ? sqrtf(fabsf(w.cov[(lane < 6 ? lane : 0) * 7])) : 0.f;
        float condition = ev[5] / ev[0];
        int eyecount = 1;
        while (fabsf(condition) > 1e6f && eyecount < 6) {
            ps += sel6(Qrow, eyecount - 1);                           // src/icet.cpp:479
            k0++;
            condition = ev[5] / sel6(ev, eyecount);
            eyecount++;
        }
        if (lane < 6) {
            w.ps[lane] = ps; w.ev[lane] = sel6(ev, lane);
#pragma unroll
            for (int j = 0; j < 6; j++) U2[lane * 6 + j] = Qrow[j];
        }
        if (lane == 0) { w.rank = rank; w.pruned = k0; }
    }
    ICET_WSYNC();
    const int m = 6 - k0;
    if (lane < 36) {
        const int i = lane / 6, j = lane - 6 * i;
        w.L2[lane] = (i < m && j == k0 + i) ? 1.f : 0.f;
        w.lam[lane] = (i == j) ? w.ev[i] : 0.f;
        w.U2t[j * 6 + i] = U2[i * 6 + j];
    }
    ICET_WSYNC();
    ICET_TS(3);
    mm_seq(w.L2, m, 6, w.lam, 6, w.T1, lane);
    mm_seq(w.T1, m, 6, w.U2t, 6, w.innards, lane);                                           // L2 * lam * U2^T   (m x 6)   src/icet.cpp:427
    ICET_TS(4);
    cod_pinv_c6_wave(w.innards, m, w.inv, lane);                                             // 6 x m
    ICET_WSYNC();
    ICET_TS(5);
    mm_seq(w.inv, 6, m, w.L2, 6, w.T2, lane);
    mm_seq(w.T2, 6, 6, w.U2t, 6, w.lhs, lane);
    mm_seq(w.lhs, 6, 6, w.g, 1, w.dx, lane);                                                 // src/icet.cpp:430
    ICET_TS(6);
}
// The helper wave's share (every lane holds the same Hm): noise_mat = pinv(HTWH) into the workspace, then the flag.
__device__ __noinline__ void gn_tail_literal_helper(const float* Hm, GnTailWs& w, volatile int* pinv_done) {
#pragma clang fp contract(off)
    const int lane = (int)(threadIdx.x & 63u);
    if (lane < 36) { float v = Hm[0];
#pragma unroll
        for (int k = 1; k < 36; k++) v = (lane == k) ? Hm[k] : v;
        w.H1[lane] = v; }
    ICET_WSYNC();
    const int rank = cod_pinv_c6_wave(w.H1, 6, w.cov, lane);
    if (lane == 0) w.rank1 = rank;
    ICET_WSYNC();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) *pinv_done = 1;
}
#undef ICET_TS
#undef ICET_WSYNC

}  // namespace icetdev
