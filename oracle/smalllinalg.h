// oracle/smalllinalg.h -- TEST INFRASTRUCTURE (parity oracle), not product code.
//
// The reference (mcdermatt/ICET) does all of its small dense algebra through Eigen, which is
// NOT vendored in /root/reference and not present in this image (ROS Noetic / Ubuntu 20.04 would
// supply Eigen 3.3.7).  This header restates, from Eigen 3.3.7's published algorithms, exactly the
// three decompositions the reference calls on the hot path:
//
//   * SelfAdjointEigenSolver<Matrix3f>              src/icet.cpp:181-183   (3x3, closed-form tridiagonalisation + implicit QR)
//   * SelfAdjointEigenSolver<MatrixXf> (6x6)        src/icet.cpp:455-458   (Householder tridiagonalisation + implicit QR)
//   * CompleteOrthogonalDecomposition::pseudoInverse src/icet.cpp:320-321, 410-411, 428-429
//     (column-pivoted Householder QR, rank threshold eps*min(rows,cols) relative to the largest pivot)
//
// PARITY UNPINNED against real Eigen: Eigen cannot be compiled here, so sign conventions of
// eigenvectors (which matter for Q9/Q12 of SURVEY.md section 8) follow this restatement.  The
// eigenvalues / pseudo-inverses themselves are cross-checked against numpy in tests/.
#pragma once
#include <cmath>
#include <cfloat>
#include <cstring>
#include <algorithm>
#include <utility>

namespace ico {

// Tiny row-major matrix, max 6x6.
struct Mat {
    int r = 0, c = 0;
    float a[36];
    Mat() { std::memset(a, 0, sizeof(a)); }
    Mat(int r_, int c_) : r(r_), c(c_) { std::memset(a, 0, sizeof(a)); }
    float& operator()(int i, int j) { return a[i * c + j]; }
    float operator()(int i, int j) const { return a[i * c + j]; }
    static Mat identity(int n) { Mat m(n, n); for (int i = 0; i < n; i++) m(i, i) = 1.f; return m; }
};

inline Mat matmul(const Mat& A, const Mat& B) {
    Mat C(A.r, B.c);
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < B.c; j++) {
            float s = 0.f;
            for (int k = 0; k < A.c; k++) s += A(i, k) * B(k, j);
            C(i, j) = s;
        }
    return C;
}
inline Mat transpose(const Mat& A) {
    Mat T(A.c, A.r);
    for (int i = 0; i < A.r; i++) for (int j = 0; j < A.c; j++) T(j, i) = A(i, j);
    return T;
}

// Eigen Householder.h makeHouseholder (real case) on x[0..m-1] with stride; essential part is
// written in place over x[1..m-1]; returns tau and beta.
inline void make_householder(float* x, int m, int stride, float& tau, float& beta) {
    float tailSq = 0.f;
    for (int i = 1; i < m; i++) tailSq += x[i * stride] * x[i * stride];
    float c0 = x[0];
    const float tol = FLT_MIN;
    if (tailSq <= tol) {
        tau = 0.f; beta = c0;
        for (int i = 1; i < m; i++) x[i * stride] = 0.f;
    } else {
        beta = std::sqrt(c0 * c0 + tailSq);
        if (c0 >= 0.f) beta = -beta;
        for (int i = 1; i < m; i++) x[i * stride] = x[i * stride] / (c0 - beta);
        tau = (beta - c0) / beta;
    }
}

// Eigen Jacobi.h JacobiRotation::makeGivens (real case).
inline void make_givens(float p, float q, float& c, float& s) {
    if (q == 0.f) { c = p < 0.f ? -1.f : 1.f; s = 0.f; }
    else if (p == 0.f) { c = 0.f; s = q < 0.f ? 1.f : -1.f; }
    else if (std::fabs(p) > std::fabs(q)) {
        float t = q / p; float u = std::sqrt(1.f + t * t); if (p < 0.f) u = -u;
        c = 1.f / u; s = -t * c;
    } else {
        float t = p / q; float u = std::sqrt(1.f + t * t); if (q < 0.f) u = -u;
        s = -1.f / u; c = -t * s;
    }
}

// Eigen 3.3.7 numext::hypot (MathFunctions.h, hypot_impl): p * sqrt(1 + (q/p)^2) with p = max(|x|,|y|) -- plain IEEE
// operations, so the device kernel can reproduce it bit for bit (libm's hypotf is not specified to the last bit).
inline float eigen_hypot(float x, float y) {
    float ax = std::fabs(x), ay = std::fabs(y);
    float p, qp;
    if (ax > ay) { p = ax; qp = ay / p; } else { p = ay; qp = ax / p; }
    if (p == 0.f) return 0.f;
    return p * std::sqrt(1.f + qp * qp);
}

// Eigen SelfAdjointEigenSolver.h tridiagonal_qr_step; Q is n x n row-major, rotation applied on the
// right to columns k,k+1 (q.applyOnTheRight(k,k+1,rot)).  libm_hypot: std::hypot instead of Eigen's formula.
inline void tridiagonal_qr_step(float* diag, float* subdiag, int start, int end, float* Q, int n, bool libm_hypot = false) {
    float td = (diag[end - 1] - diag[end]) * 0.5f;
    float e = subdiag[end - 1];
    float mu = diag[end];
    if (td == 0.f) {
        mu -= std::fabs(e);
    } else if (e != 0.f) {
        const float e2 = e * e;
        const float h = libm_hypot ? std::hypot(td, e) : eigen_hypot(td, e);
        if (e2 == 0.f) mu -= e / ((td + (td > 0.f ? h : -h)) / e);
        else           mu -= e2 / (td + (td > 0.f ? h : -h));
    }
    float x = diag[start] - mu;
    float z = subdiag[start];
    for (int k = start; k < end && z != 0.f; ++k) {
        float c, s; make_givens(x, z, c, s);
        float sdk  = s * diag[k] + c * subdiag[k];
        float dkp1 = s * subdiag[k] + c * diag[k + 1];
        diag[k]     = c * (c * diag[k] - s * subdiag[k]) - s * (c * subdiag[k] - s * diag[k + 1]);
        diag[k + 1] = s * sdk + c * dkp1;
        subdiag[k]  = c * sdk - s * dkp1;
        if (k > start) subdiag[k - 1] = c * subdiag[k - 1] - s * z;
        x = subdiag[k];
        if (k < end - 1) { z = -s * subdiag[k + 1]; subdiag[k + 1] = c * subdiag[k + 1]; }
        if (Q) {
            for (int i = 0; i < n; i++) {
                float xi = Q[i * n + k], yi = Q[i * n + k + 1];
                Q[i * n + k]     = c * xi - s * yi;
                Q[i * n + k + 1] = s * xi + c * yi;
            }
        }
    }
}

// SelfAdjointEigenSolver::compute (ComputeEigenvectors) restated for n = 3 (fixed-size closed-form
// tridiagonalisation, Tridiagonalization.h tridiagonalization_inplace_selector<MatrixType,3,false>)
// and for general n <= 6 (Householder tridiagonalisation).  Only the lower triangle of A is read.
// Output: evals ascending, evecs columns (row-major n x n).  Returns false on no-convergence.
inline bool selfadjoint_eigen(const Mat& Ain, bool fixed3, float* evals, Mat& evecs, bool libm_hypot = false) {
    const int n = Ain.r;
    Mat A(n, n);
    float scale = 0.f;
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { A(i, j) = Ain(i, j); scale = std::max(scale, std::fabs(Ain(i, j))); }
    if (scale == 0.f) scale = 1.f;
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) A(i, j) /= scale;
    float diag[6], sub[6] = {0, 0, 0, 0, 0, 0};
    Mat Q = Mat::identity(n);
    if (n == 1) { evals[0] = Ain(0, 0); evecs = Q; return true; }
    if (fixed3 && n == 3) {
        diag[0] = A(0, 0);
        float v1norm2 = A(2, 0) * A(2, 0);
        if (v1norm2 <= FLT_MIN) {
            diag[1] = A(1, 1); diag[2] = A(2, 2); sub[0] = A(1, 0); sub[1] = A(2, 1);
        } else {
            float beta = std::sqrt(A(1, 0) * A(1, 0) + v1norm2);
            float invBeta = 1.f / beta;
            float m01 = A(1, 0) * invBeta, m02 = A(2, 0) * invBeta;
            float q = 2.f * m01 * A(2, 1) + m02 * (A(2, 2) - A(1, 1));
            diag[1] = A(1, 1) + m02 * q;
            diag[2] = A(2, 2) - m02 * q;
            sub[0] = beta;
            sub[1] = A(2, 1) - m01 * q;
            Q(1, 1) = m01; Q(1, 2) = m02; Q(2, 1) = m02; Q(2, 2) = -m01;
        }
    } else {
        float h[6];
        for (int i = 0; i < n - 1; i++) {
            int rem = n - i - 1;
            float tau, beta;
            make_householder(&A.a[(i + 1) * n + i], rem, n, tau, beta);
            A(i + 1, i) = 1.f;
            float v[6], p[6];
            for (int k = 0; k < rem; k++) v[k] = A(i + 1 + k, i);
            // p = tau * (A22_sym * v)   (selfadjointView<Lower>)
            for (int a = 0; a < rem; a++) {
                float s = 0.f;
                for (int b = 0; b < rem; b++) {
                    int ra = i + 1 + a, rb = i + 1 + b;
                    float m = (ra >= rb) ? A(ra, rb) : A(rb, ra);
                    s += m * (tau * v[b]);
                }
                p[a] = s;
            }
            float dot = 0.f; for (int k = 0; k < rem; k++) dot += p[k] * v[k];
            float alpha = tau * -0.5f * dot;
            for (int k = 0; k < rem; k++) p[k] += alpha * v[k];
            // rankUpdate(v, p, -1): A22 -= v p^T + p v^T on the lower triangle
            for (int a = 0; a < rem; a++) for (int b = 0; b <= a; b++)
                A(i + 1 + a, i + 1 + b) -= v[a] * p[b] + p[a] * v[b];
            A(i + 1, i) = beta;
            h[i] = tau;
        }
        for (int i = 0; i < n; i++) diag[i] = A(i, i);
        for (int i = 0; i < n - 1; i++) sub[i] = A(i + 1, i);
        // Q = H_0 H_1 ... H_{n-2}; apply from the last on the left of the identity.
        for (int k = n - 2; k >= 0; k--) {
            int rem = n - k - 1;               // v lives in rows k+1..n-1, v[0]=1
            float v[6]; v[0] = 1.f;
            for (int t = 1; t < rem; t++) v[t] = A(k + 1 + t, k);
            for (int col = 0; col < n; col++) {
                float s = 0.f;
                for (int t = 0; t < rem; t++) s += v[t] * Q(k + 1 + t, col);
                s *= h[k];
                for (int t = 0; t < rem; t++) Q(k + 1 + t, col) -= s * v[t];
            }
        }
    }
    // computeFromTridiagonal_impl (Eigen 3.3.7)
    int end = n - 1, start = 0, iter = 0;
    const int maxIterations = 30;
    const float considerAsZero = FLT_MIN;
    const float precision = 2.f * FLT_EPSILON;
    while (end > 0) {
        for (int i = start; i < end; ++i)
            if (std::fabs(sub[i]) <= (std::fabs(diag[i]) + std::fabs(diag[i + 1])) * precision || std::fabs(sub[i]) <= considerAsZero)
                sub[i] = 0.f;
        while (end > 0 && sub[end - 1] == 0.f) end--;
        if (end <= 0) break;
        iter++;
        if (iter > maxIterations * n) break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0.f) start--;
        tridiagonal_qr_step(diag, sub, start, end, Q.a, n, libm_hypot);
    }
    bool ok = iter <= maxIterations * n;
    if (ok) {
        for (int i = 0; i < n - 1; ++i) {
            int k = 0; float mn = diag[i];
            for (int j = i + 1; j < n; j++) if (diag[j] < mn) { mn = diag[j]; k = j - i; }
            if (k > 0) {
                std::swap(diag[i], diag[k + i]);
                for (int rr = 0; rr < n; rr++) std::swap(Q(rr, i), Q(rr, k + i));
            }
        }
    }
    for (int i = 0; i < n; i++) evals[i] = diag[i] * scale;
    evecs = Q;
    return ok;
}

// CompleteOrthogonalDecomposition<MatrixXf>(A).pseudoInverse(), A is rows x cols (<= 6x6).
// Column-pivoted Householder QR exactly as ColPivHouseholderQR::computeInPlace (norm down-dating
// included), rank = #{|R_ii| > eps*min(rows,cols)*maxpivot} among the non-zero pivots.  For
// rank == cols the result is P R^-1 Q^T evaluated in float; for rank < cols the minimum-norm
// completion X^+ = X^T (X X^T)^-1 of X = [R11 R12] is evaluated in double (mathematically what
// the Z-reflector stage of Eigen's COD yields).  *rank_out receives the rank.
inline Mat cod_pinv(const Mat& Ain, int* rank_out = nullptr) {
    const int rows = Ain.r, cols = Ain.c, size = std::min(rows, cols);
    Mat qr = Ain;
    float hc[6]; int perm[6];
    float normsUpd[6], normsDir[6];
    for (int k = 0; k < cols; k++) {
        float s = 0.f; for (int i = 0; i < rows; i++) s += qr(i, k) * qr(i, k);
        normsDir[k] = std::sqrt(s); normsUpd[k] = normsDir[k]; perm[k] = k;
    }
    float maxn = 0.f; for (int k = 0; k < cols; k++) maxn = std::max(maxn, normsUpd[k]);
    float th = maxn * FLT_EPSILON; const float threshold_helper = (th * th) / float(rows);
    const float norm_downdate_threshold = std::sqrt(FLT_EPSILON);
    int nonzero_pivots = size; float maxpivot = 0.f;
    for (int k = 0; k < size; k++) {
        int big = k; float bn = normsUpd[k];
        for (int j = k + 1; j < cols; j++) if (normsUpd[j] > bn) { bn = normsUpd[j]; big = j; }
        float big_sq = bn * bn;
        if (nonzero_pivots == size && big_sq < threshold_helper * float(rows - k)) nonzero_pivots = k;
        if (k != big) {
            for (int i = 0; i < rows; i++) std::swap(qr(i, k), qr(i, big));
            std::swap(normsUpd[k], normsUpd[big]); std::swap(normsDir[k], normsDir[big]);
            std::swap(perm[k], perm[big]);
        }
        float tau, beta;
        make_householder(&qr.a[k * cols + k], rows - k, cols, tau, beta);
        qr(k, k) = beta;
        if (std::fabs(beta) > maxpivot) maxpivot = std::fabs(beta);
        hc[k] = tau;
        // apply H = I - tau v v^T (v = [1; essential]) to the bottom-right corner
        for (int j = k + 1; j < cols; j++) {
            float s = qr(k, j);
            for (int i = k + 1; i < rows; i++) s += qr(i, k) * qr(i, j);
            s *= tau;
            qr(k, j) -= s;
            for (int i = k + 1; i < rows; i++) qr(i, j) -= s * qr(i, k);
        }
        for (int j = k + 1; j < cols; ++j) {
            if (normsUpd[j] != 0.f) {
                float temp = std::fabs(qr(k, j)) / normsUpd[j];
                temp = (1.f + temp) * (1.f - temp);
                temp = temp < 0.f ? 0.f : temp;
                float ratio = normsUpd[j] / normsDir[j];
                float temp2 = temp * ratio * ratio;
                if (temp2 <= norm_downdate_threshold) {
                    float s = 0.f; for (int i = k + 1; i < rows; i++) s += qr(i, j) * qr(i, j);
                    normsDir[j] = std::sqrt(s); normsUpd[j] = normsDir[j];
                } else {
                    normsUpd[j] *= std::sqrt(temp);
                }
            }
        }
    }
    const float premult = std::fabs(maxpivot) * (FLT_EPSILON * float(size));
    int rank = 0;
    for (int i = 0; i < nonzero_pivots; i++) rank += (std::fabs(qr(i, i)) > premult) ? 1 : 0;
    if (rank_out) *rank_out = rank;
    Mat pinv(cols, rows);
    if (rank == 0) return pinv;
    // C = Q^T restricted to the first `rank` reflectors, applied to I (rows x rows)
    Mat C = Mat::identity(rows);
    for (int k = 0; k < rank; k++) {
        for (int j = 0; j < rows; j++) {
            float s = C(k, j);
            for (int i = k + 1; i < rows; i++) s += qr(i, k) * C(i, j);
            s *= hc[k];
            C(k, j) -= s;
            for (int i = k + 1; i < rows; i++) C(i, j) -= s * qr(i, k);
        }
    }
    Mat Y(cols, rows);   // permuted solution
    if (rank == cols) {
        for (int j = 0; j < rows; j++)
            for (int i = rank - 1; i >= 0; i--) {
                float s = C(i, j);
                for (int t = i + 1; t < rank; t++) s -= qr(i, t) * Y(t, j);
                Y(i, j) = s / qr(i, i);
            }
    } else {
        // X = [R11 R12] (rank x cols); X^+ = X^T (X X^T)^-1; Y = X^+ * C(0:rank,:)
        double G[36], Ginv[36];
        for (int i = 0; i < rank; i++) for (int j = 0; j < rank; j++) {
            double s = 0; for (int t = 0; t < cols; t++) { double xi = (t >= i) ? qr(i, t) : 0.0, xj = (t >= j) ? qr(j, t) : 0.0; s += xi * xj; }
            G[i * rank + j] = s; Ginv[i * rank + j] = (i == j);
        }
        for (int p = 0; p < rank; p++) {          // Gauss-Jordan with partial pivoting (G is SPD)
            int piv = p; for (int i = p + 1; i < rank; i++) if (std::fabs(G[i * rank + p]) > std::fabs(G[piv * rank + p])) piv = i;
            if (piv != p) for (int j = 0; j < rank; j++) { std::swap(G[p * rank + j], G[piv * rank + j]); std::swap(Ginv[p * rank + j], Ginv[piv * rank + j]); }
            double d = G[p * rank + p];
            for (int j = 0; j < rank; j++) { G[p * rank + j] /= d; Ginv[p * rank + j] /= d; }
            for (int i = 0; i < rank; i++) if (i != p) {
                double f = G[i * rank + p];
                for (int j = 0; j < rank; j++) { G[i * rank + j] -= f * G[p * rank + j]; Ginv[i * rank + j] -= f * Ginv[p * rank + j]; }
            }
        }
        for (int t = 0; t < cols; t++) for (int j = 0; j < rows; j++) {
            double s = 0;
            for (int i = 0; i < rank; i++) {
                double xit = (t >= i) ? qr(i, t) : 0.0;
                double w = 0; for (int m = 0; m < rank; m++) w += Ginv[i * rank + m] * C(m, j);
                s += xit * w;
            }
            Y(t, j) = (float)s;
        }
    }
    for (int k = 0; k < cols; k++) for (int j = 0; j < rows; j++) pinv(perm[k], j) = Y(k, j);
    return pinv;
}

}  // namespace ico
