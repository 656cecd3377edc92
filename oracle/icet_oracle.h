/* oracle/icet_oracle.h -- TEST INFRASTRUCTURE (parity oracle + timed CPU "port" baseline).
 * C ABI of the CPU restatement in icet_oracle.cpp.  Not part of the product; see the header of
 * icet_oracle.cpp for what it follows in /root/reference and for the "parity unpinned" statement. */
#ifndef ICET_ORACLE_H
#define ICET_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ICET_ORACLE_SERIAL = 0,   /* live path: serial voxel loop, src/icet.cpp:391-404            */
       ICET_ORACLE_POOL4  = 1,   /* parallelFitCells2 structure, 4 workers, src/icet.cpp:346-370,31 */
       ICET_ORACLE_TRUE_SORT = 2,   /* OR-ed in: non-parity extension, rows really sorted by range (twin of ICET_FLAG_TRUE_SORT) */
       ICET_ORACLE_LIBMF = 4,     /* OR-ed in: glibc FLOAT atan2/acos/sin/cos, sequential float sums and std::hypot -- the literal
                                       expression types of the reference source -- instead of the shared arithmetic rule (correctly
                                       rounded transcendentals, exact sums, Eigen's hypot; icet_oracle.cpp header) */
       ICET_ORACLE_SKIP_RT2 = 8,    /* EXPERIMENT (quantifies the device's one documented deviation): scan 2 is never round-tripped through
                                       spherical coordinates (src/icet.cpp:275, 303) */
       ICET_ORACLE_RT2_THIN = 16, /* EXPERIMENT: as SKIP_RT2, but voxels whose scan-1 Gaussian is thin (lambda_min < 1e-5 m^2) keep the round trips */
       ICET_ORACLE_REJECT_MOVING = 32, /* OR-ed in: non-parity extension, twin of ICET_FLAG_REJECT_MOVING: from iteration 4 on a voxel whose
                                       compact residual L U^T (mu2 - mu1) exceeds 0.3 m in a kept axis is skipped
                                       (python/ICET_spherical.py:175-250, RM_thresh :38, start_RM_iter :36) */
       ICET_ORACLE_HALF_GAP = 64,    /* OR-ed in: non-parity extension, twin of ICET_FLAG_HALF_GAP_BOUNDS (implies TRUE_SORT): cluster bounds reach
                                       half way to the nearest point outside the cluster, at most buff (python/utils.py:92-119) */
       ICET_ORACLE_DEVICE_ARITH = 128 };  /* ATTRIBUTION EXPERIMENT (implies SKIP_RT2): every documented last-bit deviation of the HIP path at once -- scan 2 not
                                       round-tripped; the transform (p + t) R as three FMAs per coordinate, innermost product first (icet_device_common.h transform_point:
                                       what Eigen's product kernel does on an FMA machine); the scan-2 moments as sums of d = q - mu1 and d d^T accumulated exactly and
                                       centred afterwards in double (icet_solve_body.h); dz = L U^T (mu2 - mu1) from that difference instead of L U^T mu2 - L U^T mu1.
                                       What is then left between device and oracle is the float rounding of the device's per-run partial sums (tests/test_gpu_parity.py
                                       test_attribution_*) */
       /* OR-ed in, ATTRIBUTION EXPERIMENT: the per-voxel weight W = pinv(L U^T R_noise U L^T) (src/icet.cpp:320-321) as the HIP path takes it -- in DOUBLE, rank by
          eigenvalues above 3 eps x the largest (icet_device_math.h pinv3_sym) -- instead of the reference's float CompleteOrthogonalDecomposition, whose result carries a
          relative error of cond x eps: per cents on the cond 1e6 .. 1e7 voxels of a millimetre-noise scene */
#define ICET_ORACLE_PINV3_DOUBLE 256

typedef struct icet_oracle_params {
    int32_t runlen;      /* include/icet.h:38  */
    int32_t bins_phi;    /* numBinsPhi   (elevation / polar-angle bins, 24) */
    int32_t bins_theta;  /* numBinsTheta (azimuth bins, 75)                 */
    int32_t n;           /* min cluster size, default 25  */
    float   thresh;      /* radial jump threshold, 0.1    */
    float   buff;        /* radial buffer, 0.1            */
    int32_t mode;        /* ICET_ORACLE_SERIAL / _POOL4   */
} icet_oracle_params;

/* Caller-allocated dump of intermediate state (any pointer-holding struct must be fully populated).
 * V = bins_phi*bins_theta, voxel index v = bins_theta*phi + theta (src/icet.cpp:149). */
typedef struct icet_oracle_trace {
    int32_t max_iters;       /* capacity of the per-iteration arrays */
    int32_t n_ub_voxels;     /* out: voxels where the reference would read an un-fitted map entry */
    /* keyframe (scan 1) table */
    int32_t* n1_raw;         /* V      |pointIndices1[theta][phi]|            */
    int32_t* has_fit;        /* V      1 if sigma1/mu1/U/L exist              */
    float*   bounds;         /* V x 6  clusterBounds                          */
    float*   mu1;            /* V x 3                                         */
    float*   sigma1;         /* V x 9                                         */
    float*   evecs1;         /* V x 9  eigenvectors as columns (row-major)    */
    float*   Ldiag;          /* V x 3                                         */
    float*   sigma_points;   /* V x 6 x 3                                     */
    /* per iteration */
    int32_t* n2_raw;         /* iters x V   |pointIndices2|                   */
    int32_t* n2_in;          /* iters x V   rows passing filterPointsInsideCluster (0 if gated off earlier) */
    int32_t* used;           /* iters x V   voxel contributed                 */
    float*   mu2;            /* iters x V x 3 */
    float*   sigma2;         /* iters x V x 9 */
    float*   HTWH;           /* iters x 36 */
    float*   HTWdz;          /* iters x 6  */
    float*   dx;             /* iters x 6  */
    float*   X;              /* iters x 6  (after the update) */
    float*   eigvals;        /* iters x 6  */
    int32_t* pruned;         /* iters      axes dropped by checkCondition */
} icet_oracle_trace;

/* scan = N x 3 column-major float (Eigen::MatrixXf layout: x[N] | y[N] | z[N], leading dimension ld). */
int icet_oracle_solve(const icet_oracle_params* p, const float* scan1, int64_t n1, int64_t ld1,
                      const float* scan2, int64_t n2, int64_t ld2, const float x0[6],
                      float x_out[6], float pred_stds_out[6], float cov_out[36], icet_oracle_trace* trace);

/* TESTS ONLY: the same solve with every fitted voxel's eigenvector signs aligned to evecs_ref (V x 9, eigenvectors as columns,
 * row-major; zero rows ignored).  Separates implementation-defined sign differences from algebra differences. */
int icet_oracle_solve_signed(const icet_oracle_params* p, const float* scan1, int64_t n1, int64_t ld1,
                             const float* scan2, int64_t n2, int64_t ld2, const float x0[6], const float* evecs_ref,
                             float x_out[6], float pred_stds_out[6], float cov_out[36], icet_oracle_trace* trace, int32_t* n_flips);

int icet_oracle_solve_batch(const icet_oracle_params* p, int n_pairs, const float* const* scan1, const int64_t* n1,
                            const float* const* scan2, const int64_t* n2, const float* x0,
                            float* x_out, float* pred_stds_out, float* cov_out, int n_threads);

double icet_oracle_time_pair(const icet_oracle_params* p, const float* scan1, int64_t n1, const float* scan2, int64_t n2,
                             const float x0[6], int reps, float x_out[6]);

void icet_oracle_eig_sym(const float* A, int n, int fixed3, float* evals, float* evecs);
int  icet_oracle_pinv(const float* A, int rows, int cols, float* out);
void icet_oracle_c2s(const float* xyz, int64_t n, int64_t ld, float* out);
void icet_oracle_scramble(const float* r, int64_t n, int32_t* src);
int  icet_oracle_gn_tail(const float* HTWH, const float* HTWdz, int libmf, float* out54, int32_t* rank);   /* icet.cpp:410-430 on its own: cov[36] | pred_stds[6] | dx[6] | eigenvalues[6]; returns pruned axes */
void icet_oracle_get_H(const float mu[3], const float angs[3], float* H18);
void icet_oracle_R(const float angs[3], float* R9);

/* ---- the callers around the constructor (icet_nodes_oracle.cpp): odometry.cpp:46-98, simpleMapMaker.cpp:18-59,86-172 ---- */
typedef struct icet_oracle_node_params {
    icet_oracle_params solve;
    float   min_range;       /* 2.0 odometry.cpp:58 / 0.2 simpleMapMaker.cpp:98 */
    int32_t seed_x0;         /* 1 odometry.cpp:82 / 0 simpleMapMaker.cpp:124    */
    float   trans_thresh;    /* simpleMapMaker.cpp:241-242; both <= 0: no guard  */
    float   rot_thresh;
    int32_t map_capacity;    /* 600000 simpleMapMaker.cpp:62; 0 = no map         */
    int32_t map_downsample;  /* 2000 simpleMapMaker.cpp:147                      */
    int32_t flags;           /* 1 no range filter, 2 aligned cloud, 4 snail trail: scanMatcher.cpp:44,76,79-84 */
} icet_oracle_node_params;

typedef struct icet_oracle_node_result {
    int32_t solved, diverged;
    int64_t n_kept;
    float   X[6], pred_stds[6], pose[16], quat[4];
    int64_t map_rows;
} icet_oracle_node_result;

void*   icet_oracle_node_create(const icet_oracle_node_params* p);
void    icet_oracle_node_destroy(void* node);
int     icet_oracle_node_push(void* node, const float* scan, int64_t n, int64_t ld, icet_oracle_node_result* res);
int64_t icet_oracle_node_map(void* node, float* out, int64_t ld);   /* rows; out may be NULL */
int64_t icet_oracle_node_aligned(void* node, float* out, int64_t ld);      /* scanMatcher.cpp:76   */
int64_t icet_oracle_node_snail_trail(void* node, float* out, int64_t ld);  /* scanMatcher.cpp:79-84 */

#ifdef __cplusplus
}
#endif
#endif
