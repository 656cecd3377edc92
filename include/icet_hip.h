/* include/icet_hip.h -- C ABI of the MI355X-native ICET hot path (libicet_hip.so).
 *
 * This is the drop-in boundary for the ONE path this project accelerates: the body of the reference's
 * `ICET::ICET(...)` constructor (/root/reference/src/icet.cpp:29-63, declared include/icet.h:38-40),
 * i.e. fitScan1 -> prepScan2 -> runlen x fitScan2, whose only outputs the callers read are the public
 * members `X` and `pred_stds` (src/odometry.cpp:76-79,126-131; src/simpleMapMaker.cpp:119-122).
 * The reference has no FFI; a maintainer would bind these entry points from the constructor (see
 * INTEGRATION.md for the exact stub).  Plain pointers and sizes only -- no Eigen, torch or HIP types.
 *
 * Scan layout everywhere: N x 3 float32 COLUMN-MAJOR with leading dimension ld >= N, i.e. exactly
 * `Eigen::MatrixXf::data()` of the reference's `MatrixXf& scan` arguments: x[0..N) | y[0..N) | z[0..N).
 *
 * All entry points return an icet_status; none throws, none aborts.  The library fails loudly
 * (ICET_ERR_NO_DEVICE / ICET_ERR_HIP) when no gfx950 device or kernel image is usable -- there is no
 * CPU fallback behind this ABI.
 */
#ifndef ICET_HIP_H
#define ICET_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum icet_status {
    ICET_OK = 0,
    ICET_ERR_BAD_ARG = 1,     /* null pointer, negative size, ld < n, bins <= 0, ...            */
    ICET_ERR_NO_DEVICE = 2,   /* no HIP device / device id out of range                          */
    ICET_ERR_HIP = 3,         /* a HIP runtime call or kernel launch failed (see icet_last_error) */
    ICET_ERR_NOMEM = 4,       /* device or host allocation failed                                 */
    ICET_ERR_UNSUPPORTED = 5  /* e.g. bins_phi*bins_theta above the voxel limit (10000: the voxel tables of one pair live in one block's LDS) */
} icet_status;

/* Mirrors the reference constructor's scalar arguments (include/icet.h:38-40, defaults n=25,
 * thresh=0.1, buff=0.1; call sites src/odometry.cpp:73-76, src/simpleMapMaker.cpp:113-119). */
typedef struct icet_params {
    int32_t runlen;       /* Gauss-Newton iterations (`runlen`, 7 or 12 at the call sites)       */
    int32_t bins_phi;     /* `num_bins_phi`   : polar-angle ("elevation") bins, 24               */
    int32_t bins_theta;   /* `num_bins_theta` : azimuth bins, 75                                  */
    int32_t n;            /* minimum points per cluster                                           */
    float   thresh;       /* radial jump threshold of findCluster (src/icet.cpp:557)              */
    float   buff;         /* radial buffer added to the cluster bounds                            */
    int32_t flags;        /* ICET_FLAG_* below; 0 = reference behaviour                           */
} icet_params;

enum { ICET_FLAG_NONE = 0,
       ICET_FLAG_TIMING = 1,  /* record HIP events around every bin/accumulate launch (icet_last_timing[2]) */
       ICET_FLAG_TRUE_SORT = 2, /* NON-PARITY EXTENSION: really sort scan 1 by range before clustering, instead of reproducing
                                  the reference's one-step swap loop (src/icet.cpp:78-83), which leaves the rows scrambled so
                                  that findCluster sees most bins in a shuffled order and only ~1/4 of the populated bins get a
                                  Gaussian.  Results then differ from the reference by design (more voxels, better conditioned);
                                  the oracle has the same switch so that the extension is still checked against a CPU twin. */
       ICET_FLAG_REJECT_MOVING = 4, /* NON-PARITY EXTENSION: moving-object rejection as in the reference's Python
                                  variant (python/ICET_spherical.py:175-250, the hard cutoff that is live there): from the 5th iteration on
                                  (start_RM_iter = 4), a voxel whose compact residual L U^T (mu2 - mu1) exceeds RM_thresh = 0.3 m in any
                                  kept axis is left out of that iteration's H^T W H and H^T W dz.  The C++ reference has nothing like it, so
                                  results differ from it by design; the oracle has the same switch (ICET_ORACLE_REJECT_MOVING). */
       ICET_FLAG_HALF_GAP_BOUNDS = 8, /* NON-PARITY EXTENSION: the cluster buffers the reference's Python variant
                                  DESCRIBES ("as described in spherical paper", the comments at python/utils.py:92-119 -- the TensorFlow
                                  code there adds the FULL gap when 2*gap < max_buffer, which is not this rule): the radial bounds of a
                                  voxel's cluster reach half way to the nearest point outside it, at most `buff`, instead of `buff` on either side (a neighbour
                                  that exists is more than `thresh` away, so this only tightens bounds whose neighbour lies within 2 buff).
                                  Needs the voxel's rows in ascending range, so it implies ICET_FLAG_TRUE_SORT.  Oracle twin:
                                  ICET_ORACLE_HALF_GAP. */
       ICET_FLAG_ROUNDTRIP_SCAN2 = 16, /* PARITY-STUDY OPTION (not an extension: it makes the loop MORE literal).  The reference passes scan 2 through
                                  cartesianToSpherical -> sphericalToCartesian twice: once as a whole (points2_OG, src/icet.cpp:275) and, every
                                  iteration, the in-bounds points of every voxel (:303).  Each trip moves a point by 1-2 ulp; the default path
                                  skips both (one of several last-bit triggers behind the pairs that differ from the CPU restatement by
                                  > 5e-5 m; restoring them closes some of those pairs, not all).  With this flag both trips are made, under the shared arithmetic rule (correctly rounded angles and
                                  sines / cosines): a pre-pass over scan 2, and a double-precision atan2 + acos per in-bounds point per
                                  iteration -- about twice the loop time.  Decisions (which voxel, inside the bounds) are unchanged. */
       ICET_FLAG_DOUBLE_W = 32 /* ACCURACY OPTION (the default of rounds 2-5).  The per-voxel weight W = pinv(L U^T R_noise U L^T) (src/icet.cpp:317-321) is, in the reference,
                                  Eigen's float CompleteOrthogonalDecomposition of a float matrix: its result carries a relative error of cond x eps and its rank decision is
                                  the pivot rule.  Since round 6 the device does exactly that by default (Eigen 3.3's algorithm statement by statement on the full, unsymmetrised
                                  3 x 3, bit-identical to the CPU restatement's function on the same matrix: tests test_pinv3_reference_bits).  With this flag W is taken in
                                  DOUBLE with an eigenvalue rank rule instead -- the better-conditioned answer.  On lidar scenes the two agree to rounding; on voxels whose
                                  covariance is thin to the point of cond 1e6 .. 1e7 (millimetre-noise scenes, clusters of three points) they do not: H^T W H differs by 10 %
                                  and more, a condition number at checkCondition's cutoff lands on the other side (other pruned axes, pred_stds of 1e-4 instead of -0.9995),
                                  while the reference's own answer moves by per cents with the last bit of its input. */ };

/* A scan that already lives in device memory (HBM) on the context's device. */
typedef struct icet_dev_scan {
    const float* ptr;     /* device pointer, column-major N x 3, 16-byte aligned                  */
    int64_t n;            /* points                                                               */
    int64_t ld;           /* leading dimension in floats (>= n)                                   */
} icet_dev_scan;

/* Optional side outputs of a single-pair solve: what the reference object exposes besides X and
 * pred_stds (include/icet.h:78-107).  Any pointer may be NULL.  V = bins_phi * bins_theta, voxel
 * row index v = bins_theta * phi + theta (src/icet.cpp:149). */
typedef struct icet_aux {
    float*   cluster_bounds;  /* V x 6 row-major: azMin,azMax,elMin,elMax,inner,outer  (`clusterBounds`) */
    int32_t* n1_raw;          /* V: scan-1 points per angular bin (|pointIndices1[theta][phi]|)          */
    int32_t* has_fit;         /* V: 1 where mu1/sigma1/U/L exist                                          */
    float*   mu1;             /* V x 3   (`mu1`)                                                          */
    float*   sigma1;          /* V x 9   (`sigma1`, row-major 3x3)                                        */
    float*   evecs1;          /* V x 9   eigenvectors as columns (reference stores U = transpose)        */
    float*   l_diag;          /* V x 3   diagonal of `L`                                                  */
    float*   x_hist;          /* runlen x 6: X after every iteration; X before the last update gives
                                 the transform of the reference's final `points2` member               */
    float*   htwh;            /* runlen x 36 (`HTWH_i` per iteration)                                     */
    float*   htwdz;           /* runlen x 6  (`HTWdz_i` per iteration)                                    */
    int32_t* n2_raw;          /* runlen x V: |pointIndices2| (only voxels with a scan-1 fit are counted) */
    int32_t* n2_in;           /* runlen x V: scan-2 points inside the voxel's cluster bounds             */
    float*   test_points;     /* (V x 6) x 3 row-major (`testPoints`, src/icet.cpp:41,213-231): rows 6v + 2k, 6v + 2k + 1 hold
                                 the two sigma points of axis k of voxel v when that axis was pruned (L row k = 0); every
                                 other row is zero (the reference leaves those rows uninitialised)                          */
    float*   points2;         /* n2 x 3 COLUMN-major, leading dimension n2 (`points2`, include/icet.h:80): scan 2 as the last
                                 fitScan2 transformed it, (p + t) * R with the X before the final update (src/icet.cpp:375-378
                                 precede :433), in the CALLER's row order (the device never sorts scan 2); scan 2 itself when runlen == 0 */
    /* The per-point members of the reference object (include/icet.h:79,82,95-96).  No caller in the reference reads them; they cost
     * extra kernels and several MB of D2H, so they are produced only when asked for. */
    float*   points1_spherical; /* n1 x 3 COLUMN-major (r | theta | phi), leading dimension n1 (`points1Spherical`): row p is the scan-1
                                   point that sits at position p after the reference's radial sort + swap loop (src/icet.cpp:69-83)   */
    int32_t* point_index1;      /* n1: `pointIndices1` flattened -- the positions (rows of points1_spherical) of voxel v, ascending,
                                   are point_index1[bin_start1[v] .. bin_start1[v + 1])  (src/icet.cpp:86, 534-554)                   */
    int32_t* bin_start1;        /* V + 1                                                                                              */
    float*   points2_spherical; /* n2 x 3 COLUMN-major (r | theta | phi) of `points2` (`points2Spherical` after the last fitScan2,
                                   src/icet.cpp:387), caller's row order                                                              */
    int32_t* voxel2;            /* n2: the voxel sortSphericalCoordinates assigns to every row of points2 in the last fitScan2
                                   (src/icet.cpp:388): `pointIndices2[theta][phi]` = the rows i with voxel2[i] == T * phi + theta, ascending */
    float*   cond_info;         /* runlen x 8: what ICET::checkCondition (src/icet.cpp:443-492) saw in every iteration: [0,6) the eigenvalues of
                                   HTWH_i ascending (NaN when the Cholesky route proved the matrix well conditioned and never computed them),
                                   [6] the number of pruned axes (rows dropped from L2), [7] the route: 0 Cholesky, 2 the literal restatement */
} icet_aux;

typedef struct icet_ctx icet_ctx;   /* opaque: device id, stream, workspace */

/* Create / destroy a context bound to one device.  `hip_stream` is a hipStream_t passed as void*
 * (NULL = the context creates and owns a non-blocking stream).  One context is not re-entrant; use
 * one per host thread (the reference constructs one ICET object at a time per node: ros::spin()). */
icet_status icet_create(icet_ctx** ctx, int device_id, void* hip_stream);
icet_status icet_destroy(icet_ctx* ctx);
const char* icet_last_error(const icet_ctx* ctx);      /* static/ctx-owned string, never NULL */
const char* icet_version(void);

/* --- the constructor replacement: one scan pair, HOST pointers (copies in, solves, copies out) ----
 * Replaces ICET::ICET (src/icet.cpp:29-63).  x0 = `X0`; x_out = member `X`; pred_stds_out = member
 * `pred_stds`; cov_out (may be NULL) = the 6x6 `noise_mat` local of the last fitScan2
 * (src/icet.cpp:410-411), row-major. */
icet_status icet_solve(icet_ctx* ctx, const icet_params* p,
                       const float* scan1, int64_t n1, int64_t ld1,
                       const float* scan2, int64_t n2, int64_t ld2,
                       const float x0[6], float x_out[6], float pred_stds_out[6], float cov_out[36],
                       icet_aux* aux_or_null);

/* The same call in two halves.  icet_solve_begin enqueues the uploads of both scans, the whole registration and the copy of the
 * results and returns without waiting for the device; icet_solve_end waits and fills x_out / pred_stds_out / cov_out / the aux tables
 * named at begin.  Both scans and every output pointer must stay valid -- the scans unmodified -- until icet_solve_end returns.  Host work placed between the two -- the reference's constructor deep-copies both scans into its members
 * (src/icet.cpp:30,33); include/icet.h makes those copies there -- overlaps with the device.  One begin per context at a time; the
 * other host-pointer entry points refuse to run in between (ICET_ERR_BAD_ARG). */
icet_status icet_solve_begin(icet_ctx* ctx, const icet_params* p,
                             const float* scan1, int64_t n1, int64_t ld1,
                             const float* scan2, int64_t n2, int64_t ld2,
                             const float x0[6], float x_out[6], float pred_stds_out[6], float cov_out[36],
                             icet_aux* aux_or_null);
/* Optional, between begin and end: returns as soon as the KEYFRAME tables named at begin (cluster_bounds, has_fit, mu1, sigma1, evecs1,
 * l_diag, test_points -- everything fitScan1 produces, src/icet.cpp:68-252) are in the caller's arrays, while the Gauss-Newton loop is
 * still iterating on the device: the caller can convert them (include/icet.h builds its Eigen members and std::maps) under the loop. */
icet_status icet_solve_keyframe_tables(icet_ctx* ctx);
icet_status icet_solve_end(icet_ctx* ctx);

/* --- N independent pairs, HOST pointers (one ICET object per pair in the reference) --------------- */
icet_status icet_solve_batch(icet_ctx* ctx, const icet_params* p, int32_t n_pairs,
                             const float* const* scan1, const int64_t* n1,
                             const float* const* scan2, const int64_t* n2,
                             const float* x0 /* n_pairs x 6 or NULL = zeros */,
                             float* x_out /* n_pairs x 6 */, float* pred_stds_out /* n_pairs x 6 */,
                             float* cov_out /* n_pairs x 36 or NULL */);

/* --- N independent pairs, inputs and outputs resident in HBM; asynchronous on the ctx stream ------
 * d_x0: device n_pairs x 6 or NULL (zeros).  d_out: device n_pairs x 48 floats per pair:
 * [0,6) X, [6,12) pred_stds, [12,48) cov row-major.  Returns after enqueueing; call icet_sync. */
icet_status icet_solve_batch_device(icet_ctx* ctx, const icet_params* p, int32_t n_pairs,
                                    const icet_dev_scan* scan1, const icet_dev_scan* scan2,
                                    const float* d_x0, float* d_out);
/* Waits until everything this context has enqueued is done.  Behind exactly one small icet_solve_batch_device call (<= 8 pairs: a replayed graph) it watches a word of
 * pinned memory that the solve's last kernel raises behind its results instead of synchronising the stream (10 us sooner on this part); otherwise hipStreamSynchronize. */
icet_status icet_sync(icet_ctx* ctx);

/* --- the same solve in two halves, for the sequential callers (src/odometry.cpp:73-88: scan 2 of one frame is scan 1 of the next) ---
 * icet_keyframe_device builds the keyframe of n scans -- ICET::fitScan1, src/icet.cpp:68-107 -- and parks it in the context;
 * icet_register_device runs prepScan2 + runlen x fitScan2 (:254-277, :372-436) of n scan-2s against it, with the results of
 * icet_solve_batch_device bit for bit.  Both are asynchronous on the context's stream; a parked keyframe stays valid (and can be
 * registered against again) until the context solves or parks another.  With two contexts a caller builds the keyframe of frame
 * k on one stream while frame k-1 / k still iterates on the other (include/icet_nodes.h does). */
icet_status icet_keyframe_device(icet_ctx* ctx, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan1);
icet_status icet_register_device(icet_ctx* ctx, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan2,
                                 const float* d_x0, float* d_out);

/* The same two halves for scans whose row counts only the DEVICE knows yet (a range filter that compacted them a moment ago on the same
 * stream, src/odometry.cpp:57-70): scan[k].n is an UPPER BOUND (the launch geometry is sized from it), d_rows[k] (device, int32, may be NULL =
 * the bounds are the counts) the actual number of rows, read by the kernels when they run.  The caller need not wait for the filter. */
icet_status icet_keyframe_device_n(icet_ctx* ctx, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan1, const int32_t* d_rows);
icet_status icet_register_device_n(icet_ctx* ctx, const icet_params* p, int32_t n_pairs, const icet_dev_scan* scan2, const int32_t* d_rows,
                                   const float* d_x0, float* d_out);

/* Pre-size the workspace (so the first timed call does not allocate). */
icet_status icet_reserve(icet_ctx* ctx, const icet_params* p, int32_t n_pairs, int64_t total_n1, int64_t total_n2);

/* Wall-clock-free device timing of the most recent icet_solve_batch_device call, measured with HIP
 * events on the context's stream: [0] keyframe build ms, [1] Gauss-Newton loop ms, [2] ms inside the
 * bin/accumulate kernel only (sum over iterations; -1 unless ICET_FLAG_TIMING was set), [3] number of
 * accumulate launches timed. */
icet_status icet_last_timing(icet_ctx* ctx, float out_ms[4]);
/* The same call's point-pass launches one by one (ICET_FLAG_TIMING): acc_ms[it] = HIP-event time of iteration it's launch, *n_out = how many were written (<= cap). */
icet_status icet_last_timing_iters(icet_ctx* ctx, float* acc_ms, int32_t cap, int32_t* n_out);
/* The keep list of the point pass after the most recent throughput batch (option "keep"): out[4 k ..] = { mode at the end, groups of 4 points in pair k's last list,
 * passes that walked a list, lists built }.  ICET_ERR_BAD_ARG when the last call did not use the keep list (small batch, option off). */
icet_status icet_keep_stats(icet_ctx* ctx, int32_t n_pairs, int32_t* out);

/* Diagnostic hook for the parity tests: copy an internal per-point array of the scan-1 (keyframe) build
 * of the most recent call to the host.  `what`: 0 = float32 r of scan 1 in input order
 * (utils::cartesianToSpherical, src/utils.cpp:99,116); 1 = uint16 per row: bits 0-13 the row's voxel
 * bins_theta * binPhi + binTheta (sortSphericalCoordinates, src/icet.cpp:545-549), bit 14 "classified with
 * the literal formulas", bit 15 "r is exactly 0" (theta / phi themselves are not materialised: only decisions and the
 * Gaussians need them); 3 = int32 src[v], the original row that
 * sits at position v after the reference's sort + swap loop (src/icet.cpp:72-83); 4 = int32 per-pair
 * flags (bit 0: the bounded parallel walk overflowed and the serial replay was used); 6 = ONE int32: 1 if this device passed the
 * context's LDS-atomic order self-test (option "lds_rank").  `count` elements
 * from the start of the batch's concatenated scan-1 arrays (pairs for what = 4). */
icet_status icet_debug_fetch(icet_ctx* ctx, int32_t what, void* out, int64_t count);

/* Diagnostic hook for the parity tests: the 6x6 tail of one Gauss-Newton iteration (noise_mat = pinv(HTWH), pred_stds, checkCondition, dx:
 * src/icet.cpp:410-433) evaluated on the device for n host-side matrices, through the same device function the solve kernel runs.
 * htwh: n x 36 row-major, htwdz: n x 6; out: n x 56 = cov[36] | pred_stds[6] | dx[6] | eigenvalues[6] (NaN on the Cholesky route) |
 * pruned axes | route (0 Cholesky inverse, 2 literal restatement; option "gn_cond_bound"). */
icet_status icet_debug_gn_tail(icet_ctx* ctx, const float* htwh, const float* htwdz, int32_t n, float* out);
/* Diagnostic hook for the parity tests: the per-voxel weight W = pinv(.) of ICET_FLAG_REFERENCE_W (Eigen's float CompleteOrthogonalDecomposition, src/icet.cpp:320-321) for n
 * host-side 3 x 3 matrices (row-major, n x 9 in, n x 9 out), through the device function the solve kernel runs. */
icet_status icet_debug_pinv3(icet_ctx* ctx, const float* a, int32_t n, float* out);

/* Launch-shape and diagnostic knobs of ONE context (the library never reads the environment).  Defaults are the measured
 * optima.  Launch-shape knobs yield the same result bits; "force_exact", "guard_scale" and "lut_polar_quantile" preserve every
 * DECISION (the per-voxel counts n2_raw / n2_in) but move points between the 4-point runs and the runs of one, i.e. they regroup
 * float partial sums: X agrees to rounding, not bitwise.  Names: "lds_slots", "acc_pts",
 * "acc_blocks", "kf_pts", "rs_cap", "rs_max_cell" (0: per-bucket radix sort instead of the counting sort),
 * "keep" (the KEEP LIST of the point pass, batches of >= 32 pairs: H^T W H only sees scan-2 points in the angular bin of a voxel that has a scan-1 Gaussian -- src/icet.cpp:290-302 --
 * so a full pass also marks, per aligned group of 4 points, whether any of them is within an angular margin of such a bin, and later passes walk the list of marked
 * groups instead of the scan for as long as X stays within "keep_budget_t" (m, default 0.08) / "keep_budget_r" (Frobenius norm of the rotation difference, default 0.008)
 * of the X the marks were made at; a pair that leaves the budgets walks its whole scan once more and gets a new list.  The list pass forms the float partial sums of the
 * full pass exactly: SAME BITS with the option on, off, or with any budgets.  -1 / 1 on (default), 0 off; "keep_from" (default 1): first iteration whose pass makes marks),
 * "graph" (device-resident batches of <= 8 pairs: when a call's launch geometry and pointers equal the previous call's, the whole
 * solve is captured into a hipGraph and replayed from then on -- one hipGraphLaunch instead of ~33 launches on the host; 0 = never, default -1 = on),
 * "lds_rank" (the keyframe's stable multi-splits take a row's rank among equal classes from the value its LDS atomic hands back: -1 / 1 if
 * the device passed the order self-test run at icet_create, 0 = one ballot per class-id bit; same bits either way),
 * "exec_bits_lds" (0: swap-loop bit table read from memory), "exec_pairwise" (which kernel computes the swap loop's executed-step bits:
 * 1 one block per pair from the recurrence, 0 chain walks over independent tiles, -1 by batch size), "fuse_solve" (batches below 32 pairs on grids up to 4096 voxels:
 * 1 the block of a pair that finishes its share of an iteration's point pass last runs the pair's solve in the same launch -- 7 launches less per solve, same
 * bits; default 0: measured no faster on MI355X, where a device-scope fence is an L2 write-back), "batch_parts" (0 = automatic), "batch_stage" (0..4), "force_exact"
 * (every scan-2 point through the literal classification), "library_sort" (rocPRIM radix sort instead of
 * the hand-written rank sort: only in a diagnostic build, `make EXTRA=-DICET_DIAG_LIBSORT`; the shipped library answers ICET_ERR_UNSUPPORTED), "guard_scale" (>= 1), "lut_polar_quantile" (0..1), "gn_cond_bound" (0 .. 1e6, default 2.5e5 -- a factor 4 below checkCondition's cutoff, because a float Cholesky inverse knows its own norm to a few per cent only at such condition numbers: an H^T W H whose Frobenius bound on
 * the condition number |A|_F |A^-1|_F exceeds it is inverted by the literal restatement of the reference's statements -- column-pivoted QR
 * pseudo-inverse, eigenvectors, pruning -- instead of a Cholesky factorisation; 0 = always literal.  Not a launch-shape knob: between the two
 * routes cov / dx differ by rounding times the condition number).  Unknown name or value
 * out of range: ICET_ERR_BAD_ARG. */
icet_status icet_set_option(icet_ctx* ctx, const char* name, double value);

/* The HIP stream (hipStream_t as void*) and device a context enqueues on -- for callers that produce scans on the GPU. */
void* icet_stream(icet_ctx* ctx);
int   icet_device(const icet_ctx* ctx);

/* --- the batched-pairs case over several GPUs of one node (BASELINE.json configs[3]) ---------
 * The reference has no counterpart: it constructs one ICET object per pair on the calling thread.  A handle owns one context and
 * one persistent host thread per device; pair k of a call runs on device_ids[k mod n_devices], and the 48 result floats per pair
 * are gathered into ONE buffer -- the caller's host arrays, or HBM of device_ids[0].  There is no data-path collective.  The
 * device-resident gather is, by option "gather": 0 (default) one strided peer copy per device over xGMI (peer access between
 * device_ids[0] and the others is enabled at create), or 1 ONE ncclAllGather over a communicator of the handle's devices (RCCL,
 * dlopen'ed on first use; needs distinct device ids) followed by the de-interleave on device_ids[0].  A process that runs one
 * rank per GPU gathers with torch.distributed / RCCL instead (icet_amd/dist.py). */
typedef struct icet_multi icet_multi;
icet_status icet_multi_create(icet_multi** handle, const int32_t* device_ids, int32_t n_devices);   /* each id < device count; an id may repeat (one context per ENTRY) */
icet_status icet_multi_destroy(icet_multi* handle);
const char* icet_multi_last_error(const icet_multi* handle);
int32_t     icet_multi_devices(const icet_multi* handle);
icet_ctx*   icet_multi_context(icet_multi* handle, int32_t i);      /* the context of device_ids[i] (e.g. for icet_reserve) */
/* "gather" (above), or any icet_set_option name, applied to every device's context. */
icet_status icet_multi_set_option(icet_multi* handle, const char* name, double value);
/* HOST pointers in and out, same meaning as icet_solve_batch. */
icet_status icet_multi_solve_batch(icet_multi* handle, const icet_params* p, int32_t n_pairs,
                                   const float* const* scan1, const int64_t* n1, const float* const* scan2, const int64_t* n2,
                                   const float* x0, float* x_out, float* pred_stds_out, float* cov_out);
/* Scans resident in HBM: scan1[k] / scan2[k] on device_ids[k mod n_devices]; d_x0 (n_pairs x 6 or NULL) and d_out (n_pairs x 48,
 * layout of icet_solve_batch_device) on device_ids[0].  Returns after the gather has completed (synchronous).  The devices'
 * streams are the handle's own: the scans, d_x0 and whatever last used d_out must have COMPLETED before the call -- or, for work
 * queued on one stream of device_ids[0], use the _after form: an event recorded on `producer_stream` (hipStream_t as void*, NULL =
 * none) when the call starts is waited for by every device's stream. */
icet_status icet_multi_solve_batch_device(icet_multi* handle, const icet_params* p, int32_t n_pairs,
                                          const icet_dev_scan* scan1, const icet_dev_scan* scan2, const float* d_x0, float* d_out);
icet_status icet_multi_solve_batch_device_after(icet_multi* handle, const icet_params* p, int32_t n_pairs,
                                                const icet_dev_scan* scan1, const icet_dev_scan* scan2, const float* d_x0, float* d_out,
                                                void* producer_stream);

/* The asynchronous form (review r3: the synchronous entry cost one process ~9 % against the stream-ordered single-context path, and eight
 * GPUs driven from one thread would start behind).  _async hands every device's share to its host thread and returns at once -- the
 * caller's descriptor arrays are copied, nothing waits for a device; calls queue up behind each other in order.  icet_multi_sync waits
 * until everything queued so far has completed on every device (d_out is then valid) and returns the first failure since the last sync.
 * producer_stream as in the _after form (NULL = none). */
icet_status icet_multi_solve_batch_device_async(icet_multi* handle, const icet_params* p, int32_t n_pairs,
                                                const icet_dev_scan* scan1, const icet_dev_scan* scan2, const float* d_x0, float* d_out,
                                                void* producer_stream);
icet_status icet_multi_sync(icet_multi* handle);

#ifdef __cplusplus
}
#endif
#endif /* ICET_HIP_H */
