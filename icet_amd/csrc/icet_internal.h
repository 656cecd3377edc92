// icet_amd/csrc/icet_internal.h -- structures shared by the HIP kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icet {

// Per-slot accumulator: raw count (u32), in-bounds count (u32), then 9 sums Sd[3], Sdd[6] as 64-bit FIXED POINT
// (value * 2^36, two's complement).  Integer atomics make the sums independent of arrival order, so results are
// bitwise reproducible run to run and identical between batched and single solves; 2^-36 m^2 resolution (1.5e-11) is far
// below float32 rounding of the addends.
constexpr int kAccWords = 20;        // in HBM: AoS, 80 bytes per slot (8-byte aligned)
constexpr int kAccLds   = 20;        // in LDS: the same 80-byte record per slot ([raw | in << 32], 9 x i64)
// Why 2^36 and not coarser: a voxel holding ~27 points of ONE lidar ring is a line whose covariance has a smallest eigenvalue of
// ~5e-8 m^2, i.e. a scatter of 1.4e-6 m^2 in that direction, and it enters H^T W H with a weight of 1 / lambda_min; at 2^-30 the
// truncation of ~15 flushes biased that scatter by 0.5 % (round 2, scripts/diag_voxel.py 61).  A flushed value is the partial sum
// of at most 4 points; values of 2^15 m^2 and more (coarse grids x long ranges) take the wide conversion (icet_device_common.h, to_fix); the totals have 2^27 m^2 of headroom.
constexpr float kFixScale = 68719476736.0f;           // 2^36
constexpr double kFixInv = 1.0 / 68719476736.0;
#ifndef ICET_RS_BUCKET_BITS
#define ICET_RS_BUCKET_BITS 8
#endif
constexpr int kRankSortBucketBits = ICET_RS_BUCKET_BITS;       // rank sort (icet_ranksort.hip): at most 2^bits buckets per pair, ids travel as u8
constexpr int kRankSortMaxBuckets = 1 << kRankSortBucketBits;
#ifndef ICET_KF_MAXPTS
#define ICET_KF_MAXPTS 8
#endif
constexpr int kKfMaxPtsPerThread = ICET_KF_MAXPTS;   // keyframe kernels: largest tile = 256 threads x this many rows
// Largest grid the kernels accept.  Binding constraint: k_bin_scatter keeps 4 x V running offsets (16 B per voxel) in one
// block's LDS (160 KB per CU) -- validated up front in ensure_workspace so that a too-fine grid is refused with
// ICET_ERR_UNSUPPORTED instead of failing a launch mid-sequence.  Voxel ids travel in 14 bits of a 16-bit word whose two top
// bits carry per-row flags (kRowNearBit, kRowZeroBit); slot ids travel as int16.
constexpr int kMaxVoxels = 10000;
constexpr int kGnPartWords = 4 * 20 * 27; // Workspace::gn_part (icet_solve.hip: at most 4 pairs x 20 virtual blocks of 512 slots x 27 partial sums)
constexpr uint16_t kRowZeroBit = 0x8000u;   // the row's r is exactly 0: the invalid returns of a real scan (thousands of exact-zero rows, all in ONE voxel) -- their r need not be gathered
constexpr uint16_t kRowNearBit = 0x4000u;   // the row lies within a guard band of a voxel edge: its azimuth / polar bounds test must be done with the literal formulas
constexpr uint16_t kRowBinMask = 0x3FFFu;
constexpr uint32_t kSortedNearBit = 0x80000000u;   // the same flag in the sorted-row table (rows < 2^30)
constexpr uint32_t kSortedZeroBit = 0x40000000u;   // kRowZeroBit in the sorted-row table
constexpr uint32_t kSortedRowMask = ~(kSortedNearBit | kSortedZeroBit);

// One scan pair as the kernels see it (device pointers, column-major N x 3).
struct PairDesc {
    const float* s1; const float* s2;
    int32_t n1, ld1, n2, ld2;
    int32_t off1;                    // start of this pair's segment in the scan-1 temporaries
    int32_t off2;                    // start of this pair's segment in the scan-2 overflow list (near_over)
};

// What the per-iteration point kernel needs about an active voxel (keyframe voxel with a scan-1
// Gaussian, outer > 1 and |idx1| > n: the scan-1 side of the gate at src/icet.cpp:290).
struct SlotHot {
    float az0, az1, el0, el1, inner, outer;   // clusterBounds row (src/icet.cpp:149)
    float mu[3];                              // mu1
    int32_t v;                                // voxel index T*phi + theta
    int32_t pad[2];
};

// What the per-voxel Gauss-Newton kernel needs.
struct SlotFit {
    float mu[3];
    float s1n[6];                             // sigma1 / (|idx1| - 1), upper triangle xx xy xz yy yz zz
    float M[9];                               // L * U^T = L * V, row-major (src/icet.cpp:317,329)
    int32_t n1;
    int32_t v;
};

// What k_fit_scan1 (one wave per angular bin) hands to k_fit_finish (one lane per bin): the Gaussian of the bin's cluster.
struct FitMid {
    float mean[3];
    float cov[6];                             // xx xy xz yy yz zz
    float inner, outer;                       // findCluster's bounds (0, 0 when there is no cluster)
    int32_t cnt;                              // rows in the bin
    int32_t has_fit;
    int32_t pad[3];                           // pad[0]: rows inside the radial range (k_fit_cluster -> k_fit_moments)
};
static_assert(sizeof(FitMid) == 64, "k_fit_scan1 stages this record as 16 words");

static_assert(sizeof(SlotHot) == 48 && sizeof(SlotFit) == 80, "k_compact_slots copies these records as 12 + 20 words");

// The KEEP LIST of the point pass (round 6).  H^T W H only ever sees scan-2 points that fall into the angular bin of an ACTIVE voxel
// (raw count, src/icet.cpp:290,315) or inside its bounds (sums, :299-306); on lidar data 40-60 % of scan 2 lies in bins without a scan-1
// Gaussian and pays the full classification in every iteration for nothing.  A full pass at X_ref therefore also writes, per aligned group
// of 4 points, whether ANY of them lies within an angular margin of an active bin (one 64-bit ballot per 256 points: keep_mask); the solve
// behind it expands the masks into the ordered list of kept groups (keep_list), and the following passes walk the list instead of the scan
// for as long as X stays within the budgets the margin was computed for (|t - t_ref| <= budget_t, |R - R_ref|_F <= budget_r: then no dropped
// point can have reached an active bin, see keep_margin in icet_accumulate.hip); a pair that leaves them falls back to the full scan for
// one pass, which rebuilds masks and list.  The list pass forms EXACTLY the float partial sums of the full pass (a kept group is an aligned
// group of the original stream; a suffix run is handed to the next lane only where the full pass would hand it on), so the result bits do not
// change: the keep list is a launch-shape option (tests/test_gpu_parity.py::test_keep_list_is_bit_neutral).
struct KeepState {
    int32_t mode;                    // 0: the next point pass walks the whole scan (and writes keep masks); 1: it walks keep_list
    int32_t n_keep;                  // groups in the list
    int32_t list_passes, builds;     // statistics of the current solve: passes that walked the list / lists built
    float ref[12];                   // t[3] | R[9] of the pass the masks were written in
};
static_assert(sizeof(KeepState) == 64, "one cache line quarter per pair");
// modes_next: what the point pass of the NEXT iteration does with each pair, one word per pair: 0 = the whole scan, n_keep + 1 = the list.  Its blocks take the
// pairs in the order "whole-scan pairs ascending, then list pairs descending" (keep_pair_of_slot, icet_accumulate.hip): a whole-scan block runs ~2x as long as a
// list block and a launch is only two rounds of blocks -- started in the second round, ONE such block set the end of the launch (measured: 105-115 us per launch
// whatever the lists' length, against 97 for the plain pass).
struct KeepArgs { const PairDesc* desc; const unsigned long long* mask; uint32_t* list; KeepState* state; float bt2, br2; int on; int32_t* modes_next; int n_pairs; };

// Optional dense (per-voxel) dump for the reference's public side tables; device pointers or null.
struct AuxDev {
    float* bounds; int32_t* n1_raw; int32_t* has_fit; float* mu1; float* sigma1; float* evecs1; float* l_diag;
    float* x_hist; float* htwh; float* htwdz; float* cond; int32_t* n2_raw; int32_t* n2_in; float* test_points;
    const int32_t* pair_user;   // LaunchCfg::pair_user (set by launch_gn_solve, not by callers)
    int32_t* done_flag;         // LaunchCfg::done_flag, in the last iteration's launch only (set by launch_gn_solve)
    float* xf_last;      // 48 floats: the transform record the LAST iteration's point pass uses (written by k_init_state / the solve of iteration runlen - 2), for `points2`
};

struct Workspace {
    // capacities
    int32_t cap_pairs = 0; int64_t cap_n1 = 0; int32_t cap_V = 0;
    PairDesc* desc = nullptr;
    PairDesc* desc_rt = nullptr; float* rt2 = nullptr; int64_t cap_rt2 = 0;   // ICET_FLAG_ROUNDTRIP_SCAN2: descriptors whose scan 2 is the round-tripped copy, and that copy (3 x cap_rt2 floats)
    int32_t* seg_off = nullptr;               // pairs+1, scan-1 segment offsets
    float *r1 = nullptr;                                             // radial distance of every scan-1 row, input order (bit-exact c2s r)
    float *cart1 = nullptr;                                          // 3 x cap_n1: round-tripped Cartesian rows of large bins (k_fit_scan1's second pass)
    unsigned long long *key64A = nullptr, *key64B = nullptr;         // (pair << 32 | r bits)
    uint32_t *keyA = nullptr, *keyB = nullptr, *valA = nullptr, *valB = nullptr;
    uint32_t* splitters = nullptr; int32_t* n_buckets = nullptr; int32_t* bucket_start = nullptr;   // rank sort: pairs x kRankSortMaxBuckets, pairs, pairs x (kRankSortMaxBuckets + 1)
    uint8_t* bkt = nullptr;                                          // rank sort: bucket id of every scan-1 row
    uint16_t* binpos = nullptr;                                      // angular bin of the row at every position after the scramble
    uint32_t* counts = nullptr; uint32_t* tile_base = nullptr; size_t cap_counts = 0;               // pairs x tiles x V histogram / tile base offsets
    uint16_t* bin16 = nullptr;                                       // angular bin of every scan-1 row (input order)
    unsigned long long* execbits = nullptr;                          // swap loop: "step v executed", 64 rows per word (exec_word_base, icet_keyframe.hip)
    int32_t *pred = nullptr, *src = nullptr;
    int32_t *bin_count = nullptr, *bin_start = nullptr;             // pairs x V, pairs x (V+1)
    SlotHot* hotD = nullptr; SlotFit* fitD = nullptr; int32_t* activeD = nullptr;   // dense, pairs x V
    FitMid* midD = nullptr;                                                          // dense, pairs x V
    int32_t* live_bins = nullptr; int32_t* n_live = nullptr;                         // pairs x V records of 4 words {bin, first row, rows, candidates}: bins holding >= n rows, compacted; pairs: their number
    void* fit_items = nullptr; size_t cap_fit_items = 0; uint32_t* fit_n_items = nullptr;   // work items of k_fit_roundtrip (16 B each) and their count
    SlotHot* hotS = nullptr; SlotFit* fitS = nullptr;                               // compact, pairs x V
    int16_t* slot_of_voxel = nullptr; int32_t* n_slots = nullptr;
    uint32_t* acc = nullptr;                  // pairs x V x kAccWords
    // Scan-2 points that the fast classification cannot decide and that did not fit the block's LDS queue (k_gn_accumulate):
    // point indices, one segment of n2 entries per pair, and the fill count per pair.  Drained -- and the count reset -- by
    // k_gn_solve.  Empty on ordinary data (~0.02 % of the points are undecided and a block's queue holds 512 of them).
    uint32_t* near_over = nullptr; int64_t cap_n2 = 0; uint32_t* near_over_count = nullptr;   // (2 x cap_pairs words: the second half are the point pass' per-pair block tickets, gn_done())
    uint32_t* gn_done() const { return near_over_count ? near_over_count + cap_pairs : nullptr; }
    float* xf = nullptr;                      // pairs x 48: t[3], R[9] row-major, angles[3], pad, J[27] (see write_xf)
    float* gn_part = nullptr;                 // two-stage solve of fine grids: kGnPartWords floats of partial sums (icet_solve.hip)
    float* X = nullptr;                       // pairs x 6
    int32_t* flags = nullptr;                 // pairs: bit0 = scramble walk overflow
    int32_t* zero_rows = nullptr;             // pairs x 4: exact-zero rows of scan 1 by the sign pattern of (y, x) -- the invalid returns of a real scan, thousands in ONE voxel (k_scan1_spherical counts, k_fit_cluster skips a bin that holds nothing else)
    int32_t* vrange = nullptr;                // pairs x 2: smallest / largest voxel id any scan-1 row of the pair has: the voxel multi-split's tables are touched inside it only
    int32_t* tile_vr = nullptr; size_t cap_tile_vr = 0;   // pairs x tiles x 2: the same per tile (k_scan1_spherical), reduced per pair by the rank sort's k_bin_scan
    float* thr = nullptr; int thr_T = 0, thr_P = 0;   // bin-edge tables: T+1 azimuth thresholds, then P+1 polar thresholds
    void* lut = nullptr; int lut_Mt = 0, lut_Mp = 0;  // classification LUTs of k_gn_accumulate: Mt azimuth cells, then Mp polar cells (8 B each)
    float guard_t = 0.f, guard_p = 0.f;               // guard bands (diamond-angle / cosine units) around voxel edges
    int zero_voxel[4] = {0, 0, 0, 0};                 // voxel of an exact-zero row by the sign bits of (y, x) (ensure_thresholds)
    // keep list of the point pass (KeepState above): one mask word per 256 scan-2 points (pair's first word: off2 / 256 + pair), one list entry per group of
    // 4 points (off2 / 4 + pair), the per-pair state; edges: T + 1 azimuth bin edges in diamond-angle units, then P + 1 polar edges in w = -cos(phi) units
    unsigned long long* keep_mask = nullptr; size_t cap_keep_mask = 0; uint32_t* keep_list = nullptr; size_t cap_keep_list = 0; KeepState* keep_state = nullptr; int32_t cap_keep_pairs = 0;
    int32_t* keep_modes = nullptr;            // 2 x cap_keep_pairs: per pair what the point pass of an even / odd iteration does with it (KeepArgs::modes_next)
    float* edges = nullptr;
    void* sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
};

// Launch-shape and diagnostic knobs.  Defaults are the measured optima; every value gives the same bits
// (tests/test_gpu_parity.py::test_rarely_taken_paths_give_the_same_bits).  Set per context through icet_set_option -- the
// library never reads the environment.
#ifndef ICET_FUSE_DEFAULT
#define ICET_FUSE_DEFAULT 0       /* measured: no gain on MI355X (LABNOTES round 5); 1 / -1: small batches run the solve inside the point pass' launch */
#endif
struct Tuning {
    int lds_slots = 0;            // active-voxel rows kept in LDS by k_gn_accumulate (0 = sized from the LDS budget)
    int acc_pts = 4;              // k_gn_accumulate: minimum points per thread per block
    int acc_blocks = 1536;        // k_gn_accumulate: target blocks per launch
    int force_exact = 0;          // route every scan-2 point through the literal classification
    int library_sort = 0;         // rocPRIM radix sort instead of the hand-written rank sort
    int kf_pts = kKfMaxPtsPerThread;   // keyframe tile = 256 threads x this many rows
    int batch_parts = 0;          // parts a device batch is cut into (0 = automatic)
    int batch_stage = 4;          // keyframe stage after which the next part may start (0 = lock step)
    int rs_cap = 0;               // LDS rows of the per-bucket sort (0 = from the largest scan)
    int rs_max_cell = 24;         // per-bucket counting sort: a cell above this many rows sends the bucket to the radix sort (0: always)
    int exec_bits_lds = 1;        // k_scramble_src keeps the pair's swap-loop bit table in LDS (0: reads it from memory, the path of scans above ~0.75 M rows)
    int lds_rank = -1;            // the stable multi-splits take a row's rank from the value its LDS atomic hands back (1 / -1: if this device passed lds_rank_selftest; 0: ballots per id bit)
    int fuse_solve = ICET_FUSE_DEFAULT;   // small batches: the pair's last point-pass block runs the solve in the same launch (1 / -1), or one k_gn_solve launch per iteration (0)
    int exec_pairwise = -1;       // "did step v execute": one block per pair in index order with the bit table in LDS (k_exec_flags_pair) 1, chain walks (k_exec_flags) 0, by batch size -1
    double guard_scale = 1.0;     // multiplies the classification guard bands (tables are rebuilt)
    double lut_polar_quantile = 0.25;   // polar LUT cell width = this quantile of the polar bin widths
    int keep = 0;                 // keep list of the point pass (KeepState): 1 throughput batches (>= 32 pairs), 0 never (default: measured a net loss on MI355X)
    int keep_from = 1;            // first iteration whose full pass writes keep masks (0 pays when X0 is already close: a seeded sequential caller)
    double keep_budget_t = 0.08;  // metres / radians (Frobenius) X may move from the pass the masks were written in before the pair falls back to the full scan
    double keep_budget_r = 0.008;
    double keep_check_scale = 1.0; // TIMING EXPERIMENTS ONLY: the budget CHECK (not the margins) times this; above 1 the results are wrong (a pair keeps its list beyond what the margins cover)
    double gn_cond_bound = 2.5e5;   // H^T W H whose Frobenius condition bound exceeds this takes the literal 6x6 tail (COD / eigenvectors / pruning) instead of the Cholesky inverse; 0 = always literal
};

struct LaunchCfg {
    int T, P, V, n, runlen;
    float thresh, buff;
    int n_pairs;
    int max_n1, max_n2;
    int64_t total_n1;
    int lds_slots = 0;   /* 0 = sized from the LDS budget in launch_gn_accumulate */              // active voxels kept in LDS by k_gn_accumulate (the rest go straight to HBM)
    int acc_min_pts_per_thread = 4;   // launch shaping of k_gn_accumulate: a block handles at least one trip (512 threads x 4 points); matters for single pairs
    int acc_target_blocks = 1536;     // 6 blocks per CU = two rounds of the three that are resident; each block's start-up (LUTs, map, hot records into LDS) is paid once per ~20 k points
    int kf_chunks = 1;                // tiles per pair in the keyframe kernels (set by the host from max_n1)
    int kf_pts_per_thread = kKfMaxPtsPerThread;        // keyframe kernels: points per thread (sets chunks per pair)
    hipEvent_t stage_event = nullptr; /* recorded inside launch_keyframe after stage `stage_at` (4 spherical, 1 sort, 2 scramble, 3 gather): lets the next batch part start there */
    int stage_at = 0;
    int use_library_sort = 0;         // diagnostic: rocPRIM device radix sort instead of the hand-written rank sort
    int vec4_ok = 0;                  // every scan-2 pointer and leading dimension is 16-byte aligned
    int true_sort = 0;                // ICET_FLAG_TRUE_SORT (non-parity extension): src[] = the sorted order itself
    int force_exact = 0;              // diagnostic: route every point through the literal evaluation
    int rs_cap = 0;                   // Tuning::rs_cap
    int rs_max_cell = 24, exec_bits_lds = 1, exec_pairwise = -1;   // Tuning::rs_max_cell, ::exec_bits_lds, ::exec_pairwise
    int fuse_solve = 0;               // Tuning::fuse_solve (-1 resolved to 1)
    int reject_moving = 0;            // ICET_FLAG_REJECT_MOVING (non-parity extension)
    int half_gap = 0;                 // ICET_FLAG_HALF_GAP_BOUNDS (non-parity extension; sets true_sort as well)
    int rt2 = 0;                      // ICET_FLAG_ROUNDTRIP_SCAN2 (parity-study option)
    int ref_w = 1;                    // per-voxel W by the reference's float COD (default); 0: ICET_FLAG_DOUBLE_W, in double with an eigenvalue rank rule
    int lds_rank = 0;                 // Tuning::lds_rank, resolved against the context's self-test
    int keep = 0, keep_from = 1;      // Tuning::keep resolved for this launch (batch size, flags), Tuning::keep_from
    float keep_bt = 0.f, keep_br = 0.f, keep_check_scale = 1.f;   // Tuning::keep_budget_t / _r / keep_check_scale
    float gn_cond_bound2 = 6.25e10f;     // Tuning::gn_cond_bound squared: k_gn_solve's Cholesky route needs |A|_F |A^-1|_F <= the bound (icet_solve.hip gn_tail)
    int32_t* done_flag = nullptr;          // set: the solve of the last iteration stores 1 there (pinned host memory) behind its results (icet_ctx_set_done_flag)
    const int32_t* pair_user = nullptr;    // set (ragged throughput batches, icet_capi.hip solve_device_part): slot s of the tables holds the caller's pair pair_user[s] -- X0 is read and the results are written there
    const PairDesc* h_desc_up = nullptr; const int32_t* h_seg_up = nullptr;      // set: the keyframe's first kernel (k_rs_splitters) copies the descriptors / segment offsets from this pinned staging itself (small batches: launch_upload_desc's job)
};
constexpr float kRejectMovingThresh = 0.3f;        // python/ICET_spherical.py:38  RM_thresh
constexpr int kRejectMovingStartIter = 4;          // python/ICET_spherical.py:36  start_RM_iter

// icet_keyframe.hip
hipError_t launch_keyframe(const Workspace& w, const LaunchCfg& c, const AuxDev* aux, hipStream_t st, const int32_t* d_n1 = nullptr);   // d_n1: scan-1 row counts known to the device only (the descriptors hold upper bounds)
// icet_solve.hip
hipError_t launch_init_state(const Workspace& w, const LaunchCfg& c, const float* d_x0, hipStream_t st, float* xf_last = nullptr, const int32_t* d_n2 = nullptr, const PairDesc* h_desc = nullptr, const int32_t* h_seg = nullptr);   // (also clears the block tickets and the keep-list state)
// keep_pass: the point pass in front of this solve was launched with keep_pass (below): build the list of every pair that walked its whole scan, check every pair's budgets
hipError_t launch_gn_solve(const Workspace& w, const LaunchCfg& c, int iter, float* d_out, const AuxDev* aux, hipStream_t st, int keep_pass = 0);
hipError_t launch_gn_tail_debug(const float* d_H, const float* d_g, float* d_out, int n, float bound2, hipStream_t st);     // test hook: the 6x6 tail on its own
hipError_t launch_pinv3_debug(const float* d_A, float* d_out, int n, hipStream_t st);                                        // test hook: the 3x3 float COD pseudo-inverse of ICET_FLAG_REFERENCE_W
// `points2` of pair 0 (include/icet.h:80): scan 2 under the transform record `xf` (AuxDev::xf_last); out = n2 x 3 column-major, ld n2 (may be pinned host memory)
hipError_t launch_points2(const Workspace& w, const LaunchCfg& c, const float* xf, float* out, hipStream_t st);
// icet_accumulate.hip
// fuse: when given and the launch qualifies (a small batch on a grid of <= 4096 voxels, no scan-2 round trip, option "fuse_solve" not 0), the block of each pair that
// finishes LAST runs that pair's solve of iteration fuse->iter inside the same launch (*fused = true: the caller skips launch_gn_solve)
struct FuseArgs { int iter; float* d_out; const AuxDev* aux; };
// keep_pass: the kernel with the keep list (a pair in list mode walks its list, any other pair its whole scan and writes keep masks), 1 + 2 x (iteration & 1); 0: the plain kernel
hipError_t launch_gn_accumulate(const Workspace& w, const LaunchCfg& c, hipStream_t st, const FuseArgs* fuse = nullptr, bool* fused = nullptr, int keep_pass = 0);
size_t acc_fixed_lds_bytes(int T, int P, int Mt, int Mp, bool small_batch, bool keep = false);   // LDS of a k_gn_accumulate block without its slot rows
size_t acc_row_lds_bytes();
// ICET_FLAG_ROUNDTRIP_SCAN2: points2_OG = sphericalToCartesian(cartesianToSpherical(scan 2)) (src/icet.cpp:263-275, without the permutation): w.desc -> w.desc_rt
hipError_t launch_rt2_prepare(const Workspace& w, const LaunchCfg& c, hipStream_t st);
hipError_t launch_patch_counts(const Workspace& w, const LaunchCfg& c, const int32_t* d_n1, const int32_t* d_n2, hipStream_t st);   // icet_solve.hip
// Descriptors and segment offsets of a SMALL batch from the pinned staging into the workspace by a kernel (round 6): a copy command of a few dozen bytes costs a stream
// -- and a replayed graph, as a memcpy node -- 10 to 60 us on this part before the next kernel starts; a kernel that reads the pinned words itself costs a launch.
constexpr int kUploadDescMaxPairs = 64;
hipError_t launch_upload_desc(const Workspace& w, const PairDesc* h_desc, const int32_t* h_seg, int n_pairs, hipStream_t st);   // icet_solve.hip
// icet_sidetables.hip: the per-point members of the reference object, on request (pair 0 of a single-pair solve)
hipError_t launch_side_scan1(const Workspace& w, const LaunchCfg& c, float* sph, int32_t* index, hipStream_t st);
hipError_t launch_side_scan2(const Workspace& w, const LaunchCfg& c, const float* xf, float* pts, float* sph, int32_t* voxel, hipStream_t st);
// Raise the dynamic-LDS limit of the kernels that need more than the default; called once per context (icet_create) with the
// context's device current -- no process-global "done" flags.
hipError_t init_keyframe_kernels();
// does this device hand back the pre-add values of one wave's LDS atomic in ascending lane order (what Tuning::lds_rank rests on)?  Synchronises st.
hipError_t lds_rank_selftest(int32_t* d_scratch, hipStream_t st, int* ok);
hipError_t init_accumulate_kernels();
hipError_t init_rank_sort_kernels();

// ranksort.hip
hipError_t launch_rank_sort_splitters(const Workspace& w, const LaunchCfg& c, hipStream_t st, const int32_t* d_n1 = nullptr);   // before k_scan1_spherical; d_n1: device-side row counts to patch into the descriptors
hipError_t launch_rank_sort(const Workspace& w, const LaunchCfg& c, hipStream_t st);             // after it (buckets + histogram done there)

// bucket(key) = number of splitters strictly below key (splitters sorted ascending in sp[1 .. kRankSortMaxBuckets-1], sp[0] ignored)
__device__ __forceinline__ int rank_sort_bucket_of(uint32_t key, const uint32_t* sp) {
    int lo = 0;
#pragma unroll
    for (int step = kRankSortMaxBuckets / 2; step > 0; step >>= 1) lo += (sp[lo + step] < key) ? step : 0;
    return lo;
}
hipError_t launch_class_scan(const uint32_t* counts, uint32_t* tile_base, int32_t* class_start, int n_classes, int chunks, int n_pairs, hipStream_t st,
                             int32_t* live = nullptr, int32_t* n_live = nullptr, int live_min = 0, uint32_t* n_items = nullptr, const int32_t* class_range = nullptr,
                             const int32_t* tile_vr = nullptr, int32_t* vrange_out = nullptr);

// sort.hip
size_t sort_temp_bytes(int64_t total_n);
hipError_t sort_pairs_u64(void* tmp, size_t tmp_bytes, const unsigned long long* key_in, unsigned long long* key_out,
                          const uint32_t* val_in, uint32_t* val_out, int64_t total_n, int end_bit, hipStream_t st);
hipError_t sort_pairs_u32(void* tmp, size_t tmp_bytes, const uint32_t* key_in, uint32_t* key_out,
                          const uint32_t* val_in, uint32_t* val_out, int64_t total_n, int end_bit, hipStream_t st);

}  // namespace icet

// nodes: point a context at another stream for the calls that follow (the caller restores it; a captured graph replays into whatever stream is set)
struct icet_ctx;
void icet_ctx_set_stream(icet_ctx* c, hipStream_t s);
