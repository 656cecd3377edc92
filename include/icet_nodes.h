/* include/icet_nodes.h -- C ABI of the callers on either side of the hot path,
 * kept on the MI355X so that a scan stream never leaves HBM between frames.
 *
 * What it restates (the ROS plumbing itself -- subscribers, publishers, tf -- is out of scope):
 *   * the per-frame body of `OdometryNode::pointCloudCallback`   /root/reference/src/odometry.cpp:46-98
 *   * the per-frame body of `MapMakerNode::pointCloudCallback`   /root/reference/src/simpleMapMaker.cpp:86-172
 *   * `EigenQueue` (the HD-map FIFO)                             /root/reference/src/simpleMapMaker.cpp:18-59
 *   * the per-frame body of `ScanRegistrationNode::pointcloudCallback` /root/reference/src/scanMatcher.cpp:30-110
 * i.e. first scan stored as-is, every later scan range-filtered (row norm > min_range), ICET(prev, cur, ...) solved
 * through icet_hip.h, X0 seeded for the next frame, the divergence guard, the accumulated pose X_homo, and the map
 * queue re-expressed in the new sensor frame.  Scan layout as in icet_hip.h: N x 3 float32 column-major, ld >= N.
 *
 * Quirks kept (do not "fix"): the FIRST scan is not range-filtered (odometry.cpp:46-52); the guard zeroes X but the
 * pose is still chained with the zeroed X (simpleMapMaker.cpp:129-172); the map transform is applied to all
 * `map_capacity` rows, filled or not (simpleMapMaker.cpp:40); the down-sample is the head of a std::shuffle driven by a
 * default-seeded std::mt19937 that lives as long as the node (simpleMapMaker.cpp:147-158,257-258).
 * One deviation: the reference copies `map_downsample` rows even when the scan has fewer (reads past the shuffled
 * index vector, simpleMapMaker.cpp:155-157); here min(map_downsample, rows) rows are taken.
 */
#ifndef ICET_NODES_H
#define ICET_NODES_H
#include "icet_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct icet_node_params {
    icet_params solve;        /* odometry: {7, 24, 75, 25, .1, .1} (odometry.cpp:73-76); map maker: runlen 12 (simpleMapMaker.cpp:113-119) */
    float   min_range;        /* keep rows with norm > min_range: 2.0 (odometry.cpp:58) / 0.2 (simpleMapMaker.cpp:98)     */
    int32_t seed_x0;          /* 1: X0 <- X for the next frame (odometry.cpp:82); 0: X0 <- 0 (simpleMapMaker.cpp:124)       */
    float   trans_thresh;     /* divergence guard |X[0..2]| > trans_thresh ... (simpleMapMaker.cpp:129-137, 0.3 / 0.3);   */
    float   rot_thresh;       /*   both <= 0: no guard (the odometry node has none)                                        */
    int32_t map_capacity;     /* rows of the map FIFO: 600000 (simpleMapMaker.cpp:62); 0: no map (odometry node)           */
    int32_t map_downsample;   /* rows of each scan that enter the map: 2000 (simpleMapMaker.cpp:147)                        */
    int32_t flags;            /* ICET_NODE_* below; 0 for the two nodes above                                              */
} icet_node_params;

/* scan_registration_node (src/scanMatcher.cpp:30-110): no range filter at all (:44, every row -- NaN rows included -- goes
 * to the constructor), X0 = 0 every frame (:59-62), and two published clouds: scan 2 expressed in scan 1's frame,
 * `(pcl_matrix * rot_mat.inverse()).rowwise() - trans` (:76; note the order: rotate, THEN subtract -- the map queue does the
 * opposite), and the "snail trail" of past sensor positions, transformed the same way each frame with a new origin row
 * appended (:79-84). */
enum { ICET_NODE_NO_RANGE_FILTER = 1,   /* every row of every scan is kept (min_range ignored)                              */
       ICET_NODE_ALIGNED_CLOUD   = 2,   /* keep scan2_in_scan1_frame of the last frame on the device (icet_node_aligned)    */
       ICET_NODE_SNAIL_TRAIL     = 4,   /* maintain the snail trail (icet_node_snail_trail)                                 */
       ICET_NODE_NO_PIPELINE     = 8,   /* build every frame's keyframe inside its own solve, as the reference does, instead of one frame
                                           ahead on a second stream (same result bits either way; for A/B timing and the tests)          */
       ICET_NODE_SERIAL_ENQUEUE  = 16, /* a one-launch frame (see ICET_NODE_TIME_PHASES): enqueue the keyframe side on the calling thread, behind the loop's
                                           launch, instead of on the node's helper thread beside it (same result bits; for A/B timing)   */
       ICET_NODE_DOUBLE_W        = 32,   /* the solves run with ICET_FLAG_DOUBLE_W (include/icet_hip.h): the per-voxel weight in double instead of the reference's float
                                            CompleteOrthogonalDecomposition -- 2.4 us less per Gauss-Newton iteration (a 64-channel frame: 0.224 -> 0.205 ms), not the reference's
                                            arithmetic on thin voxels */
       ICET_NODE_TIME_PHASES     = 64 }; /* keep a frame's range filter and its loop apart, with timing events in between, for icet_node_last_timing (the arrangement
                                            of rounds 4-5).  Without it a pipelined odometry frame is ONE graph launch -- filter and loop together, no events: each costs
                                            the host 5-10 us in front of the launch -- and the keyframe build filters the raw frame a second time for itself on its own
                                            stream (same result bits either way) */

typedef struct icet_node_result {
    int32_t solved;           /* 0 for the first scan: it is only stored (odometry.cpp:46-52)                              */
    int32_t diverged;         /* 1 if the guard zeroed X                                                                   */
    int64_t n_kept;           /* rows of this scan after the range filter (= rows of the next frame's scan 1)              */
    float   X[6];             /* solution of this frame (after the guard)                                                  */
    float   pred_stds[6];     /* `it.pred_stds`                                                                            */
    float   pose[16];         /* accumulated X_homo, row-major 4 x 4 (odometry.cpp:91-98)                                  */
    float   quat[4];          /* x, y, z, w of Eigen::Quaternionf(X_homo.topLeftCorner(3,3)) (odometry.cpp:113-118)        */
    int64_t map_rows;         /* rows currently valid in the map queue (getQueue().rows())                                 */
} icet_node_result;

typedef struct icet_node icet_node;   /* opaque: previous scan in HBM, X0, pose, RNG, map queue */

/* The node borrows `ctx` (its device, stream and workspace); destroy the node before the context. */
icet_status icet_node_create(icet_ctx* ctx, const icet_node_params* p, icet_node** out);
icet_status icet_node_destroy(icet_node* node);
/* What went wrong in the last call on this node that did not return ICET_OK (the failing HIP call or the solve's own message); "" if nothing did. */
const char* icet_node_last_error(const icet_node* node);

/* One lidar frame.  `scan` is a HOST pointer in icet_node_push and a DEVICE pointer in icet_node_push_device (the
 * buffer may be reused as soon as the call returns: the node keeps its own filtered copy).  Both return after the frame's
 * result is on the host (the nodes publish every frame; the next frame's X0 depends on it). */
icet_status icet_node_push(icet_node* node, const float* scan, int64_t n, int64_t ld, icet_node_result* res);
icet_status icet_node_push_device(icet_node* node, const float* d_scan, int64_t n, int64_t ld, icet_node_result* res);

/* A BURST of frames, device pointers, results for all of them at the end: what a caller gets that replays a log or batches frames between publications.  The
 * frames are pushed one after the other by the call itself, with the bits of frame-by-frame pushes (what the caller saves is its own per-frame call).  Rounds 4-5
 * chained a burst's frames on the device (X0 <- X device to device, one copy of all results at the end); with two contexts that costs two hand-overs between streams
 * per frame, and since a frame is one graph launch (round 6) frame-by-frame is the faster arrangement on this part: 5.0-5.2 k frames/s against 4.0-4.7 k.  Every
 * frame's buffer must stay valid until the call returns.  results[k] belongs to frames[k] (a first-ever frame: solved = 0). */
icet_status icet_node_push_many_device(icet_node* node, const icet_dev_scan* frames, int32_t n_frames, icet_node_result* results);

/* `EigenQueue::getQueue()` (simpleMapMaker.cpp:43-50): copies the valid rows, oldest first, to the host as rows x 3
 * column-major with leading dimension `ld` (>= rows).  `out` may be NULL to query `rows` only. */
icet_status icet_node_map(icet_node* node, float* out, int64_t ld, int64_t* rows);

/* The node's `prev_pcl_matrix` (odometry.cpp:88): the scan the next frame will be registered against, i.e. the most
 * recent push after the range filter (the first push unfiltered).  rows x 3 column-major, leading dimension ld;
 * `out` may be NULL to query `rows` only. */
icet_status icet_node_prev_scan(icet_node* node, float* out, int64_t ld, int64_t* rows);

/* scanMatcher.cpp:76 / :79-84 (flags ICET_NODE_ALIGNED_CLOUD / ICET_NODE_SNAIL_TRAIL): rows x 3 column-major copies to the host. */
icet_status icet_node_aligned(icet_node* node, float* out, int64_t ld, int64_t* rows);
icet_status icet_node_snail_trail(icet_node* node, float* out, int64_t ld, int64_t* rows);

/* Device-side time of the pieces of the most recent push, measured with HIP events on the context's stream:
 * [0] range filter ms, [1] ICET solve ms, [2] map-queue kernel ms (0 if no map).  ICET_ERR_BAD_ARG when the push recorded none: the first cloud and the
 * one-launch odometry frame (range filter on, no map, no aligned cloud / snail trail, pipelined) unless the node has ICET_NODE_TIME_PHASES. */
icet_status icet_node_last_timing(icet_node* node, float out_ms[3]);

#ifdef __cplusplus
}
#endif
#endif /* ICET_NODES_H */
