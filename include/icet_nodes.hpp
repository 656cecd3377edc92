// include/icet_nodes.hpp -- C++ host classes over include/icet_nodes.h and include/icet_io.h (Eigen-free, header only).
//
// `OdometryNode` / `MapMakerNode` hold what the reference's node classes hold between callbacks
// (/root/reference/src/odometry.cpp:170-186: prev_pcl_matrix, X0, X_homo; src/simpleMapMaker.cpp:241-262: + thresholds,
// the map queue `q`, the RNG) -- on the device -- and `pointCloudCallback(scan)` runs the per-frame body
// (odometry.cpp:46-98, simpleMapMaker.cpp:86-172) without the ROS publishing.  `loadPointCloudCSV` keeps the name and
// the second argument of utils::loadPointCloudCSV (include/utils.h:10) and returns the column-major N x 3 buffer.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "icet_host.hpp"
#include "icet_io.h"
#include "icet_nodes.h"

namespace icet_amd {

// utils::loadPointCloudCSV(filename, datasetType = "csv") -- "ouster" or the generic tab-separated branch (src/utils.cpp:12-91)
inline std::vector<float> loadPointCloudCSV(const std::string& filename, const std::string& datasetType = "csv", int64_t* rows = nullptr,
                                            icet_status* status = nullptr) {
    float* p = nullptr; int64_t n = 0;
    const icet_status st = icet_load_scan(filename.c_str(), datasetType == "ouster" ? ICET_FMT_OUSTER_CSV : ICET_FMT_XYZ_TSV, &p, &n);
    if (status) *status = st;
    std::vector<float> out;
    if (st == ICET_OK) out.assign(p, p + 3 * n);
    icet_free_scan(p);
    if (rows) *rows = (st == ICET_OK) ? n : 0;
    return out;
}

// any supported file (by extension): .npy, KITTI .bin, Ouster .csv, tab-separated text
inline std::vector<float> loadScan(const std::string& filename, int64_t* rows, icet_status* status = nullptr) {
    float* p = nullptr; int64_t n = 0;
    const icet_status st = icet_load_scan(filename.c_str(), ICET_FMT_AUTO, &p, &n);
    if (status) *status = st;
    std::vector<float> out;
    if (st == ICET_OK) out.assign(p, p + 3 * n);
    icet_free_scan(p);
    if (rows) *rows = (st == ICET_OK) ? n : 0;
    return out;
}

class LidarNode {
public:
    explicit LidarNode(const icet_node_params& p, int device = 0) {
        icet_ctx* ctx = thread_context(device, &status);
        if (!ctx) { error = "icet_create failed (no usable HIP device; this path has no CPU fallback)"; return; }
        status = icet_node_create(ctx, &p, &node_);
        if (status != ICET_OK) error = "icet_node_create failed";
    }
    ~LidarNode() { if (node_) icet_node_destroy(node_); }
    LidarNode(const LidarNode&) = delete;
    LidarNode& operator=(const LidarNode&) = delete;

    // scan: column-major N x 3 host buffer, leading dimension ld.  Returns false for the first frame (nothing solved yet).
    bool pointCloudCallback(const float* scan, int64_t n, int64_t ld) {
        if (!node_) return false;
        status = icet_node_push(node_, scan, n, ld, &last);
        if (status != ICET_OK) { error = std::string("icet_node_push failed: ") + icet_node_last_error(node_); return false; }
        return last.solved != 0;
    }
    // A burst of frames that already live in HBM (icet_node_push_many_device): pushed one after the other by the library; results[k] belongs to
    // frames[k] (`last` = the burst's last frame).  Returns the number of frames that were solved (a first-ever frame is only stored).
    int pointCloudBurst(const icet_dev_scan* frames, int n_frames, std::vector<icet_node_result>& results) {
        results.assign(n_frames > 0 ? (size_t)n_frames : 0, icet_node_result{});
        if (!node_ || n_frames <= 0) return 0;
        status = icet_node_push_many_device(node_, frames, n_frames, results.data());
        if (status != ICET_OK) { error = std::string("icet_node_push_many_device failed: ") + icet_node_last_error(node_); return 0; }
        last = results.back();
        int solved = 0; for (const icet_node_result& r : results) solved += r.solved != 0;
        return solved;
    }
    // EigenQueue::getQueue(): rows x 3 column-major
    std::vector<float> mapPC(int64_t* rows) {
        int64_t r = 0; std::vector<float> out;
        if (!node_ || icet_node_map(node_, nullptr, 0, &r) != ICET_OK) { if (rows) *rows = 0; return out; }
        out.assign((size_t)3 * (r > 0 ? r : 1), 0.f);
        if (r) status = icet_node_map(node_, out.data(), r, &r);
        out.resize((size_t)3 * r);
        if (rows) *rows = r;
        return out;
    }
    icet_node_result last{};       // X, pred_stds, X_homo (pose), quaternion of the most recent frame
    icet_status status = ICET_OK;
    std::string error;

private:
    icet_node* node_ = nullptr;
};

// odometry_node's settings: src/odometry.cpp:58 (minD = 2), :73-76 (7, 24, 75), :82 (X0 <- X)
struct OdometryNode : LidarNode {
    explicit OdometryNode(int device = 0) : LidarNode(icet_node_params{{7, 24, 75, 25, 0.1f, 0.1f, ICET_FLAG_NONE}, 2.0f, 1, 0.f, 0.f, 0, 0, 0}, device) {}
};

// map_maker_node's settings: src/simpleMapMaker.cpp:62 (600000 x 3), :98 (minD = 0.2), :113-119, :124 (X0 <- 0), :147, :241-242
struct MapMakerNode : LidarNode {
    explicit MapMakerNode(int device = 0)
        : LidarNode(icet_node_params{{12, 24, 75, 25, 0.1f, 0.1f, ICET_FLAG_NONE}, 0.2f, 0, 0.3f, 0.3f, 600000, 2000, 0}, device) {}
};

// scan_registration_node's settings: src/scanMatcher.cpp:44 (no range filter), :55-62 (7, 24, 75, X0 = 0), :76 and :79-84 (published clouds)
struct ScanRegistrationNode : LidarNode {
    explicit ScanRegistrationNode(int device = 0)
        : LidarNode(icet_node_params{{7, 24, 75, 25, 0.1f, 0.1f, ICET_FLAG_NONE}, 0.f, 0, 0.f, 0.f, 0, 0,
                                     ICET_NODE_NO_RANGE_FILTER | ICET_NODE_ALIGNED_CLOUD | ICET_NODE_SNAIL_TRAIL}, device) {}
};

}  // namespace icet_amd
