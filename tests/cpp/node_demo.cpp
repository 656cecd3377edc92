// tests/cpp/node_demo.cpp -- C++ caller of include/icet_nodes.hpp in the shape of the reference's nodes: load frames
// from files (utils::loadPointCloudCSV's replacement / .npy), feed them to the callback one by one, print what the node
// would publish (odometry.cpp:99-131: position, quaternion, pred_stds).
// usage: node_demo odometry|mapmaker frame0 frame1 [frame2 ...]
#include <cstdio>
#include <cstring>
#include <vector>
#include "icet_nodes.hpp"

template <class Node> int run(Node& node, int argc, char** argv) {
    if (node.status != ICET_OK) { std::fprintf(stderr, "status %d: %s\n", (int)node.status, node.error.c_str()); return 1; }
    for (int k = 2; k < argc; k++) {
        int64_t rows = 0; icet_status st;
        std::vector<float> scan = icet_amd::loadScan(argv[k], &rows, &st);
        if (st != ICET_OK) { std::fprintf(stderr, "cannot load %s (status %d)\n", argv[k], (int)st); return 2; }
        const bool solved = node.pointCloudCallback(scan.data(), rows, rows);
        if (node.status != ICET_OK) { std::fprintf(stderr, "status %d: %s\n", (int)node.status, node.error.c_str()); return 1; }
        const icet_node_result& r = node.last;
        std::printf("frame %d solved %d kept %lld X %.9g %.9g %.9g %.9g %.9g %.9g pos %.9g %.9g %.9g quat %.9g %.9g %.9g %.9g map %lld\n", k - 2, (int)solved,
                    (long long)r.n_kept, r.X[0], r.X[1], r.X[2], r.X[3], r.X[4], r.X[5], r.pose[3], r.pose[7], r.pose[11], r.quat[0], r.quat[1], r.quat[2], r.quat[3],
                    (long long)r.map_rows);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: node_demo odometry|mapmaker frame0 frame1 ...\n"); return 2; }
    if (std::strcmp(argv[1], "odometry") == 0) { icet_amd::OdometryNode node; return run(node, argc, argv); }
    icet_amd::MapMakerNode node;
    const int rc = run(node, argc, argv);
    int64_t rows = 0; node.mapPC(&rows);
    std::printf("map_rows %lld\n", (long long)rows);
    return rc;
}
