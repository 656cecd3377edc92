"""Per-frame wall time of the odometry node when frames arrive in HOST memory (icet_node_push) vs already in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls, api
dev = torch.device("cuda", 0)
frames = ls.make_sequence(24, motion=(0.25, 0.02, 0.005, 0.001, -0.001, 0.006), device=dev)
host = [np.ascontiguousarray(f.T.cpu().numpy()) for f in frames]
ctx = icet_amd.Context(0)
for mode in ("host", "device"):
    nd = api.Node(ctx, **api.ODOMETRY_NODE)
    for k in range(4):
        nd.push(host[k]) if mode == "host" else nd.push_device(frames[k].data_ptr(), frames[k].shape[1], frames[k].shape[1])
    t0 = time.perf_counter()
    for k in range(4, 24):
        nd.push(host[k]) if mode == "host" else nd.push_device(frames[k].data_ptr(), frames[k].shape[1], frames[k].shape[1])
    print("%s frames: %.3f ms/frame" % (mode, (time.perf_counter() - t0) / 20 * 1e3))
    nd.close()
