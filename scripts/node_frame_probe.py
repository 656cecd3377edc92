import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls, api
dev = torch.device("cuda", 0)
frames = ls.make_sequence(64, motion=(0.25, 0.02, 0.005, 0.001, -0.001, 0.006), device=dev)
ctx = icet_amd.Context(0)
nd = api.Node(ctx, **api.ODOMETRY_NODE)
for k in range(8): nd.push_device(frames[k].data_ptr(), frames[k].shape[1], frames[k].shape[1])
t0 = time.perf_counter()
for k in range(8, 64): nd.push_device(frames[k].data_ptr(), frames[k].shape[1], frames[k].shape[1])
print("python loop: %.1f us per frame" % ((time.perf_counter() - t0) / 56 * 1e6))
nd.close()
