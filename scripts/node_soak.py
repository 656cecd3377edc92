"""Soak of the one-launch frame path: thousands of frames through an odometry node (helper thread, pinned done word, folded filter) against a node that keeps the phased
frame (ICET_NODE_TIME_PHASES) -- every frame's X / pred_stds / pose / kept rows bit for bit; the map maker likewise (ring contents at the end).  Usage (GPU box):
python scripts/node_soak.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls, api
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
frames = ls.make_sequence(48, motion=(0.25, 0.02, 0.005, 0.001, -0.001, 0.006), device=dev)
frames = [f if k % 5 else f[:, : f.shape[1] - 311 * (k % 7)].contiguous() for k, f in enumerate(frames)]       # ragged row counts
ctx = icet_amd.Context(0)
for name, kw in (("odometry", api.ODOMETRY_NODE), ("map maker", dict(api.MAP_MAKER_NODE, map_capacity=50000, runlen=4))):
    a, b = api.Node(ctx, **kw), api.Node(ctx, **dict(kw, flags=api.NODE_TIME_PHASES))
    bad = 0; t0 = time.perf_counter()
    for k in range(n_frames):
        f = frames[(k * 7) % len(frames)]
        ra, rb = a.push_device(f.data_ptr(), f.shape[1], f.shape[1]), b.push_device(f.data_ptr(), f.shape[1], f.shape[1])
        if not (np.array_equal(ra["X"], rb["X"]) and np.array_equal(ra["pred_stds"], rb["pred_stds"]) and np.array_equal(ra["pose"], rb["pose"], equal_nan=True) and ra["n_kept"] == rb["n_kept"] and ra["diverged"] == rb["diverged"]):
            bad += 1
            if bad < 4: print(name, "frame", k, "differs", ra["X"], rb["X"])
    same_map = True
    if kw.get("map_capacity", 0): same_map = bool(np.array_equal(a.map(), b.map()))
    print("%s: %d frames, %d differ from the phased node, maps equal %s, %.1f s" % (name, n_frames, bad, same_map, time.perf_counter() - t0))
    a.close(); b.close()

# two nodes driven from two host threads at once (each its own context, helper thread and streams) against the results of one node alone
import threading
ref_node = api.Node(ctx, **api.ODOMETRY_NODE)
m = min(n_frames, 1500)
ref = []
for k in range(m):
    f = frames[(k * 7) % len(frames)]
    ref.append(ref_node.push_device(f.data_ptr(), f.shape[1], f.shape[1])["X"].copy())
ref_node.close()
bad_t = [0, 0]
def drive(i):
    c = icet_amd.Context(0); nd = api.Node(c, **api.ODOMETRY_NODE)
    for k in range(m):
        f = frames[(k * 7) % len(frames)]
        r = nd.push_device(f.data_ptr(), f.shape[1], f.shape[1])
        if not np.array_equal(r["X"], ref[k]): bad_t[i] += 1
    nd.close(); c.close()
ts = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
print("two threads, two nodes: %d frames each, frames that differ from one node alone: %s, %.1f s" % (m, bad_t, time.perf_counter() - t0))
