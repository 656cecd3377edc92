import sys, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import icet_amd
from icet_amd import lidar_sim as ls
ctx = icet_amd.Context(0)
g = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden")
base = [tuple(np.load(os.path.join(g, f))[k] for k in ("scan1", "scan2")) for f in ("scans_frame_804_805.npz", "scans_sample_pc_1_2.npz")]
routes = np.zeros((64, 7), int)
for k in range(64):
    R = ls.real_batch_rotation(k)
    a = np.ascontiguousarray((base[k % 2][0] @ R.T).astype(np.float32)); b = np.ascontiguousarray((base[k % 2][1] @ R.T).astype(np.float32))
    r = ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
    routes[k] = r["aux"]["cond_info"][:, 7].astype(int)
print("literal-route (2) count per iteration over 64 real pairs:", (routes == 2).sum(0).tolist())
print("pairs with any literal route:", np.nonzero((routes == 2).any(1))[0].tolist())
