import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import icet_amd
from icet_amd import lidar_sim as ls
ctx = icet_amd.Context(0)
for name in ("tunnel_s10", "ground_s02", "ground_s10_m"):
    a, b, _ = ls.make_degenerate_named(name)
    a = np.ascontiguousarray(a.T.numpy()); b = np.ascontiguousarray(b.T.numpy())
    r = ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
    H = r["aux"]["htwh"][-1]
    print(name, flush=True)
    for _ in range(3): ctx.debug_gn_tail(H[None], np.ones((1, 6), np.float32))
