"""Throughput of the default bench workload for different numbers of batch parts / stagger stages (icet_set_option).  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls, api
dev = torch.device("cuda", 0)
N = int(os.environ.get("N", "256"))
pairs = [ls.make_batch_pair(k, device=dev) for k in range(N)]
def padded(s):
    n = s.shape[1]; ld = (n + 63) // 64 * 64
    b = torch.zeros((3, ld), dtype=torch.float32, device=dev); b[:, :n] = s; return b
b1 = [padded(p[0]) for p in pairs]; b2 = [padded(p[1]) for p in pairs]
d1 = [(b.data_ptr(), p[0].shape[1], b.shape[1]) for b, p in zip(b1, pairs)]; d2 = [(b.data_ptr(), p[1].shape[1], b.shape[1]) for b, p in zip(b2, pairs)]
out = torch.zeros((N, 48), dtype=torch.float32, device=dev)
prm = api.Params(7, 24, 75, 25, 0.1, 0.1, 0)
ctx = icet_amd.Context(0)
for parts, stage in [(1, 4), (2, 4), (3, 4), (4, 4), (2, 1), (2, 2), (2, 3), (3, 1), (3, 2)]:
    ctx.set_option("batch_parts", parts); ctx.set_option("batch_stage", stage)
    for _ in range(3): ctx.solve_batch_device(d1, d2, prm, out.data_ptr())
    ctx.sync(); t = time.perf_counter()
    for _ in range(15): ctx.solve_batch_device(d1, d2, prm, out.data_ptr())
    ctx.sync(); dt = (time.perf_counter() - t) / 15
    print("parts %d stage %d: %.3f ms per step, %.0f pairs/s" % (parts, stage, dt * 1e3, N / dt), flush=True)
