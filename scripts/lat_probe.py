"""Single-pair latency: wall clock per solve vs HIP-event time of its keyframe / GN-loop parts.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import icet_amd
from icet_amd import lidar_sim as ls, api
dev = torch.device("cuda", 0)
s1, s2, _ = ls.make_batch_pair(0, device=dev)
def padded(s):
    n = s.shape[1]; ld = (n + 63) // 64 * 64
    b = torch.zeros((3, ld), dtype=torch.float32, device=dev); b[:, :n] = s; return b
b1, b2 = padded(s1), padded(s2)
d1 = [(b1.data_ptr(), s1.shape[1], b1.shape[1])]; d2 = [(b2.data_ptr(), s2.shape[1], b2.shape[1])]
out = torch.zeros((1, 48), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
ctx = icet_amd.Context(0)
for flags in (0, api.FLAG_TIMING):
    p = api.Params(7, 24, 75, 25, 0.1, 0.1, flags)
    for _ in range(5): ctx.solve_batch_device(d1, d2, p, out.data_ptr()); ctx.sync()
    t0 = time.perf_counter(); enq = 0.0
    for _ in range(50):
        e0 = time.perf_counter(); ctx.solve_batch_device(d1, d2, p, out.data_ptr()); enq += time.perf_counter() - e0; ctx.sync()
    wall = (time.perf_counter() - t0) / 50 * 1e3
    t = ctx.last_timing()
    print("flags %d: wall %.3f ms/solve, host enqueue %.3f ms, events: keyframe %.3f + gn loop %.3f = %.3f ms, accumulate total %.3f" % (
        flags, wall, enq / 50 * 1e3, t["keyframe_ms"], t["gn_loop_ms"], t["keyframe_ms"] + t["gn_loop_ms"], t["accumulate_ms"]))
