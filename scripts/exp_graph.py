"""Single-pair latency with and without the hipGraph replay (option "graph"), bits compared.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls, api
dev = torch.device("cuda", 0)
for tag, rings, steps, T, P, iters in (("64-ch 75x24 7 it", 64, 2048, 75, 24, 7), ("128-ch 150x48 10 it", 128, 4096, 150, 48, 10)):
    s1, s2, _ = ls.make_pair(9000, 9001, ls.DEFAULT_MOTION, rings, steps, device=dev) if rings == 128 else ls.make_batch_pair(0, device=dev)
    def padded(s):
        n = s.shape[1]; ld = (n + 63) // 64 * 64
        b = torch.zeros((3, ld), dtype=torch.float32, device=dev); b[:, :n] = s; return b
    b1, b2 = padded(s1), padded(s2)
    d1 = [(b1.data_ptr(), s1.shape[1], b1.shape[1])]; d2 = [(b2.data_ptr(), s2.shape[1], b2.shape[1])]
    p = api.Params(iters, P, T, 25, 0.1, 0.1, 0)
    stream = torch.cuda.Stream(device=dev)
    res = {}
    for g in (0, 1, 0, 1):
        ctx = icet_amd.Context(0, stream=stream.cuda_stream)
        ctx.set_option("graph", g)
        out = torch.zeros((1, 48), dtype=torch.float32, device=dev)
        for _ in range(6):
            ctx.solve_batch_device(d1, d2, p, out.data_ptr()); ctx.sync()
        t0 = time.perf_counter(); n = 300
        for _ in range(n):
            ctx.solve_batch_device(d1, d2, p, out.data_ptr()); ctx.sync()
        ms = (time.perf_counter() - t0) / n * 1e3
        # back-to-back without a host sync in between: what the device alone needs
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.solve_batch_device(d1, d2, p, out.data_ptr())
        ctx.sync()
        ms2 = (time.perf_counter() - t0) / n * 1e3
        res[g] = out.cpu().numpy().copy()
        print("%s graph=%d: %.4f ms per solve+sync, %.4f ms per solve back to back" % (tag, g, ms, ms2), flush=True)
        ctx.close()
    print("bits equal:", np.array_equal(res[0], res[1]))
