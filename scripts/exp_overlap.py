"""Experiment: does splitting a batch over several contexts/streams (keyframe of one part overlapping the GN loop of
another) raise aggregate throughput?  Run on the GPU box: python scripts/exp_overlap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import icet_amd
from icet_amd import lidar_sim, api

dev = torch.device("cuda", 0)
N = int(os.environ.get("N", "256"))
scans1, scans2 = [], []
for k in range(N):
    s1, s2, _ = lidar_sim.make_batch_pair(k, 64, 2048, device=dev, order="ring")
    scans1.append(s1); scans2.append(s2)
def padded(s):
    n = s.shape[1]; ld = (n + 63) // 64 * 64
    buf = torch.zeros((3, ld), dtype=torch.float32, device=s.device); buf[:, :n] = s
    return buf
b1 = [padded(s) for s in scans1]; b2 = [padded(s) for s in scans2]
d1 = [(b.data_ptr(), s.shape[1], b.shape[1]) for b, s in zip(b1, scans1)]
d2 = [(b.data_ptr(), s.shape[1], b.shape[1]) for b, s in zip(b2, scans2)]
p = api.Params(7, 24, 75, 25, 0.1, 0.1, 0)
out = torch.zeros((N, 48), dtype=torch.float32, device=dev)

def run(parts, order="seq", steps=6):
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    ctxs = [icet_amd.Context(0, stream=s.cuda_stream) for s in streams]
    sl = [list(range(i, N, parts)) for i in range(parts)]
    outs = [torch.zeros((len(ix), 48), dtype=torch.float32, device=dev) for ix in sl]
    for c, ix in zip(ctxs, sl):
        c.reserve(p, len(ix), sum(d1[i][1] for i in ix), sum(d2[i][1] for i in ix))
    def step():
        for c, ix, o in zip(ctxs, sl, outs):
            c.solve_batch_device([d1[i] for i in ix], [d2[i] for i in ix], p, o.data_ptr())
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = torch.zeros((N, 48), dtype=torch.float32, device=dev)
    for ix, o in zip(sl, outs): res[ix] = o
    for c in ctxs: c.close() if hasattr(c, "close") else None
    return dt, res

ref = None
for parts in (1, 2, 3, 4, 8):
    dt, res = run(parts)
    if ref is None: ref = res
    same = bool(torch.equal(ref, res))
    print("parts %d: %.3f ms/step  %.0f pairs/s  bitwise_same=%s" % (parts, dt * 1e3, N / dt, same), flush=True)
