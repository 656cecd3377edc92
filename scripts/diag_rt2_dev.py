"""Per-iteration view of ICET_FLAG_ROUNDTRIP_SCAN2 on chosen bench pairs: the device without / with the flag against the oracle with / without the
round trips (X after every iteration, voxels whose in-bounds counts differ).  Run on the GPU box.  usage: diag_rt2_dev.py 18 176"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls, api
from oracle import pyoracle as po
dev = torch.device("cuda", 0)
ctx = icet_amd.Context(0)
for k in [int(v) for v in sys.argv[1:]] or [18, 176]:
    s1, s2, _ = ls.make_batch_pair(k, device=dev)
    a, b = s1.T.cpu().numpy(), s2.T.cpu().numpy()
    t = po.solve(a, b, trace=True)["trace"]; ts = po.solve(a, b, trace=True, mode=po.SKIP_RT2)["trace"]
    g0 = ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)["aux"]
    g1 = ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True, flags=api.FLAG_ROUNDTRIP_SCAN2)["aux"]
    f = t["has_fit"] == 1
    print("pair %d   (X after each iteration: max |difference|; voxels whose n2_in differ)" % k)
    for it in range(7):
        dv = lambda g, o: [int(v) for v in np.nonzero(f & (g["n2_in"][it] != o["n2_in"][it]))[0][:4]]
        print("%d  plain-ref %.2e  rt-ref %.2e  plain-skip %.2e  rt-skip %.2e  rt-plain %.2e | n2_in: plain/ref %s plain/skip %s rt/ref %s rt/skip %s ref/skip %s" % (
            it, np.abs(g0["x_hist"][it] - t["X"][it]).max(), np.abs(g1["x_hist"][it] - t["X"][it]).max(), np.abs(g0["x_hist"][it] - ts["X"][it]).max(),
            np.abs(g1["x_hist"][it] - ts["X"][it]).max(), np.abs(g1["x_hist"][it] - g0["x_hist"][it]).max(),
            dv(g0, t), dv(g0, ts), dv(g1, t), dv(g1, ts), [int(v) for v in np.nonzero(f & (t["n2_in"][it] != ts["n2_in"][it]))[0][:4]]))
