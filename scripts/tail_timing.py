"""What the literal 6x6 route (gn_tail_literal, icet_device_math.h) costs a frame: the degenerate scenes of icet_amd/lidar_sim against an ordinary pair, single-pair
solves with the scans resident on the host (icet_solve) -- keyframe / loop milliseconds of the call, the routes taken -- and the route alone through the test hook
(icet_debug_gn_tail on n matrices = n one-wave blocks: throughput; n = 1: latency incl. the hook's copies).  ICET_HIP_LIB selects the library build (A/B).
Run on the GPU box."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import icet_amd
from icet_amd import lidar_sim as ls
ctx = icet_amd.Context(0)
def timed(a, b, reps=30):
    ks, ls_, ws = [], [], []
    r = None
    for k in range(reps):
        t0 = time.perf_counter()
        r = ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=(k == 0))
        ws.append((time.perf_counter() - t0) * 1e3)
        if k == 0: r0 = r; routes = r["aux"]["cond_info"][:, 7].astype(int).tolist(); pruned = r["aux"]["cond_info"][:, 6].astype(int).tolist()
        t = ctx.last_timing(); ks.append(t["keyframe_ms"]); ls_.append(t["gn_loop_ms"])
    return np.median(ks[3:]), np.median(ls_[3:]), np.median(ws[3:]), routes, pruned, r0
s1, s2, _ = ls.make_pair()
a = np.ascontiguousarray(s1.T.numpy()); b = np.ascontiguousarray(s2.T.numpy())
k, l, w, routes, pruned, _ = timed(a, b)
print("%-13s keyframe %.3f ms loop %.3f ms wall %.3f ms routes %s pruned %s" % ("ordinary", k, l, w, routes, pruned))
for name in ls.DEGENERATE_SCENES:
    a, b, _ = ls.make_degenerate_named(name)
    a = np.ascontiguousarray(a.T.numpy()); b = np.ascontiguousarray(b.T.numpy())
    k, l, w, routes, pruned, r = timed(a, b)
    print("%-13s keyframe %.3f ms loop %.3f ms wall %.3f ms routes %s pruned %s X %s" % (name, k, l, w, routes, pruned, np.array2string(r["X"], precision=6)))
    H = r["aux"]["htwh"][-1]; g = np.zeros(6, np.float32)
    if name == "tunnel_s10":
        for n in (1, 4096):
            Hs = np.repeat(H[None], n, 0); gs = np.repeat(g[None], n, 0)
            ctx.debug_gn_tail(Hs, gs)
            t0 = time.perf_counter()
            for _ in range(20): o = ctx.debug_gn_tail(Hs, gs)
            dt = (time.perf_counter() - t0) / 20
            print("   icet_debug_gn_tail n = %d: %.1f us per call (route %d)" % (n, dt * 1e6, int(o["route"][0])))
