"""Round 6 (review r5, weak 1): where do device and oracle part?  For the degenerate scenes and real rotated pairs: free-running |dX| / dH against the unmodified oracle and
against the oracle in DEVICE_ARITH mode (the device's documented deviations applied together), and every iteration REPLAYED from the oracle's state.  GPU box."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icet_amd
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po
ctx = icet_amd.Context(0)
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cases = {}
for nm in ("wall_s10", "ground_s10_m", "tunnel_s05", "ground_s02"):
    a, b, _ = ls.make_degenerate_named(nm); cases[nm] = (np.ascontiguousarray(a.T.numpy()), np.ascontiguousarray(b.T.numpy()))
base = [tuple(np.load(os.path.join(g, f))[k] for k in ("scan1", "scan2")) for f in ("scans_frame_804_805.npz", "scans_sample_pc_1_2.npz")]
for k in (0,):
    R = ls.real_batch_rotation(k)
    cases["real%d" % k] = (np.ascontiguousarray((base[k % 2][0] @ R.T).astype(np.float32)), np.ascontiguousarray((base[k % 2][1] @ R.T).astype(np.float32)))
for k in ():
    s1, s2, _ = ls.make_batch_pair(k); cases["bench%d" % k] = (np.ascontiguousarray(s1.T.numpy()), np.ascontiguousarray(s2.T.numpy()))
def relH(A, B): return float(np.abs(A - B).max() / np.abs(B).max())
for nm, (a, b) in cases.items():
    r = ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True); ax = r["aux"]
    for mode, tag in ((0, "oracle"), (po.DEVICE_ARITH, "oracle+device arithmetic"), (po.DEVICE_ARITH | po.PINV3_DOUBLE, "oracle+device arith+pinv3 dbl"), (po.PINV3_DOUBLE, "oracle+pinv3 double")):
        ref = po.solve(a, b, trace=True, mode=mode); t = ref["trace"]
        act = (t["has_fit"] == 1) & (t["n1_raw"] > 25) & (t["bounds"][:, 5] > 1)
        dX = np.abs(r["X"] - ref["X"])
        free = [relH(ax["htwh"][it], t["HTWH"][it]) for it in range(7)]
        rep, flips = [], []
        for it in range(7):
            x_in = np.zeros(6, np.float32) if it == 0 else t["X"][it - 1]
            rp = ctx.solve(a, b, 1, x_in, 24, 75, aux=True)
            rep.append(relH(rp["aux"]["htwh"][0], t["HTWH"][it])); flips.append(int((rp["aux"]["n2_raw"][0][act] != t["n2_raw"][it][act]).sum()))
        print("%-12s vs %-26s |dX| %.2e m %.2e rad  d pred_stds %.2e | free-running dH max %.2e | replayed dH per iter %s  count flips %s" % (
            nm, tag, dX[:3].max(), dX[3:].max(), float(np.abs(r["pred_stds"] - ref["pred_stds"]).max()), max(free), " ".join("%.1e" % x for x in rep), flips), flush=True)
