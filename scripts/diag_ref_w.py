"""Round 6: ICET_FLAG_REFERENCE_W (per-voxel W by the reference's float COD) against the UNMODIFIED oracle, beside the default path.  GPU box."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icet_amd
from icet_amd import lidar_sim as ls, api
from oracle import pyoracle as po
from tests.param_sweep import draw_case, pools
ctx = icet_amd.Context(0)
cases = {}
for nm in ("wall_s10", "ground_s10_m", "ground_s05", "ground_s02", "tunnel_s05", "wall_s30"):
    a, b, _ = ls.make_degenerate_named(nm); cases[nm] = (np.ascontiguousarray(a.T.numpy()), np.ascontiguousarray(b.T.numpy()), 75, 24, dict(n=25, thresh=0.1, buff=0.1), 7, np.zeros(6, np.float32))
rng = np.random.default_rng(12); pl = pools()
for c in range(5):
    a, b, T, P, kw, runlen, x0 = draw_case(rng, pl, with_flags=True)
    if c == 4: cases["sweep12_case4"] = (a, b, T, P, {k: v for k, v in kw.items() if k != "_twin"}, runlen, x0)
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
d = np.load(os.path.join(g, "scans_frame_804_805.npz")); cases["frame_804_805"] = (d["scan1"], d["scan2"], 75, 24, dict(n=25, thresh=0.1, buff=0.1), 7, np.zeros(6, np.float32))
s1, s2, _ = ls.make_batch_pair(0); cases["bench0"] = (np.ascontiguousarray(s1.T.numpy()), np.ascontiguousarray(s2.T.numpy()), 75, 24, dict(n=25, thresh=0.1, buff=0.1), 7, np.zeros(6, np.float32))
def relH(A, B): return float(np.abs(A - B).max() / max(np.abs(B).max(), 1e-30))
for nm, (a, b, T, P, kw, rl, x0) in cases.items():
    ref = po.solve(a, b, x0=x0, runlen=rl, bins_phi=P, bins_theta=T, trace=True, **kw); t = ref["trace"]
    for flag, tag in ((0, "default (double W)"), (api.FLAG_REFERENCE_W, "ICET_FLAG_REFERENCE_W")):
        r = ctx.solve(a, b, rl, x0, P, T, aux=True, flags=flag, **kw); ax = r["aux"]
        dX = np.abs(r["X"] - ref["X"]); d0 = np.abs(ax["x_hist"][0] - t["X"][0])
        print("%-14s %-24s |dX| %.2e m %.2e rad  first update %.2e  d pred_stds %.2e  dH per iter %s  pruned dev %s orc %s" % (
            nm, tag, dX[:3].max(), dX[3:].max(), d0.max(), float(np.abs(r["pred_stds"] - ref["pred_stds"]).max()), " ".join("%.1e" % relH(ax["htwh"][i], t["HTWH"][i]) for i in range(rl)),
            ax["cond_info"][:, 6].astype(int).tolist(), t["pruned"].tolist()), flush=True)
