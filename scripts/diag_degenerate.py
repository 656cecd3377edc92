"""Device vs oracle on the degenerate scenes (icet_amd/lidar_sim.DEGENERATE_SCENES): pruned counts, eigenvalue / HTWH / pred_stds / X deltas,
with and without ICET_FLAG_ROUNDTRIP_SCAN2, next to the oracle's own 1-ulp sensitivity.  Run on the GPU box (scripts/: uses the oracle)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import icet_amd
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po
ctx = icet_amd.Context(0)
rng = np.random.default_rng(123)
for name in ls.DEGENERATE_SCENES:
    a, b, _ = ls.make_degenerate_named(name)
    a = np.ascontiguousarray(a.T.numpy()); b = np.ascontiguousarray(b.T.numpy())
    ref = po.solve(a, b, trace=True); t = ref["trace"]
    # the oracle against itself under a 1-ulp perturbation of scan 2
    sens = dict(ev=0.0, H=0.0, ps=0.0, Xt=0.0, Xr=0.0, pruned_same=True)
    for _ in range(3):
        bp = (b.astype(np.float64) * (1.0 + rng.uniform(-1e-7, 1e-7, b.shape))).astype(np.float32)
        o2 = po.solve(a, bp, trace=True); t2 = o2["trace"]
        sens["ev"] = max(sens["ev"], float((np.abs(t2["eigvals"] - t["eigvals"]).max(1) / np.abs(t["eigvals"]).max(1)).max()))
        sens["H"] = max(sens["H"], float((np.abs(t2["HTWH"] - t["HTWH"]).reshape(7, -1).max(1) / np.abs(t["HTWH"]).reshape(7, -1).max(1)).max()))
        sens["ps"] = max(sens["ps"], float(np.abs(o2["pred_stds"] - ref["pred_stds"]).max()))
        sens["Xt"] = max(sens["Xt"], float(np.abs(o2["X"][:3] - ref["X"][:3]).max())); sens["Xr"] = max(sens["Xr"], float(np.abs(o2["X"][3:] - ref["X"][3:]).max()))
        sens["pruned_same"] = sens["pruned_same"] and bool(np.array_equal(t2["pruned"], t["pruned"]))
    print("%-13s oracle: pruned %s cond %.3g pred_stds %s | 1-ulp sensitivity: ev %.2e H %.2e ps %.2e X %.2e m %.2e rad pruned_same %s" % (
        name, t["pruned"].tolist(), t["eigvals"][-1, 5] / t["eigvals"][-1, 0], np.round(ref["pred_stds"], 4).tolist(), sens["ev"], sens["H"], sens["ps"], sens["Xt"], sens["Xr"], sens["pruned_same"]))
    for flags in (0, 16):
        r = ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True, flags=flags); ax = r["aux"]; ci = ax["cond_info"]
        dev_ev = np.where(np.isnan(ci[:, :6]), t["eigvals"], ci[:, :6])
        print("   flags %2d device: pruned %s route %s pred_stds %s | d ev %.2e (weakest rel-to-itself %.2e) d H %.2e d ps %.2e dX %.2e m %.2e rad n2_in diff %d" % (
            flags, ci[:, 6].astype(int).tolist(), ci[:, 7].astype(int).tolist(), np.round(r["pred_stds"], 4).tolist(),
            float((np.abs(dev_ev - t["eigvals"]).max(1) / np.abs(t["eigvals"]).max(1)).max()), float(np.nanmax(np.abs(dev_ev[:, 0] / t["eigvals"][:, 0] - 1))),
            float((np.abs(ax["htwh"] - t["HTWH"]).reshape(7, -1).max(1) / np.abs(t["HTWH"]).reshape(7, -1).max(1)).max()),
            float(np.abs(r["pred_stds"] - ref["pred_stds"]).max()), float(np.abs(r["X"][:3] - ref["X"][:3]).max()), float(np.abs(r["X"][3:] - ref["X"][3:]).max()),
            int((ax["n2_in"][0] != np.maximum(t["n2_in"][0], 0)).sum())))
