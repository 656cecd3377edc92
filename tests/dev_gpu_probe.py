"""Development probe (not a test): run the GPU path on the sample frames and print diffs vs golden."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icet_amd
from tests.conftest import load_pair, load_golden

for name in ("frame_804_805", "sample_pc_1_2"):
    a, b = load_pair(name); g = load_golden(name)
    ctx = icet_amd.default_context(0)
    t = time.time(); r = ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True); dt = time.time() - t
    ax = r["aux"]
    print("==", name, "first-call s", round(dt, 3))
    print("X gpu   ", r["X"]); print("X oracle", g["X"]); print("dX", r["X"] - g["X"])
    print("stds gpu", r["pred_stds"]); print("stds orc", g["pred_stds"])
    print("n1 eq", (ax["n1_raw"] == g["n1_raw"]).all(), "n mism", (ax["n1_raw"] != g["n1_raw"]).sum())
    print("bounds maxdiff", np.abs(ax["cluster_bounds"] - g["bounds"]).max(), "rows differing", (np.abs(ax["cluster_bounds"] - g["bounds"]).max(1) > 0).sum())
    print("has_fit eq", (ax["has_fit"] == g["has_fit"]).all(), ax["has_fit"].sum(), g["has_fit"].sum())
    f = g["has_fit"] == 1
    print("mu1 maxdiff", np.abs(ax["mu1"] - g["mu1"])[f].max(), "sigma1 maxdiff", np.abs(ax["sigma1"] - g["sigma1"])[f].max())
    print("L mismatches", (ax["l_diag"] != g["Ldiag"])[f].any(1).sum())
    ev = np.abs(np.abs(np.einsum("vij,vij->vj", ax["evecs1"][f], g["evecs1"][f])) - 1).max()
    sg = (np.einsum("vij,vij->vj", ax["evecs1"][f], g["evecs1"][f]) < 0).sum()
    print("evec |dot|-1 max", ev, "sign flips", sg)
    for it in range(7):
        m = g["n2_raw"][it] * 0
        act = (g["has_fit"] == 1) & (g["n1_raw"] > 25) & (g["bounds"][:, 5] > 1)
        print(" it", it, "n2_raw mism", (ax["n2_raw"][it][act] != g["n2_raw"][it][act]).sum(), "n2_in mism", (ax["n2_in"][it][act] != np.maximum(g["n2_in"][it][act], 0)).sum(),
              "X diff", np.abs(ax["x_hist"][it] - g["X_hist"][it]).max(), "HTWH rel", np.abs(ax["htwh"][it] - g["HTWH"][it]).max() / np.abs(g["HTWH"][it]).max())
    t = time.time()
    for _ in range(5): r = ctx.solve(a, b, 7, np.zeros(6), 24, 75)
    print("warm host-pointer solve ms", (time.time() - t) / 5 * 1e3, ctx.last_timing())
