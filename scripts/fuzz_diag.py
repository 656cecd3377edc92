"""Looks into single cases of scripts/fuzz_params.py (same seed -> same draws): per-iteration tables against the oracle and the oracle's own 1-ulp spread.
Usage: python scripts/fuzz_diag.py seed case [case ...] [flags]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from icet_amd import api
from oracle import pyoracle as po
from tests.param_sweep import draw_case, run_case, pools as make_pools

flags = "flags" in sys.argv
seed = int(sys.argv[1]); want = [int(x) for x in sys.argv[2:] if x != "flags"]
rng = np.random.default_rng(seed)
pools = make_pools()
ctx = api.Context()
np.set_printoptions(linewidth=220, precision=6)
for c in range(max(want) + 1):
    a, b, T, P, kw, runlen, x0 = draw_case(rng, pools, with_flags=flags)
    if c not in want: continue
    bits, d, r, ref, fits = run_case(ctx, a, b, T, P, kw, runlen, x0)
    t, ax = ref["trace"], r["aux"]
    print("=== case", c, "T", T, "P", P, kw, "runlen", runlen, "x0", x0, "fits", fits, "bits", {k: v for k, v in bits.items() if not v})
    f = t["has_fit"] == 1
    if not bits["evecs1"]:
        bad = np.nonzero(f & (ax["evecs1"].view(np.uint32) != t["evecs1"].view(np.uint32)).reshape(f.size, -1).any(axis=1))[0]
        for v in bad[:4]:
            print(" voxel", v, "n1", t["n1_raw"][v], "bounds", t["bounds"][v], "Ldiag", t["Ldiag"][v], ax["l_diag"][v])
            print("  sigma1", t["sigma1"][v].ravel())
            print("  evecs oracle", t["evecs1"][v].ravel()); print("  evecs device", ax["evecs1"][v].ravel())
    act = f & (t["n1_raw"] > kw["n"]) & (t["bounds"][:, 5] > 1)
    if not bits["n2_raw0"]:
        bad = np.nonzero(act & (ax["n2_raw"][0] != t["n2_raw"][0]))[0]
        print(" n2_raw[0] differs in", bad.size, "voxels of", int(act.sum()), "active;  sum device", int(ax["n2_raw"][0][act].sum()), "oracle", int(t["n2_raw"][0][act].sum()))
        for v in bad[:8]:
            print("  voxel", v, "(theta bin", v // P, "phi bin", v % P, ") device", ax["n2_raw"][0][v], "oracle", t["n2_raw"][0][v], "n2_in device", ax["n2_in"][0][v], "oracle", t["n2_in"][0][v], "bounds", t["bounds"][v])
    for it in range(runlen):
        dr = (ax["n2_raw"][it][act] != t["n2_raw"][it][act]).sum(); di = (ax["n2_in"][it][act] != np.maximum(t["n2_in"][it][act], 0)).sum()
        used_o = t["used"][it] if "used" in t else None
        hd = np.abs(ax["htwh"][it] - t["HTWH"][it]).max() / max(np.abs(t["HTWH"][it]).max(), 1e-30)
        ev = np.linalg.eigvalsh(t["HTWH"][it].astype(np.float64))
        print("  it", it, "n2_raw diff", dr, "n2_in diff", di, "rel dHTWH %.2e" % hd, "cond %.2e" % (ev[-1] / max(ev[0], 1e-300)), "pruned o/d", t["pruned"][it], int(ax["cond_info"][it][6]) if "cond_info" in ax else "?",
              "dX %.2e" % np.abs(ax["x_hist"][it] - t["X"][it]).max())
    base = ref["X"]; spread = np.zeros(6); spread0 = np.zeros(6)
    prng = np.random.default_rng(123)
    for _ in range(4):
        bp = (b.astype(np.float64) * (1.0 + prng.uniform(-1e-7, 1e-7, b.shape))).astype(np.float32)
        okw = {k: v for k, v in kw.items() if k != "_twin"}
        if kw.get("_twin", (0, None))[1] is not None: okw["mode"] = kw["_twin"][1]
        pr = po.solve(a, bp, x0=x0, runlen=runlen, bins_phi=P, bins_theta=T, trace=True, **okw)
        spread = np.maximum(spread, np.abs(pr["X"] - base)); spread0 = np.maximum(spread0, np.abs(pr["trace"]["X"][0] - t["X"][0]))
    print("  |X_gpu - X_oracle|", d, "\n  oracle 1-ulp spread", spread, "\n  first iteration: |X_gpu - X_oracle|", np.abs(ax["x_hist"][0] - t["X"][0]), "oracle spread", spread0)
    print("  used voxels it0 (oracle):", int(t["used"][0].sum()) if "used" in t else "?", " HTWH it0 oracle diag", np.diag(t["HTWH"][0]), "\n  device diag", np.diag(ax["htwh"][0]))
