"""Reproduce one draw of tests/param_sweep (seed, with_flags) and print what device and oracle do with it.  GPU box.  usage: diag_sweep_case.py seed [flags]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icet_amd
from oracle import pyoracle as po
from tests.param_sweep import draw_case, run_case, pools
seed = int(sys.argv[1]); wf = len(sys.argv) > 2 and sys.argv[2] == "1"
ctx = icet_amd.Context(0); rng = np.random.default_rng(seed); pl = pools()
for c in range(40):
    a, b, T, P, kw, runlen, x0 = draw_case(rng, pl, with_flags=wf)
    bits, d, r, ref, fits = run_case(ctx, a, b, T, P, kw, runlen, x0)
    bad = not np.isfinite(r["X"]).all() or not all(bits.values())
    only = [int(x[5:]) for x in sys.argv if x.startswith("case=")]
    if only and c in only:
        flag, mode = kw.get("_twin", (0, None)); okw = {k: v for k, v in kw.items() if k != "_twin"}
        for m_, tag in ((0, "oracle"), (po.DEVICE_ARITH, "oracle + device arithmetic"), (po.PINV3_DOUBLE, "oracle + double W"), (po.DEVICE_ARITH | po.PINV3_DOUBLE, "oracle + device arithmetic + double W")):
            o = po.solve(a, b, x0=x0, runlen=runlen, bins_phi=P, bins_theta=T, trace=True, mode=(mode or 0) | m_, **okw)
            print("case %d vs %-40s first update |dX| %s  final |dX| %s" % (c, tag, np.abs(r["aux"]["x_hist"][0] - o["trace"]["X"][0]).max(), np.abs(r["X"] - o["X"]).max()))
        bad = True
    if bad or "-v" in sys.argv:
        ax, t = r["aux"], ref["trace"]
        print("case", c, "T P", T, P, "kw", kw, "runlen", runlen, "x0", x0, "rows", a.shape, b.shape, "fits", fits)
        print("  device X", r["X"], "\n  oracle X", ref["X"], "\n  bits", {k: v for k, v in bits.items() if not v})
        for it in range(runlen):
            print("  it %d: device X %s cond %s | oracle X %s pruned %d eig %s" % (it, ax["x_hist"][it], ax["cond_info"][it], t["X"][it], t["pruned"][it], t["eigvals"][it]))
            print("        device HTWH diag %s htwdz %s\n        oracle HTWH diag %s htwdz %s" % (np.diag(ax["htwh"][it]), ax["htwdz"][it], np.diag(t["HTWH"][it]), t["HTWdz"][it]))
            act = (t["has_fit"] == 1) & (t["n1_raw"] > kw["n"]) & (t["bounds"][:, 5] > 1)
            print("        used voxels oracle %d; n2_raw diff %d n2_in diff %d" % (int(t["used"][it].sum()), int((ax["n2_raw"][it][act] != t["n2_raw"][it][act]).sum()), int((ax["n2_in"][it][act] != np.maximum(t["n2_in"][it][act], 0)).sum())))
