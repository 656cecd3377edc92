"""Adversarial scans (tests/param_sweep.spoil: equal sort keys by the thousand, lattice points exactly on voxel edges, duplicates, NaN / inf, extreme magnitudes, signed
zeros) through the GPU path and the oracle: keyframe table and first-iteration counts must be equal bits.  Usage (GPU box): python scripts/fuzz_adversarial.py [cases] [seed] [rt2]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from icet_amd import api
from tests.param_sweep import draw_adversarial, run_case, pools as make_pools

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rt2 = len(sys.argv) > 3 and sys.argv[3] == "rt2"       # ICET_FLAG_ROUNDTRIP_SCAN2: the device round-trips scan 2 through spherical coordinates like the reference
rng = np.random.default_rng(seed)
pools = make_pools(); ctx = api.Context(); bad = 0
for c in range(cases):
    a, b, T, P, kw, runlen, x0, what = draw_adversarial(rng, pools)
    if rt2: kw["_twin"] = (api.FLAG_ROUNDTRIP_SCAN2, None)
    bits, d, r, ref, fits = run_case(ctx, a, b, T, P, kw, runlen, x0)
    ok = all(bits.values()); bad += 0 if ok else 1
    fin = np.isfinite(r["X"]).all() == np.isfinite(ref["X"]).all()
    print("case %3d n1=%6d T=%3d P=%2d n=%3d thresh=%.2f buff=%.1f runlen=%d fits=%4d bits=%s finite=%s dX=%.2e  [%s]%s" % (
        c, a.shape[0], T, P, kw["n"], kw["thresh"], kw["buff"], runlen, fits, "ok" if ok else "DIFF", fin, np.nanmax(d) if np.isfinite(d).any() else float("nan"), what,
        "" if ok else "  " + str({k: v for k, v in bits.items() if not v})), flush=True)
print("cases with differing bits:", bad)
