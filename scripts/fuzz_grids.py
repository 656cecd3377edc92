"""Extreme grids through the GPU path and the oracle (tests/param_sweep.run_case): one voxel, one ring of voxels, one column, the 10 000-voxel limit in three shapes,
grids just above the limit (refused on the device).  Usage (GPU box): python scripts/fuzz_grids.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from icet_amd import api
from tests.param_sweep import run_case, pools as make_pools

pools = make_pools(); ctx = api.Context(); bad = 0
a, b = pools[0]
for (T, P) in ((1, 1), (1, 24), (75, 1), (2, 2), (3, 1), (200, 50), (100, 100), (10000, 1), (1, 10000), (5000, 2), (2, 5000), (2500, 4), (4, 2500), (624, 16), (16, 624), (1250, 8), (8, 1250)):
    for x0 in (np.zeros(6, np.float32), np.array([0.1, -0.05, 0.02, 0.001, -0.002, 0.01], np.float32)):
        kw = dict(n=25, thresh=0.1, buff=0.1)
        try:
            bits, d, r, ref, fits = run_case(ctx, a, b, T, P, kw, 4, x0)
        except api.IcetError as e:
            print("T=%5d P=%5d x0=%s: ERROR %s" % (T, P, "0" if not x0.any() else "r", str(e)[:120]), flush=True); bad += 1
            ctx = api.Context()
            continue
        ok = all(bits.values()); bad += 0 if ok else 1
        print("T=%5d P=%5d x0=%s fits=%5d bits=%s dX_t=%.2e dX_r=%.2e%s" % (T, P, "0" if not x0.any() else "r", fits, "ok" if ok else "DIFF", d[:3].max(), d[3:].max(),
              "" if ok else "  " + str({k: v for k, v in bits.items() if not v})), flush=True)
for (T, P) in ((10001, 1), (101, 100), (1, 10001), (0, 5), (5, 0), (-1, 3)):
    try:
        ctx.solve(a, b, 3, np.zeros(6), P, T); print("T=%d P=%d: ACCEPTED" % (T, P)); bad += 1
    except api.IcetError as e:
        print("T=%d P=%d: refused (%s)" % (T, P, str(e)[:90]))
print("grids with differing bits or wrongly accepted:", bad)
