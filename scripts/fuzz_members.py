"""The per-point members of the reference object (aux="full": points1Spherical, pointIndices1, points2, points2Spherical, pointIndices2) over random parameter draws and
adversarial scans: scan-1 members bit for bit the oracle's (cartesianToSpherical of the rows in the order sort + swap loop leave them; stable grouping by voxel), scan-2
members the oracle's c2s / voxel rule on the returned `points2`, and `points2` itself against a float64 evaluation of (p + t) R.  Usage (GPU box): python scripts/fuzz_members.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from icet_amd import api
from oracle import pyoracle as po
from tests.param_sweep import draw_case, draw_adversarial, pools as make_pools

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed); pools = make_pools(); ctx = api.Context(); bad = 0
for c in range(cases):
    if c % 3 == 2: a, b, T, P, kw, runlen, x0, what = draw_adversarial(rng, pools)
    else: a, b, T, P, kw, runlen, x0 = draw_case(rng, pools); what = "plain"
    if a.shape[0] > 70000: a, b = a[::2].copy(), b[::2].copy()
    g = ctx.solve(a, b, runlen, x0, P, T, aux="full", **kw); ax = g["aux"]
    base = ctx.solve(a, b, runlen, x0, P, T, aux=True, **kw)
    ok = {}
    ok["X"] = np.array_equal(g["X"].view(np.uint32), base["X"].view(np.uint32))
    sph = po.c2s(a); src = po.scramble(sph[:, 0])
    ok["points1Spherical"] = np.array_equal(ax["points1_spherical"].view(np.uint32), sph[src].view(np.uint32))
    vox = po.voxel_of(sph[src], P, T)
    ok["pointIndices1"] = np.array_equal(ax["point_index1"], np.argsort(vox, kind="stable")) and np.array_equal(np.diff(ax["bin_start1"]), np.bincount(vox, minlength=T * P))
    p2 = np.ascontiguousarray(ax["points2"])
    xprev = (x0 if runlen == 1 else ax["x_hist"][runlen - 2]).astype(np.float64)
    R = po.euler_R(xprev[3:].astype(np.float32)).reshape(3, 3).astype(np.float64)
    with np.errstate(all="ignore"):
        want = (b.astype(np.float64) + xprev[:3]) @ R
        fin = np.isfinite(want).all(axis=1) & (np.abs(want).max(axis=1) < 1e15)
        ok["points2"] = bool(np.abs(p2[fin] - want[fin]).max(initial=0) <= 3e-6 * max(1.0, np.abs(want[fin]).max(initial=0)))
    s2 = po.c2s(p2)
    ok["points2Spherical"] = np.array_equal(ax["points2_spherical"].view(np.uint32), s2.view(np.uint32))
    ok["pointIndices2"] = np.array_equal(ax["voxel2"], po.voxel_of(s2, P, T))
    good = all(ok.values()); bad += 0 if good else 1
    print("case %3d n1=%6d T=%3d P=%2d runlen=%d x0=%s [%s]  %s" % (c, a.shape[0], T, P, runlen, "0" if not x0.any() else "r", what[:40], "ok" if good else "DIFF " + str({k: v for k, v in ok.items() if not v})), flush=True)
print("cases with a differing member:", bad)
