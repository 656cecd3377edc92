"""Keyframe bit-exactness over the whole bench batch: r of every row, the scramble result, per-bin counts, cluster bounds, fit flags and L
masks of the device against the oracle, pair by pair.  Run on the GPU box.  env N = number of pairs (default 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po
from concurrent.futures import ThreadPoolExecutor
N = int(os.environ.get("N", "256"))
dev = torch.device("cuda", 0)
ctx = icet_amd.Context(0)
bad = []
def oracle_side(ab):
    a, b = ab
    sph = po.c2s(a)
    return sph[:, 0].copy(), po.scramble(sph[:, 0]), po.solve(a, b, runlen=1, trace=True)["trace"]
pairs = []
for k in range(N):
    s1, s2, _ = ls.make_batch_pair(k, device=dev)
    pairs.append((s1.T.cpu().numpy(), s2.T.cpu().numpy()))
with ThreadPoolExecutor(16) as ex:
    ora = list(ex.map(oracle_side, pairs))
lmask_diff = 0; fits = 0
for k, ((a, b), (r, src, t)) in enumerate(zip(pairs, ora)):
    g = ctx.solve(a, b, 1, np.zeros(6), 24, 75, aux=True)["aux"]
    n = len(a)
    ok = (np.array_equal(ctx.debug_fetch("r", n).view(np.uint32), r.view(np.uint32)) and np.array_equal(ctx.debug_fetch("src", n), src)
          and np.array_equal(g["n1_raw"], t["n1_raw"]) and np.array_equal(g["cluster_bounds"], t["bounds"]) and np.array_equal(g["has_fit"], t["has_fit"]))
    hf = t["has_fit"].astype(bool)
    lmask_diff += int((g["l_diag"][hf] != t["Ldiag"][hf]).any(1).sum()); fits += int(hf.sum())
    if not ok: bad.append(k)
print("pairs with a keyframe difference (r / scramble / counts / bounds / fit flags): %d of %d %s" % (len(bad), N, bad[:10]))
print("L masks differing: %d of %d fitted voxels" % (lmask_diff, fits))
