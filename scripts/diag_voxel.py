"""Diagnostic: which voxel's first-iteration contribution to HTWH differs between device and oracle?  Contributions are
additive over voxels, so scan 2 is restricted to the points of a voxel subset and the subset is bisected.  usage: diag_voxel.py K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po

K = int(sys.argv[1]); T, P = 75, 24
ctx = icet_amd.Context(0)
s1, s2, _ = ls.make_batch_pair(K, device=torch.device("cuda", 0))
a, b = s1.T.cpu().numpy(), s2.T.cpu().numpy()
sph = po.c2s(b).astype(np.float64)
vox = (np.floor(sph[:, 2] / np.pi * P).astype(int) % P) * T + (np.floor(sph[:, 1] / (2 * np.pi) * T).astype(int) % T)
full = po.solve(a, b, runlen=1, trace=True)["trace"]
used = np.nonzero(full["used"][0])[0]
print("used voxels:", len(used))

def diff(S):
    m = np.isin(vox, S)
    bb = b[m]
    g = ctx.solve(a, bb, 1, np.zeros(6), 24, 75, aux=True)["aux"]["htwh"][0]
    o = po.solve(a, bb, runlen=1, trace=True)["trace"]["HTWH"][0]
    return np.abs(g - o).max(), np.abs(o).max()

cands = list(used)
print("all used:", diff(cands))
per = []
for v in used:
    d, s = diff([v])
    per.append((d / max(s, 1e-30), d, s, v))
per.sort(reverse=True)
for rel, d, s, v in per[:8]:
    t = full
    print("voxel %4d rel %.2e abs %.3e scale %.3e | n2_raw %d n2_in %d n1 %d Ldiag %s | sigma2 diag %s | sigma1 diag %s" % (
        v, rel, d, s, t["n2_raw"][0][v], t["n2_in"][0][v], t["n1_raw"][v], t["Ldiag"][v], np.diag(t["sigma2"][0][v].reshape(3, 3)), np.diag(t["sigma1"][v].reshape(3, 3))))
    ev = np.linalg.eigvalsh(t["sigma2"][0][v].reshape(3, 3).astype(np.float64)); print("        eig(sigma2)", ev, " mu2-mu1", t["mu2"][0][v] - t["mu1"][v])
