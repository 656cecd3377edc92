"""Diagnostic: where does the Gauss-Newton loop of ONE bench pair part ways between device and oracle?  Every iteration is replayed
on its own from the ORACLE's state before it (x0 = the oracle's X after the previous iteration), so drift does not accumulate and
any difference belongs to that iteration: scan-2 counts per voxel, the per-voxel Gaussians' effect on HTWH / HTWdz, dx.
usage: diag_loop.py K [K ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po

ctx = icet_amd.Context(0)
for K in [int(a) for a in sys.argv[1:]]:
    s1, s2, _ = ls.make_batch_pair(K, device=torch.device("cuda", 0))
    a, b = s1.T.cpu().numpy(), s2.T.cpu().numpy()
    full_o = po.solve(a, b, trace=True); full_g = ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
    print("== pair %d: final |dX| t %.2e r %.2e" % (K, np.abs(full_g["X"][:3] - full_o["X"][:3]).max(), np.abs(full_g["X"][3:] - full_o["X"][3:]).max()))
    for it in range(7):
        x0 = np.zeros(6, np.float32) if it == 0 else full_o["trace"]["X"][it - 1]
        g = ctx.solve(a, b, 1, x0, 24, 75, aux=True); ga = g["aux"]
        o = po.solve(a, b, runlen=1, x0=x0, trace=True); ot = o["trace"]
        act = (ot["has_fit"] == 1) & (ot["n1_raw"] > 25) & (ot["bounds"][:, 5] > 1)
        raw_bad = np.nonzero(act & (ga["n2_raw"][0] != ot["n2_raw"][0]))[0]
        in_bad = np.nonzero(act & (ga["n2_in"][0] != np.maximum(ot["n2_in"][0], 0)) & (ot["n2_raw"][0] > 25))[0]
        dH = np.abs(ga["htwh"][0] - ot["HTWH"][0]).max() / np.abs(ot["HTWH"][0]).max()
        dg = np.abs(ga["htwdz"][0] - ot["HTWdz"][0]).max() / max(np.abs(ot["HTWdz"][0]).max(), 1e-30)
        ddx = np.abs((g["X"] - x0) - ot["dx"][0])
        print(" iter %d (from the oracle's X): voxels with other n2_raw %s n2_in %s | rel dHTWH %.2e rel dHTWdz %.2e | |d dx| t %.2e r %.2e | drift so far t %.2e" % (
            it, list(raw_bad[:6]), list(in_bad[:6]), dH, dg, ddx[:3].max(), ddx[3:].max(), np.abs(full_g["aux"]["x_hist"][it][:3] - full_o["trace"]["X"][it][:3]).max()))
        for v in list(raw_bad[:3]) + list(in_bad[:3]):
            print("     voxel %d: n2_raw gpu %d oracle %d, n2_in gpu %d oracle %d, used %d, n1 %d, L %s" % (v, ga["n2_raw"][0][v], ot["n2_raw"][0][v], ga["n2_in"][0][v], ot["n2_in"][0][v], ot["used"][0][v], ot["n1_raw"][v], ot["Ldiag"][v]))
