"""Diagnostic: how far the ORACLE's own answer moves when scan 1 / scan 2 / both are perturbed by ~1 float32 ulp (relative 1e-7),
for given bench pairs -- the yardstick for device-vs-oracle differences on ill-conditioned pairs.  usage: diag_sens.py K [K ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po
dev = torch.device("cuda", 0) if torch.cuda.is_available() else "cpu"
for K in [int(a) for a in sys.argv[1:]]:
    s1, s2, _ = ls.make_batch_pair(K, device=dev)
    a, b = s1.T.cpu().numpy(), s2.T.cpu().numpy()
    base = po.solve(a, b)["X"]
    rng = np.random.default_rng(7)
    out = {}
    for which in ("scan1", "scan2", "both"):
        dev_ = np.zeros(6)
        for _ in range(6):
            ap = (a.astype(np.float64) * (1.0 + rng.uniform(-1e-7, 1e-7, a.shape))).astype(np.float32) if which != "scan2" else a
            bp = (b.astype(np.float64) * (1.0 + rng.uniform(-1e-7, 1e-7, b.shape))).astype(np.float32) if which != "scan1" else b
            dev_ = np.maximum(dev_, np.abs(po.solve(ap, bp)["X"] - base))
        out[which] = dev_
    print("pair %d: oracle moves by  scan1-perturbed dt %.2e dr %.2e | scan2-perturbed dt %.2e dr %.2e | both dt %.2e dr %.2e" % (
        K, out["scan1"][:3].max(), out["scan1"][3:].max(), out["scan2"][:3].max(), out["scan2"][3:].max(), out["both"][:3].max(), out["both"][3:].max()))
