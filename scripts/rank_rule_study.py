"""CPU study (review r4, next-round item 1): does the eigenvalue rank rule the device used in rounds 2-4 (|lambda_k| > 6 eps lambda_max) agree with the
pivot rule of Eigen's CompleteOrthogonalDecomposition (|R_kk| > 6 eps max|R_ii|, oracle/smalllinalg.h cod_pinv) on H^T W H matrices whose
condition number crosses checkCondition's cutoff?  HTWHs of the degenerate scenes (icet_amd/lidar_sim.make_degenerate_pair) are rescaled along
their weakest eigenvector so that cond sweeps [3e5, 3e7].  Prints the disagreements.  Uses the oracle: scripts/ only."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po

EPS = np.float32(np.finfo(np.float32).eps)
mats = []
for kind, sigma in (("tunnel", 0.0005), ("tunnel", 0.001), ("wall", 0.001), ("wall", 0.002), ("ground", 0.0002), ("ground", 0.0005), ("ground", 0.001)):
    a, b, _ = ls.make_degenerate_pair(kind, sigma=sigma)
    o = po.solve(a.T.numpy().copy(), b.T.numpy().copy(), trace=True)
    mats += [(kind, sigma, o["trace"]["HTWH"][i].astype(np.float64), o["trace"]["HTWdz"][i]) for i in (0, 3, 6)]
n = dis = 0
conds = np.geomspace(3e5, 3e7, 400)
for kind, sigma, H, g in mats:
    w, Q = np.linalg.eigh(H)
    for c in conds:
        w2 = w.copy(); w2[0] = w[5] / c
        H2 = ((Q * w2) @ Q.T).astype(np.float32); H2 = (H2 + H2.T) / 2
        t = po.gn_tail(H2, g)
        ev = t["eigvals"]
        rank_ev = int((np.abs(ev) > np.float32(6) * EPS * np.abs(ev).max()).sum())
        n += 1
        if rank_ev != t["rank"]:
            dis += 1
            if dis <= 12:
                print("%s sigma=%g cond=%.3g: COD rank %d, eigenvalue rule %d, pruned %d, ev0/ev5 = %.3g" % (kind, sigma, c, t["rank"], rank_ev, t["pruned"], ev[0] / ev[5]))
print("%d matrices, %d rank disagreements between the COD pivot rule and the eigenvalue rule" % (n, dis))
