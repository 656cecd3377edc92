"""Condition numbers of H^T W H over the 256 bench pairs (oracle, CPU): how many pairs would leave k_gn_solve's Cholesky route for a given
"gn_cond_bound" (the route test is the Frobenius bound |A|_F |A^-1|_F).  scripts/ only (uses the oracle)."""
import sys, os
import numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po

def one(k):
    a, b, _ = ls.make_batch_pair(k)
    o = po.solve(a.T.numpy().copy(), b.T.numpy().copy(), trace=True)
    H = o["trace"]["HTWH"].astype(np.float64)
    c2 = [np.linalg.cond(h) for h in H]
    cf = [np.linalg.norm(h) * np.linalg.norm(np.linalg.inv(h)) for h in H]
    return max(c2), max(cf)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
with ThreadPoolExecutor(8) as ex:
    r = np.array(list(ex.map(one, range(N))))
for b in (1e3, 3e3, 1e4, 3e4, 1e5, 2.5e5, 1e6):
    print("bound %.1e: pairs with max-over-iterations cond_2 above it: %d, Frobenius bound above it: %d" % (b, (r[:, 0] > b).sum(), (r[:, 1] > b).sum()))
print("cond_2 median %.3g p90 %.3g max %.3g (pair %d); Frobenius median %.3g max %.3g" % (np.median(r[:, 0]), np.percentile(r[:, 0], 90), r[:, 0].max(), r[:, 0].argmax(), np.median(r[:, 1]), r[:, 1].max()))
