"""Diagnostic: per-pair |X_gpu - X_oracle| over the bench batch, with the oracle's natural eigenvector signs and with its
signs aligned to the device's (oracle/pyoracle.solve(sign_ref=...)).  Run on the GPU box.  env N = number of pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po
from concurrent.futures import ThreadPoolExecutor

N = int(os.environ.get("N", "256"))
dev = torch.device("cuda", 0)
pairs = [ls.make_batch_pair(k, device=dev) for k in range(N)]
h1 = [p[0].T.cpu().numpy() for p in pairs]; h2 = [p[1].T.cpu().numpy() for p in pairs]
ctx = icet_amd.Context(0)
gpu = [ctx.solve(h1[k], h2[k], 7, np.zeros(6), 24, 75, aux=True) for k in range(N)]
def one(k):
    nat = po.solve(h1[k], h2[k])
    ali = po.solve(h1[k], h2[k], sign_ref=gpu[k]["aux"]["evecs1"])
    return nat["X"], ali["X"], ali["n_sign_flips"], int(gpu[k]["aux"]["has_fit"].sum())
with ThreadPoolExecutor(16) as ex:
    res = list(ex.map(one, range(N)))
X = np.stack([g["X"] for g in gpu])
dn = np.abs(X - np.stack([r[0] for r in res])); da = np.abs(X - np.stack([r[1] for r in res]))
flips = np.array([r[2] for r in res]); fits = np.array([r[3] for r in res])
print("natural signs : pairs with dt > 3e-4: %d, dr > 1e-4: %d of %d; max dt %.2e" % ((dn[:, :3].max(1) > 3e-4).sum(), (dn[:, 3:].max(1) > 1e-4).sum(), N, dn[:, :3].max()))
print("aligned signs : pairs with dt > 3e-4: %d, dr > 1e-4: %d of %d; max dt %.2e max dr %.2e" % ((da[:, :3].max(1) > 3e-4).sum(), (da[:, 3:].max(1) > 1e-4).sum(), N, da[:, :3].max(), da[:, 3:].max()))
print("eigenvector columns flipped: %d of %d (%.3f %%), in %d pairs" % (flips.sum(), 3 * fits.sum(), 100.0 * flips.sum() / (3 * fits.sum()), (flips > 0).sum()))
for k in np.argsort(-da[:, :3].max(1))[:5]:
    # the oracle's own answer under a 1-ulp perturbation of scan 2 (same sign alignment): how much of the difference is the pair's sensitivity
    rng = np.random.default_rng(123); sens = np.zeros(6)
    base = res[k][1]
    for _ in range(3):
        bp = (h2[k].astype(np.float64) * (1.0 + rng.uniform(-1e-7, 1e-7, h2[k].shape))).astype(np.float32)
        sens = np.maximum(sens, np.abs(po.solve(h1[k], bp, sign_ref=gpu[k]["aux"]["evecs1"])["X"] - base))
    print("  pair %3d aligned dt %.2e dr %.2e (natural dt %.2e) flips %d | oracle 1-ulp sensitivity dt %.2e dr %.2e" % (
        k, da[k, :3].max(), da[k, 3:].max(), dn[k, :3].max(), flips[k], sens[:3].max(), sens[3:].max()))
