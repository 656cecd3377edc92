"""Diagnostic / measurement behind DESIGN.md section 7: the device against the UNMODIFIED oracle (natural eigenvector signs) over
the bench batch -- keyframe bit-exactness per pair, and the per-quantity maxima of the Gauss-Newton result that the test
tolerances are derived from.  Also how far the oracle itself moves between its two arithmetic modes (shared rule vs the literal
float-libm expression types, ICET_ORACLE_LIBMF).  Run on the GPU box.  env N = number of pairs (default 256)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po
from concurrent.futures import ThreadPoolExecutor

N = int(os.environ.get("N", "256"))
ids = [int(v) for v in os.environ["PAIRS"].split(",")] if os.environ.get("PAIRS") else list(range(N))     # PAIRS=79,190: only these bench pairs
N = len(ids)
dev = torch.device("cuda", 0)
pairs = [ls.make_batch_pair(k, device=dev) for k in ids]
h1 = [p[0].T.cpu().numpy() for p in pairs]; h2 = [p[1].T.cpu().numpy() for p in pairs]
ctx = icet_amd.Context(0)
gpu = [ctx.solve(h1[k], h2[k], 7, np.zeros(6), 24, 75, aux=True) for k in range(N)]
def one(k):
    ref = po.solve(h1[k], h2[k], trace=True)
    lm = po.solve(h1[k], h2[k], mode=po.LIBMF)
    return ref, lm["X"], lm["pred_stds"]
with ThreadPoolExecutor(16) as ex:
    res = list(ex.map(one, range(N)))
kf_bad, cols, sign_diff, lmask_diff, fits = [], 0, 0, 0, 0
dX = np.zeros((N, 6)); dps = np.zeros(N); dcov = np.zeros(N); dlib = np.zeros((N, 6)); dlps = np.zeros(N)
for k in range(N):
    ref, xl, pl = res[k]; t, ax = ref["trace"], gpu[k]["aux"]
    dlps[k] = np.abs(pl / ref["pred_stds"] - 1).max()
    f = t["has_fit"] == 1
    same = (np.array_equal(ax["n1_raw"], t["n1_raw"]) and np.array_equal(ax["cluster_bounds"], t["bounds"]) and np.array_equal(ax["has_fit"], t["has_fit"])
            and all(np.array_equal(ax[g][f].view(np.uint32), t[o][f].view(np.uint32)) for g, o in (("mu1", "mu1"), ("sigma1", "sigma1"), ("evecs1", "evecs1"), ("l_diag", "Ldiag"))))
    if not same: kf_bad.append(k)
    dots = np.einsum("vij,vij->vj", ax["evecs1"][f], t["evecs1"][f])
    cols += dots.size; sign_diff += int((dots < 0).sum()); lmask_diff += int((ax["l_diag"][f] != t["Ldiag"][f]).any(1).sum()); fits += int(f.sum())
    dX[k] = np.abs(gpu[k]["X"] - ref["X"]); dlib[k] = np.abs(ref["X"] - xl)
    dps[k] = np.abs(gpu[k]["pred_stds"] / ref["pred_stds"] - 1).max()
    d = np.sqrt(np.abs(np.diag(ref["cov"]))); dcov[k] = (np.abs(gpu[k]["cov"] - ref["cov"]) / np.outer(d, d)).max()
out = {
    "pairs": N, "pairs_with_any_keyframe_bit_difference": len(kf_bad), "first_such": kf_bad[:8],
    "eigenvector_columns": cols, "columns_with_opposite_sign": sign_diff, "fitted_voxels": fits, "L_masks_differing": lmask_diff,
    "dX_t_max": float(dX[:, :3].max()), "dX_t_median": float(np.median(dX[:, :3].max(1))), "dX_t_p99": float(np.quantile(dX[:, :3].max(1), 0.99)),
    "dX_r_max": float(dX[:, 3:].max()), "dX_r_median": float(np.median(dX[:, 3:].max(1))), "dX_r_p99": float(np.quantile(dX[:, 3:].max(1), 0.99)),
    "rel_pred_stds_max": float(dps.max()), "rel_cov_max": float(dcov.max()),
    "pairs_over_1e-4m_or_1e-5rad": int(((dX[:, :3].max(1) > 1e-4) | (dX[:, 3:].max(1) > 1e-5)).sum()),
    "pairs_over_3e-4m_or_1e-4rad": int(((dX[:, :3].max(1) > 3e-4) | (dX[:, 3:].max(1) > 1e-4)).sum()),
    "oracle_rule_vs_libmf_dX_t_max": float(dlib[:, :3].max()), "oracle_rule_vs_libmf_dX_t_median": float(np.median(dlib[:, :3].max(1))),
    "oracle_rule_vs_libmf_pairs_over_3e-4m": int((dlib[:, :3].max(1) > 3e-4).sum()),
    "oracle_rule_vs_libmf_rel_pred_stds_max": float(dlps.max()), "oracle_rule_vs_libmf_rel_pred_stds_p99": float(np.quantile(dlps, 0.99)),
    "rel_pred_stds_p99": float(np.quantile(dps, 0.99)), "rel_cov_p99": float(np.quantile(dcov, 0.99)),
    "rel_pred_stds_top5": [float(v) for v in np.sort(dps)[-5:]], "rel_pred_stds_top5_pairs": [int(k) for k in np.argsort(dps)[-5:]],
}
order = np.argsort(-dX[:, :3].max(1))
out["dX_t_top5"] = [float(dX[k, :3].max()) for k in order[:5]]; out["dX_t_top5_pairs"] = [int(k) for k in order[:5]]
keep = np.ones(N, bool); keep[order[:1]] = False       # without the single worst pair (reported below with the oracle's own sensitivity)
out["without_worst_pair"] = {"dX_t_max": float(dX[keep, :3].max()), "dX_r_max": float(dX[keep, 3:].max()), "rel_pred_stds_max": float(dps[keep].max()), "rel_cov_max": float(dcov[keep].max())}
print(json.dumps(out, indent=1))
worst = np.argsort(-dX[:, :3].max(1))[:6]
for k in worst:
    rng = np.random.default_rng(123); sens = np.zeros(6); base = res[k][0]["X"]
    for _ in range(3):
        bp = (h2[k].astype(np.float64) * (1.0 + rng.uniform(-1e-7, 1e-7, h2[k].shape))).astype(np.float32)
        sens = np.maximum(sens, np.abs(po.solve(h1[k], bp)["X"] - base))
    print("  pair %3d dt %.2e dr %.2e | oracle 1-ulp(scan 2) sensitivity dt %.2e dr %.2e | rule-vs-libmf dt %.2e" % (
        k, dX[k, :3].max(), dX[k, 3:].max(), sens[:3].max(), sens[3:].max(), dlib[k, :3].max()))
