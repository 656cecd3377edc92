"""Attribution of the device-vs-oracle differences on the WHOLE headline batch.

The device's one documented deviation from the reference is that scan 2 is never round-tripped through spherical coordinates
(/root/reference/src/icet.cpp:275, :303).  The oracle can do the same (ICET_ORACLE_SKIP_RT2) or keep the round trip only for
voxels whose scan-1 Gaussian is thin (ICET_ORACLE_RT2_THIN).  For every bench pair this script solves
    gpu   : the HIP path through the C ABI
    ref   : the unmodified oracle (shared arithmetic rule)
    skip  : the oracle with SKIP_RT2
    thin  : the oracle with RT2_THIN
and reports |gpu - ref|, |gpu - skip|, |skip - ref|, |thin - ref| per pair: if the skipped round trip is what separates device and
oracle on a pair, |gpu - skip| << |gpu - ref| ~ |skip - ref| there.  Run on the GPU box (the bench pairs are generated with the
device's RNG).  env N = pairs (default 256); DUMP=232,39 also saves those pairs (scans + device aux) under gpurun_out/."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po
from concurrent.futures import ThreadPoolExecutor

N = int(os.environ.get("N", "256"))
dump = [int(v) for v in os.environ.get("DUMP", "").split(",") if v]
dev = torch.device("cuda", 0)
pairs = [ls.make_batch_pair(k, device=dev) for k in range(N)]
h1 = [p[0].T.cpu().numpy() for p in pairs]; h2 = [p[1].T.cpu().numpy() for p in pairs]
ctx = icet_amd.Context(0)
gpu = [ctx.solve(h1[k], h2[k], 7, np.zeros(6), 24, 75, aux=(k in dump)) for k in range(N)]
gpu_rt = [ctx.solve(h1[k], h2[k], 7, np.zeros(6), 24, 75, flags=16) for k in range(N)]      # ICET_FLAG_ROUNDTRIP_SCAN2: the device WITH the two round trips
os.makedirs("gpurun_out", exist_ok=True)
for k in dump:
    np.savez_compressed("gpurun_out/bench_pair_%d.npz" % k, scan1=h1[k], scan2=h2[k], X_gpu=gpu[k]["X"], pred_stds_gpu=gpu[k]["pred_stds"], cov_gpu=gpu[k]["cov"],
                        **{"aux_" + a: v for a, v in gpu[k]["aux"].items()})


def one(k):
    return [po.solve(h1[k], h2[k], mode=m) for m in (po.SERIAL, po.SKIP_RT2, po.RT2_THIN)]


with ThreadPoolExecutor(16) as ex:
    res = list(ex.map(one, range(N)))
X = np.stack([g["X"] for g in gpu]); R = np.stack([r[0]["X"] for r in res]); S = np.stack([r[1]["X"] for r in res]); T = np.stack([r[2]["X"] for r in res])
PS = np.stack([g["pred_stds"] for g in gpu]); RPS = np.stack([r[0]["pred_stds"] for r in res]); SPS = np.stack([r[1]["pred_stds"] for r in res])


def tr(a, b):
    d = np.abs(a - b)
    return d[:, :3].max(1), d[:, 3:].max(1)


g_r, g_rr = tr(X, R); g_s, g_sr = tr(X, S); s_r, s_rr = tr(S, R); t_r, t_rr = tr(T, R)
XR = np.stack([g["X"] for g in gpu_rt]); q_r, q_rr = tr(XR, R); q_s, q_sr = tr(XR, S)
ps_gr = np.abs(PS / RPS - 1).max(1); ps_gs = np.abs(PS / SPS - 1).max(1)
np.save("gpurun_out/diag_rt2_X.npy", np.stack([X, R, S, T]))
out = {"pairs": N,
       "gpu_vs_ref_dt_max": float(g_r.max()), "gpu_vs_skip_dt_max": float(g_s.max()), "skip_vs_ref_dt_max": float(s_r.max()), "thin_vs_ref_dt_max": float(t_r.max()),
       "gpu_vs_ref_dt_median": float(np.median(g_r)), "gpu_vs_skip_dt_median": float(np.median(g_s)), "skip_vs_ref_dt_median": float(np.median(s_r)),
       "pairs_gpu_vs_ref_over_1e-4m_or_1e-5rad": [int(k) for k in np.nonzero((g_r > 1e-4) | (g_rr > 1e-5))[0]],
       "pairs_gpu_vs_skip_over_1e-4m_or_1e-5rad": [int(k) for k in np.nonzero((g_s > 1e-4) | (g_sr > 1e-5))[0]],
       "pairs_skip_vs_ref_over_1e-4m_or_1e-5rad": [int(k) for k in np.nonzero((s_r > 1e-4) | (s_rr > 1e-5))[0]],
       "pairs_thin_vs_ref_over_1e-4m_or_1e-5rad": [int(k) for k in np.nonzero((t_r > 1e-4) | (t_rr > 1e-5))[0]],
       "gpu_with_round_trips_vs_ref_dt_max": float(q_r.max()), "gpu_with_round_trips_vs_ref_dt_median": float(np.median(q_r)), "gpu_with_round_trips_vs_ref_dt_p99": float(np.quantile(q_r, 0.99)),
       "gpu_vs_ref_dt_p99": float(np.quantile(g_r, 0.99)),
       "pairs_gpu_with_round_trips_vs_ref_over_1e-4m_or_1e-5rad": [int(k) for k in np.nonzero((q_r > 1e-4) | (q_rr > 1e-5))[0]],
       "pairs_over_2e-5m: gpu_vs_ref / gpu_with_round_trips_vs_ref / gpu_vs_skip": [int((g_r > 2e-5).sum()), int((q_r > 2e-5).sum()), int((g_s > 2e-5).sum())],
       "pairs_closer_to_ref_with_round_trips / farther (by > 1e-6 m)": [int((q_r < g_r - 1e-6).sum()), int((q_r > g_r + 1e-6).sum())],
       "rel_pred_stds_gpu_vs_ref_max": float(ps_gr.max()), "rel_pred_stds_gpu_vs_skip_max": float(ps_gs.max()),
       "rel_pred_stds_gpu_vs_ref_top3": [(int(k), float(ps_gr[k])) for k in np.argsort(-ps_gr)[:3]],
       "rel_pred_stds_gpu_vs_skip_top3": [(int(k), float(ps_gs[k])) for k in np.argsort(-ps_gs)[:3]]}
print(json.dumps(out, indent=1))
print("worst pairs by |gpu - ref| (m):   pair   gpu-ref    gpu-skip   skip-ref   thin-ref   gpu(with round trips)-ref")
for k in np.argsort(-g_r)[:12]:
    print("  %3d   %.2e   %.2e   %.2e   %.2e   %.2e" % (k, g_r[k], g_s[k], s_r[k], t_r[k], q_r[k]))
print("worst pairs by |gpu(with round trips) - ref| (m):")
for k in np.argsort(-q_r)[:8]:
    print("  %3d   %.2e   %.2e   %.2e   %.2e   %.2e" % (k, g_r[k], g_s[k], s_r[k], t_r[k], q_r[k]))
print("worst pairs by |gpu - skip| (m):")
for k in np.argsort(-g_s)[:10]:
    print("  %3d   %.2e   %.2e   %.2e   %.2e   %.2e" % (k, g_r[k], g_s[k], s_r[k], t_r[k], q_r[k]))
