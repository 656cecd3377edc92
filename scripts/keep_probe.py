"""Round 6: per-iteration point-pass times (HIP events, ICET_FLAG_TIMING) and keep-list statistics of a 256-pair batch under several option sets.  GPU box."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icet_amd
from icet_amd import api, lidar_sim as ls
dev = torch.device("cuda", 0)
NP = int(os.environ.get("PAIRS", "256"))
pairs = [ls.make_batch_pair(k, device=dev)[:2] for k in range(NP)]
if os.environ.get("REAL"):
    g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    base = []
    for nm in ("scans_frame_804_805.npz", "scans_sample_pc_1_2.npz"):
        dd = np.load(os.path.join(g, nm)); base.append(tuple(torch.from_numpy(np.ascontiguousarray(dd[k].T)).to(dev) for k in ("scan1", "scan2")))
    pairs = []
    for k in range(NP):
        R = torch.as_tensor(ls.real_batch_rotation(k), device=dev)
        kind = int(os.environ["REALKIND"]) if os.environ.get("REALKIND") else k % 2
        pairs.append(((R @ base[kind][0]).contiguous(), (R @ base[kind][1]).contiguous()))
SC = float(os.environ.get("SCALE2", "1"))
if SC != 1: pairs = [(a, b[:, :int(b.shape[1] * SC) // 256 * 256].contiguous()) for a, b in pairs]
ONLY = os.environ.get("ONLY")
def padded(t):
    n = t.shape[1]; l = (n + 63) // 64 * 64
    b = torch.zeros((3, l), dtype=torch.float32, device=dev); b[:, :n] = t; return b
b1 = [padded(p[0]) for p in pairs]; b2 = [padded(p[1]) for p in pairs]
d1 = [(b.data_ptr(), p[0].shape[1], b.shape[1]) for b, p in zip(b1, pairs)]; d2 = [(b.data_ptr(), p[1].shape[1], b.shape[1]) for b, p in zip(b2, pairs)]
out = torch.zeros((NP, 48), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
ctx = icet_amd.Context(0)
ref = None
sets = [dict(keep=0), dict(keep=1), dict(keep=1, keep_budget_t=100.0, keep_budget_r=1.0), dict(keep=1, keep_budget_t=0.15, keep_budget_r=0.03), dict(keep=1, keep_budget_t=0.3, keep_budget_r=0.03), dict(keep=1, keep_from=0)]
if ONLY: sets = sets[:int(ONLY)]
for extra in sys.argv[1:]:
    sets.append({k: float(v) for k, v in (kv.split("=") for kv in extra.split(","))})
for opts in sets:
    for k, v in dict(keep=-1, keep_from=1, keep_budget_t=0.08, keep_budget_r=0.008, keep_check_scale=1).items(): ctx.set_option(k, v)
    for k, v in opts.items(): ctx.set_option(k, v)
    pt = api.Params(7, 24, 75, 25, 0.1, 0.1, api.FLAG_TIMING); pp = api.Params(7, 24, 75, 25, 0.1, 0.1, 0)
    for _ in range(3): ctx.solve_batch_device(d1, d2, pp, out.data_ptr())
    ctx.sync()
    import time
    t0 = time.perf_counter()
    for _ in range(20): ctx.solve_batch_device(d1, d2, pp, out.data_ptr())
    ctx.sync(); ms = (time.perf_counter() - t0) / 20 * 1e3
    its = np.zeros(7); loop = 0.0
    for _ in range(5):
        ctx.solve_batch_device(d1, d2, pt, out.data_ptr()); its += ctx.last_timing_iters(); loop += ctx.last_timing()["gn_loop_ms"]
    res = out.cpu().numpy()
    if ref is None: ref = res
    line = "%-60s step %.3f ms  loop %.3f  acc per iter (us) %s  same bits %s" % (opts, ms, loop / 5, " ".join("%.0f" % (x * 200) for x in its), np.array_equal(ref.view(np.uint32), res.view(np.uint32)))
    if opts.get("keep"):
        st = ctx.keep_stats(NP); ng = np.array([(p[1].shape[1] + 3) // 4 for p in pairs])
        line += "  kept %.2f  list passes/pair %.2f  builds/pair %.2f" % (float(np.median(st[:, 1] / ng)), st[:, 2].mean(), st[:, 3].mean())
    print(line, flush=True)
