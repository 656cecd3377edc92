"""Size limits: scans of 0.8 M .. 4.3 M rows (past the LDS bit table of the swap loop, past 2^20 rows per block of the point pass, past 2^22) against the oracle, and batches of
5000 and 20000 tiny pairs against single solves.  Usage (GPU box): python scripts/fuzz_sizes.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from icet_amd import api
from tests.param_sweep import run_case, pools as make_pools

pools = make_pools(); ctx = api.Context(); bad = 0
a0, b0 = pools[1]                                   # 131072 rows
rng = np.random.default_rng(3)
for rows in (800_000, 1_050_000, 2_200_000, 4_300_000):
    rep = -(-rows // len(a0))
    a = np.concatenate([a0 * np.float32(1 + 1e-4 * k) for k in range(rep)])[:rows]      # every copy a slightly different range: no ties, same directions
    b = np.concatenate([b0 * np.float32(1 + 1e-4 * k) for k in range(rep)])[:rows + 17]
    t0 = time.time()
    bits, d, r, ref, fits = run_case(ctx, np.ascontiguousarray(a), np.ascontiguousarray(b), 75, 24, dict(n=25, thresh=0.1, buff=0.1), 3, np.zeros(6, np.float32))
    ok = all(bits.values()); bad += 0 if ok else 1
    print("rows=%8d fits=%4d bits=%s dX_t=%.2e dX_r=%.2e (%.1f s with the oracle)%s" % (rows, fits, "ok" if ok else "DIFF", d[:3].max(), d[3:].max(), time.time() - t0,
          "" if ok else "  " + str({k: v for k, v in bits.items() if not v})), flush=True)
dev = torch.device("cuda", 0)
single = api.Context()
for n_pairs in (5000, 20000):
    k = 400
    starts = rng.integers(0, len(a0) - k, n_pairs)
    A = torch.from_numpy(np.ascontiguousarray(a0.T)).to(dev); B = torch.from_numpy(np.ascontiguousarray(b0.T)).to(dev)
    ld = A.shape[1]
    d1 = [(A.data_ptr() + 4 * int(s), k, ld) for s in starts]; d2 = [(B.data_ptr() + 4 * int(s), k + 3, ld) for s in starts]
    out = torch.zeros(n_pairs, 48, device=dev)
    prm = api.Params(4, 8, 16, 10, 0.1, 0.1, 0)
    t0 = time.time(); ctx.solve_batch_device(d1, d2, prm, out.data_ptr()); torch.cuda.synchronize(); t1 = time.time()
    o = out.cpu().numpy(); nd = 0
    for j in list(range(0, n_pairs, n_pairs // 40)) + [n_pairs - 1]:
        s = int(starts[j])
        r = single.solve(a0[s:s + k], b0[s:s + k + 3], 4, np.zeros(6), 8, 16, n=10)
        if not (np.array_equal(o[j, :6].view(np.uint32), r["X"].view(np.uint32)) and np.array_equal(o[j, 12:48].view(np.uint32), r["cov"].reshape(36).view(np.uint32))): nd += 1
    bad += 1 if nd else 0
    print("batch of %5d pairs x %d rows: %.1f ms, sampled pairs differing from their single solve: %d, finite: %s" % (n_pairs, k, 1e3 * (t1 - t0), nd, bool(np.isfinite(o).all())), flush=True)
print("size cases with complaints:", bad)
