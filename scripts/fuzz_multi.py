"""icet_multi_* with 1-6 shards on ONE card (an id may repeat: one context and one host thread per entry), random ragged batches through the host-pointer entry and the
device-resident entries (synchronous and asynchronous): every pair carries the bits of its single solve.  Usage (GPU box): python scripts/fuzz_multi.py [batches] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from icet_amd import api
from tests.param_sweep import pools as make_pools, draw_scan_pair

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
pools = make_pools(); single = api.Context(); bad = 0
dev = torch.device("cuda", 0)
for bno in range(batches):
    shards = int(rng.integers(1, 7)); k = int(rng.choice([1, 3, 8, 17, 40]))
    T = int(rng.choice([40, 75, 128])); P = int(rng.choice([11, 24, 48])); runlen = int(rng.integers(1, 8))
    kw = dict(n=int(rng.choice([10, 25])), thresh=0.1, buff=float(rng.choice([0.0, 0.1])))
    pairs = [draw_scan_pair(rng, pools) for _ in range(k)]
    s1 = [np.ascontiguousarray(p[0]) for p in pairs]; s2 = [np.ascontiguousarray(p[1]) for p in pairs]
    x0 = (rng.normal(size=(k, 6)) * np.array([0.1, 0.1, 0.03, 0.003, 0.003, 0.01])).astype(np.float32); x0[rng.random(k) < 0.4] = 0
    m = api.MultiContext([0] * shards)
    host = m.solve_batch(s1, s2, runlen, x0, P, T, **kw)
    bufs1 = [torch.from_numpy(np.ascontiguousarray(s.T) if len(s) else np.zeros((3, 4), np.float32)).to(dev) for s in s1]
    bufs2 = [torch.from_numpy(np.ascontiguousarray(s.T) if len(s) else np.zeros((3, 4), np.float32)).to(dev) for s in s2]
    d1 = [(b.data_ptr(), len(s), b.shape[1]) for b, s in zip(bufs1, s1)]; d2 = [(b.data_ptr(), len(s), b.shape[1]) for b, s in zip(bufs2, s2)]
    prm = api.Params(runlen, P, T, kw["n"], kw["thresh"], kw["buff"], 0)
    out = torch.zeros(k, 48, device=dev); dx0 = torch.from_numpy(x0).to(dev); torch.cuda.synchronize()
    m.solve_batch_device(d1, d2, prm, out.data_ptr(), dx0.data_ptr()); o_sync = out.cpu().numpy()
    out.zero_(); torch.cuda.synchronize()
    m.solve_batch_device(d1, d2, prm, out.data_ptr(), dx0.data_ptr(), asynchronous=True); m.sync(); o_async = out.cpu().numpy()
    nd = 0
    for j in range(k):
        r = single.solve(s1[j], s2[j], runlen, x0[j], P, T, **kw)
        ref = np.concatenate([r["X"], r["pred_stds"], r["cov"].reshape(36)]).view(np.uint32)
        h = np.concatenate([host["X"][j], host["pred_stds"][j], host["cov"][j].reshape(36)]).view(np.uint32)
        if not (np.array_equal(h, ref) and np.array_equal(o_sync[j].view(np.uint32), ref) and np.array_equal(o_async[j].view(np.uint32), ref)): nd += 1
    bad += 1 if nd else 0
    print("batch %2d shards=%d pairs=%2d T=%3d P=%2d runlen=%d  %s" % (bno, shards, k, T, P, runlen, "ok" if not nd else "DIFF in %d pairs" % nd), flush=True)
    m.close()
print("batches with differing pairs:", bad)
