"""Two flags that must not change a result: ICET_FLAG_TIMING (per-iteration events; same bits as without) and, against the oracle, ICET_FLAG_ROUNDTRIP_SCAN2 over the
parameter space with random X0.  Usage (GPU box): python scripts/fuzz_flags2.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from icet_amd import api
from tests.param_sweep import draw_case, run_case, pools as make_pools

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed); pools = make_pools(); ctx = api.Context(); bad_t = bad_r = 0
for c in range(cases):
    a, b, T, P, kw, runlen, x0 = draw_case(rng, pools)
    r0 = ctx.solve(a, b, runlen, x0, P, T, **kw)
    r1 = ctx.solve(a, b, runlen, x0, P, T, flags=api.FLAG_TIMING, **kw)
    same = all(np.array_equal(r0[k].view(np.uint32), r1[k].view(np.uint32)) for k in ("X", "pred_stds", "cov"))
    bad_t += 0 if same else 1
    kw2 = dict(kw); kw2["_twin"] = (api.FLAG_ROUNDTRIP_SCAN2, None)
    bits, d, r, ref, fits = run_case(ctx, a, b, T, P, kw2, runlen, x0)
    ok = all(bits.values()); bad_r += 0 if ok else 1
    print("case %3d T=%3d P=%2d runlen=%d x0=%s timing-flag bits %s | rt2 keyframe bits %s dX_t=%.2e%s" % (c, T, P, runlen, "0" if not x0.any() else "r", "ok" if same else "DIFF", "ok" if ok else "DIFF", d[:3].max(),
          "" if ok else "  " + str({k: v for k, v in bits.items() if not v})), flush=True)
print("timing flag changed bits in", bad_t, "cases; rt2 keyframe / count differences in", bad_r)
