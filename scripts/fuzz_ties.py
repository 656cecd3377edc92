"""Sort-key stress: scans made of a few distinct points repeated many times (every range a tie of tens of thousands), of points on one exact sphere shell, of one point only.
GPU against the oracle (keyframe bits, first-iteration counts) with the time of the solve.  Usage (GPU box): python scripts/fuzz_ties.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from icet_amd import api
from tests.param_sweep import run_case, tie_cases

ctx = api.Context(); bad = 0
cases = tie_cases()
for name, a, b in cases:
    t0 = time.time()
    bits, d, r, ref, fits = run_case(ctx, np.ascontiguousarray(a), np.ascontiguousarray(b), 75, 24, dict(n=25, thresh=0.1, buff=0.1), 3, np.zeros(6, np.float32))
    ok = all(bits.values()); bad += 0 if ok else 1
    t1 = time.time(); ctx.solve(a, b, 3, np.zeros(6), 24, 75); t2 = time.time()
    print("%-28s n1=%6d distinct r=%6d fits=%3d bits=%s gpu solve %.1f ms%s" % (name, len(a), len(np.unique(np.linalg.norm(a.astype(np.float32), axis=1))), fits, "ok" if ok else "DIFF", 1e3 * (t2 - t1),
          "" if ok else "  " + str({k: v for k, v in bits.items() if not v})), flush=True)
print("cases with differing bits:", bad)
