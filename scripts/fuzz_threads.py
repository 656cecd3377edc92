"""Four host threads, each with its own context (and one sharing none), solving different pairs at the same time: every result carries the bits of the same solve
done alone.  (ctypes releases the GIL inside the library.)  Usage (GPU box): python scripts/fuzz_threads.py [rounds] [seed]"""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from icet_amd import api
from tests.param_sweep import draw_case, pools as make_pools

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed); pools = make_pools(); bad = 0
single = api.Context()
ctxs = [api.Context() for _ in range(4)]
for rnd in range(rounds):
    jobs = [[draw_case(rng, pools) for _ in range(5)] for _ in range(4)]
    refs = [[single.solve(a, b, rl, x0, P, T, **kw) for (a, b, T, P, kw, rl, x0) in js] for js in jobs]
    out = [[None] * 5 for _ in range(4)]; errs = []
    def work(t):
        try:
            for rep in range(3):
                for i, (a, b, T, P, kw, rl, x0) in enumerate(jobs[t]): out[t][i] = ctxs[t].solve(a, b, rl, x0, P, T, **kw)
        except Exception as e: errs.append(repr(e))
    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    nd = sum(0 if all(np.array_equal(out[t][i][k].view(np.uint32), refs[t][i][k].view(np.uint32)) for k in ("X", "pred_stds", "cov")) else 1 for t in range(4) for i in range(5))
    bad += 1 if (nd or errs) else 0
    print("round %d: 4 threads x 5 pairs x 3 repeats, differing results %d, errors %s" % (rnd, nd, errs[:2]), flush=True)
print("rounds with a difference or an error:", bad)
