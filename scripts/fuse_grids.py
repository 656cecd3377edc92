import sys, numpy as np
sys.path.insert(0, "/root/repo")
from icet_amd import api
from tests.param_sweep import pools
a, b = pools()[0]
for (T, P) in ((4096, 1), (1, 4096), (2048, 2), (64, 64), (75, 24), (512, 8)):
    r = []
    for f in (0, 1):
        ctx = api.Context(); ctx.set_option("fuse_solve", f)
        r.append(ctx.solve(a, b, 4, np.array([0.02, 0, 0, 0, 0, 0.001], np.float32), P, T)); ctx.close()
    same = all(np.array_equal(r[0][k].view(np.uint32), r[1][k].view(np.uint32)) for k in ("X", "pred_stds", "cov"))
    print("T=%d P=%d fused == unfused bits: %s  X=%s" % (T, P, same, np.round(r[1]["X"], 5)))
