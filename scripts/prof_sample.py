"""Per-kernel profile driver for the reference's sample pairs (real lidar data): N device-resident solves of one fixture pair.
usage (GPU box, under rocprofv3 --kernel-trace --stats): python3 scripts/prof_sample.py frame_804_805|sample_pc_1_2 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import api
name = sys.argv[1] if len(sys.argv) > 1 else "frame_804_805"; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scans_%s.npz" % name))
dev = torch.device("cuda", 0)
def up(a):
    n = a.shape[0]; ld = (n + 63) // 64 * 64
    t = torch.zeros((3, ld), dtype=torch.float32, device=dev); t[:, :n] = torch.from_numpy(np.ascontiguousarray(a.T)).to(dev); return t, n
(t1, n1), (t2, n2) = up(fx["scan1"]), up(fx["scan2"])
ctx = icet_amd.Context(0); ctx.set_option("graph", 0)
out = torch.zeros((1, 48), dtype=torch.float32, device=dev)
p = api.Params(7, 24, 75, 25, 0.1, 0.1, 0)
torch.cuda.synchronize()
for _ in range(reps):
    ctx.solve_batch_device([(t1.data_ptr(), n1, t1.shape[1])], [(t2.data_ptr(), n2, t2.shape[1])], p, out.data_ptr()); ctx.sync()
print(name, out[0, :6].cpu().numpy())
