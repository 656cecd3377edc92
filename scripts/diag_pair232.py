"""Where device and oracle part on bench pair 232, the one pair of the headline batch outside the stated tolerance (VERDICT r2, next #1:
"name the voxel").  Uses the dump scripts/diag_rt2_all.py writes with DUMP=232 (gpurun_out/bench_pair_232.npz: the pair as generated on
the GPU box, the device's X history and per-iteration per-voxel counts) and the oracle on this machine; no GPU needed.
For every iteration: |X_gpu - X_oracle|, the voxels whose raw / in-bounds scan-2 counts differ (= a DECISION differed, which can only
happen once the two X differ), and the conditioning of H^T W H."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po

k = int(sys.argv[1]) if len(sys.argv) > 1 else 232
d = np.load("gpurun_out/bench_pair_%d.npz" % k)
a, b = d["scan1"], d["scan2"]
ref = po.solve(a, b, trace=True); t = ref["trace"]
skip = po.solve(a, b, trace=True, mode=po.SKIP_RT2); ts = skip["trace"]
xh = d["aux_x_hist"]; T = 75
print("pair %d  X_gpu - X_oracle = %s" % (k, d["X_gpu"] - ref["X"]))
print("oracle pred_stds          = %s   (|dX| / pred_std = %s)" % (ref["pred_stds"], np.abs(d["X_gpu"] - ref["X"]) / ref["pred_stds"]))
print("eigenvalues of H^T W H (last iteration): %s  -> condition %.3g" % (t["eigvals"][-1], t["eigvals"][-1][-1] / t["eigvals"][-1][0]))
print("iter  |gpu-ref|   |gpu-skip|  |skip-ref|   voxels whose counts differ gpu vs oracle (voxel: n2_raw gpu/ref, n2_in gpu/ref)")
for it in range(xh.shape[0]):
    act = t["has_fit"] == 1
    dv = np.nonzero(act & ((d["aux_n2_raw"][it] != t["n2_raw"][it]) | (d["aux_n2_in"][it] != t["n2_in"][it])))[0]
    desc = ", ".join("%d(th %d, ph %d): %d/%d, %d/%d" % (v, v % T, v // T, d["aux_n2_raw"][it][v], t["n2_raw"][it][v], d["aux_n2_in"][it][v], t["n2_in"][it][v]) for v in dv[:4])
    print("%3d   %.2e   %.2e   %.2e    %d voxel(s)  %s%s" % (it, np.abs(xh[it] - t["X"][it]).max(), np.abs(xh[it] - ts["X"][it]).max(), np.abs(ts["X"][it] - t["X"][it]).max(),
                                                             dv.size, desc, " ..." if dv.size > 4 else ""))
