#!/usr/bin/env python3
"""Bitwise A/B of two builds of the library on the same scan pairs (kernel work: "does this change alter any result bit?").
Usage (GPU box): python scripts/cmp_libs.py icet_amd/lib_exp_base/libicet_hip.so [n_pairs]
Each library is loaded in its own child process (ICET_HIP_LIB), solves the first n bench pairs as one batch and as single
pairs on a real scan pair, and dumps X | pred_stds | cov; the parent compares the dumps bit for bit."""
import os, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(out_path, n):
    sys.path.insert(0, ROOT)
    import torch
    from icet_amd import api, lidar_sim
    ctx = api.Context()
    s1, s2 = [], []
    for k in range(n):
        a, b, _ = lidar_sim.make_pair(1000 + 2 * k, 1001 + 2 * k, device="cuda")
        s1.append(a.contiguous()); s2.append(b.contiguous())
    import numpy as np
    for nm in ("frame_804_805", "sample_pc_1_2"):                     # the reference's real pairs (thousands of exact-zero rows), as given and turned by small rotations
        fx = np.load(os.path.join(ROOT, "tests", "golden", "scans_%s.npz" % nm))
        a = torch.from_numpy(np.ascontiguousarray(fx["scan1"].T)).cuda(); b = torch.from_numpy(np.ascontiguousarray(fx["scan2"].T)).cuda()
        for j in range(int(os.environ.get("CMP_REAL", "3"))):
            ang = np.random.RandomState(7000 + j).uniform(-1, 1, 3) * np.array([0.01, 0.01, 0.05]) * (j > 0)
            R = torch.as_tensor(lidar_sim.euler_R(*ang).astype(np.float32), device="cuda")
            s1.append((R @ a).contiguous()); s2.append((R @ b).contiguous())
    n = len(s1)
    d1 = [(t.data_ptr(), t.shape[1], t.shape[1]) for t in s1]; d2 = [(t.data_ptr(), t.shape[1], t.shape[1]) for t in s2]
    rows = []
    for T, P, iters in ((75, 24, 7), (150, 48, 4)):
        out = torch.zeros((n, 48), dtype=torch.float32, device="cuda")
        ctx.solve_batch_device(d1, d2, api.Params(iters, P, T, 25, 0.1, 0.1, 0), out.data_ptr())
        torch.cuda.synchronize()
        rows.append(out.cpu().numpy().reshape(-1))
    np.save(out_path, np.concatenate(rows))


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit("usage: cmp_libs.py path/to/other/libicet_hip.so [pairs]   (bitwise comparison of this build's results with another build's)")
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    other = os.path.abspath(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    d = tempfile.mkdtemp()
    outs = []
    for tag, lib in (("this", None), ("other", other)):
        env = dict(os.environ)
        if lib: env["ICET_HIP_LIB"] = lib
        o = os.path.join(d, tag + ".npy")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", o, str(n)], env=env)
        outs.append(np.load(o))
    a, b = outs
    same = a.view(np.uint32) == b.view(np.uint32)
    print("values %d, bitwise different %d, max |diff| %.3g" % (a.size, int((~same).sum()), float(np.abs(a - b).max())))
    sys.exit(0 if same.all() else 1)
