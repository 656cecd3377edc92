"""Diagnostic: keyframe tables and first-iteration sums of ONE bench pair, device vs oracle.  usage: diag_pair.py K [K ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po

ctx = icet_amd.Context(0)
for K in [int(a) for a in sys.argv[1:]]:
    s1, s2, _ = ls.make_batch_pair(K, device=torch.device("cuda", 0))
    a, b = s1.T.cpu().numpy(), s2.T.cpu().numpy()
    g = ctx.solve(a, b, 2, np.zeros(6), 24, 75, aux=True); ga = g["aux"]
    o = po.solve(a, b, runlen=2, trace=True); ot = o["trace"]
    print("== pair %d  n1 %d n2 %d" % (K, len(a), len(b)))
    sph = po.c2s(a)
    print("r bit-exact:", np.array_equal(ctx.debug_fetch("r", len(a)).view(np.uint32), sph[:, 0].view(np.uint32)),
          " src exact:", np.array_equal(ctx.debug_fetch("src", len(a)), po.scramble(sph[:, 0])))
    print("n1_raw equal:", np.array_equal(ga["n1_raw"], ot["n1_raw"]), " bounds equal:", np.array_equal(ga["cluster_bounds"], ot["bounds"]),
          " has_fit equal:", np.array_equal(ga["has_fit"], ot["has_fit"]))
    bad = np.nonzero((ga["cluster_bounds"] != ot["bounds"]).any(1))[0]
    for v in bad[:5]:
        print("   voxel", v, "gpu", ga["cluster_bounds"][v], "oracle", ot["bounds"][v], "n1", ga["n1_raw"][v])
    hf = ot["has_fit"].astype(bool) & ga["has_fit"].astype(bool)
    dmu = np.abs(ga["mu1"][hf] - ot["mu1"][hf]).max(1); dsg = np.abs(ga["sigma1"][hf] - ot["sigma1"][hf]).max(1)
    print("fits %d/%d  max|dmu| %.2e  max|dsigma| %.2e  Ldiag equal on %d of %d" % (ga["has_fit"].sum(), ot["has_fit"].sum(), dmu.max(), dsg.max(),
          (ga["l_diag"][hf] == ot["Ldiag"][hf]).all(1).sum(), hf.sum()))
    vox = np.nonzero(hf)[0]
    ld_bad = vox[~(ga["l_diag"][hf] == ot["Ldiag"][hf]).all(1)]
    for v in ld_bad[:6]:
        print("   L differs voxel", v, "gpu", ga["l_diag"][v], "oracle", ot["Ldiag"][v], "evals-ish sigma diag", np.diag(ot["sigma1"][v].reshape(3, 3)))
    for it in range(2):
        print(" iter %d: n2_raw equal %s n2_in equal %s |dHTWH|max %.3e rel %.2e |dX| %.2e" % (it, np.array_equal(ga["n2_raw"][it], ot["n2_raw"][it] * (ot["has_fit"] > 0)),
              np.array_equal(ga["n2_in"][it], ot["n2_in"][it]), np.abs(ga["htwh"][it] - ot["HTWH"][it]).max(),
              np.abs(ga["htwh"][it] - ot["HTWH"][it]).max() / np.abs(ot["HTWH"][it]).max(), np.abs(ga["x_hist"][it] - ot["X"][it]).max()))
        bad = np.nonzero(ga["n2_in"][it] != ot["n2_in"][it])[0]
        for v in bad[:6]:
            print("    voxel", v, "n2_in gpu", ga["n2_in"][it][v], "oracle", ot["n2_in"][it][v], "n2_raw", ga["n2_raw"][it][v], ot["n2_raw"][it][v], "used", ot["used"][it][v])
