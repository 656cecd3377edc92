"""Regenerates tests/golden/*.npz.  Run in the BUILD container only (needs /root/reference for the
sample scans); the GPU box only ever reads the committed .npz files.

Fixtures are DATA, not code:
  scans_frame_804_805.npz   the reference's sample scan pair src/sample_data/frame_804.npy / frame_805.npy
                            (65536 x 3, float32-exact values, zero rows kept) re-saved as float32
  scans_sample_pc_1_2.npz   python/point_clouds/sample_pc_1.npy / sample_pc_2.npy (131072 x 3) as float32
  golden_<pair>.npz         outputs of the CPU oracle (oracle/icet_oracle.cpp) on that pair at the
                            BASELINE config-1 parameters: X, pred_stds, cov, per-iteration X / HTWH /
                            HTWdz / dx, keyframe table (bounds, n1, has_fit, mu1, sigma1, L) and
                            per-iteration per-voxel counts.  The reference ships no golden vectors
                            (SURVEY.md section 4), so these pin the oracle against ITSELF across
                            platforms/compilers and pin the GPU path against the oracle.
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyoracle as po  # noqa: E402

REF = "/root/reference"
PAIRS = {
    "frame_804_805": ("src/sample_data/frame_804.npy", "src/sample_data/frame_805.npy"),
    "sample_pc_1_2": ("python/point_clouds/sample_pc_1.npy", "python/point_clouds/sample_pc_2.npy"),
}


def main():
    for name, (f1, f2) in PAIRS.items():
        a = np.load(os.path.join(REF, f1)); b = np.load(os.path.join(REF, f2))
        a32 = np.ascontiguousarray(a, dtype=np.float32); b32 = np.ascontiguousarray(b, dtype=np.float32)
        if name.startswith("frame"):
            assert (a32.astype(np.float64) == a).all() and (b32.astype(np.float64) == b).all()
        np.savez_compressed(os.path.join(HERE, "scans_%s.npz" % name), scan1=a32, scan2=b32)
        o = po.solve(a32, b32, trace=True, runlen=7, bins_phi=24, bins_theta=75, n=25, thresh=0.1, buff=0.1)
        t = o["trace"]
        # the same pair with the literal expression types of the reference source (glibc float functions, sequential float sums,
        # std::hypot) instead of the shared arithmetic rule: how far the choice of rule moves the answer (oracle/icet_oracle.cpp header)
        lm = po.solve(a32, b32, runlen=7, bins_phi=24, bins_theta=75, n=25, thresh=0.1, buff=0.1, mode=po.LIBMF)
        np.savez_compressed(os.path.join(HERE, "golden_%s.npz" % name), X=o["X"], pred_stds=o["pred_stds"], cov=o["cov"],
                            X_hist=t["X"], HTWH=t["HTWH"], HTWdz=t["HTWdz"], dx=t["dx"], eigvals=t["eigvals"], pruned=t["pruned"],
                            bounds=t["bounds"], n1_raw=t["n1_raw"], has_fit=t["has_fit"], mu1=t["mu1"], sigma1=t["sigma1"],
                            Ldiag=t["Ldiag"], evecs1=t["evecs1"], n2_raw=t["n2_raw"], n2_in=t["n2_in"], used=t["used"],
                            n_ub_voxels=np.int32(o["n_ub_voxels"]), X_libmf=lm["X"], pred_stds_libmf=lm["pred_stds"])
        print(name, "X =", o["X"], "fits =", int(t["has_fit"].sum()), "ub =", o["n_ub_voxels"], "|X - X_libmf| =", np.abs(o["X"] - lm["X"]).max())


def degenerate():
    """golden_degenerate.npz: the oracle on the seeded degenerate scenes of icet_amd/lidar_sim.py (tunnel, single wall, open ground: the inputs
    ICET::checkCondition, src/icet.cpp:443-492, exists for).  The scans are regenerated from their seeds by the tests (torch CPU generator; float64
    checksums of both scans are stored so that a drifting generator is noticed); stored per scene: X, pred_stds, cov, per-iteration X / HTWH /
    HTWdz / dx / eigenvalues / pruned-axis count."""
    from icet_amd import lidar_sim as ls
    out = {}
    for name in ls.DEGENERATE_SCENES:
        a, b, _ = ls.make_degenerate_named(name)
        a = np.ascontiguousarray(a.T.numpy()); b = np.ascontiguousarray(b.T.numpy())
        o = po.solve(a, b, trace=True, runlen=7, bins_phi=24, bins_theta=75)
        t = o["trace"]
        for k, v in (("X", o["X"]), ("pred_stds", o["pred_stds"]), ("cov", o["cov"]), ("X_hist", t["X"]), ("HTWH", t["HTWH"]), ("HTWdz", t["HTWdz"]), ("dx", t["dx"]),
                     ("eigvals", t["eigvals"]), ("pruned", t["pruned"]), ("has_fit_count", np.int32(t["has_fit"].sum())),
                     ("checksum", np.array([a.shape[0], b.shape[0], a.astype(np.float64).sum(), b.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum(), np.abs(b.astype(np.float64)).sum()]))):
            out[name + "/" + k] = v
        print(name, "pruned", t["pruned"], "pred_stds", o["pred_stds"], "cond", t["eigvals"][-1, 5] / t["eigvals"][-1, 0])
    np.savez_compressed(os.path.join(HERE, "golden_degenerate.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "degenerate":
        degenerate()
    else:
        main()
        degenerate()
