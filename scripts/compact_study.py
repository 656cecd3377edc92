"""CPU study for the keep-list of the point pass (round 6): which aligned 4-groups of scan 2 can matter after iteration 1, and how far X moves after X_1.
Uses the oracle's trace (test infrastructure) -- a scratch study, not part of the product path."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po
from icet_amd import lidar_sim as ls

T, P, n = 75, 24, 25

def euler_R(a):
    return po.euler_R(a).astype(np.float64)

def study(s1, s2, name, deltas=(0.0, 0.005, 0.01, 0.02, 0.04)):
    s1 = np.asarray(s1, np.float32); s2 = np.asarray(s2, np.float32)
    if s1.shape[0] == 3: s1 = s1.T
    if s2.shape[0] == 3: s2 = s2.T
    ref = po.solve(s1, s2, runlen=7, bins_phi=P, bins_theta=T, trace=True)
    tr = ref["trace"]
    active = (tr["has_fit"] != 0) & (tr["n1_raw"] > n) & (tr["bounds"][:, 5] > 1.0)
    Xs = np.vstack([np.zeros(6, np.float32), tr["X"]])          # X_0 .. X_7 ; iteration i evaluates at Xs[i]
    print(name, "N2", s2.shape[0], "active voxels", int(active.sum()))
    p = s2.astype(np.float64)
    def classify(X):
        R = euler_R(X[3:]); q = (p + X[:3].astype(np.float64)) @ R
        r = np.linalg.norm(q, axis=1); rho = np.hypot(q[:, 0], q[:, 1])
        th = np.arctan2(q[:, 1], q[:, 0]); th = np.where(th < 0, th + 2 * np.pi, th)
        with np.errstate(invalid="ignore", divide="ignore"):
            ph = np.arccos(q[:, 2] / r)
        ph = np.where(np.isnan(ph), 1000.0, ph)
        return th, ph, r, rho
    th1, ph1, r1, rho1 = classify(Xs[1])
    # displacement after X_1
    R1 = euler_R(Xs[1][3:])
    for i in range(2, 7):
        dR = np.linalg.norm(euler_R(Xs[i][3:]) - R1); dt = np.linalg.norm(Xs[i][:3].astype(np.float64) - Xs[1][:3])
        print("  iter %d: |dt| %.2e m  |dR|_F %.2e" % (i, dt, dR))
    bt = (th1 / (2 * np.pi) * T); bp = (ph1 / np.pi * P)
    N = len(p); G = (N + 3) // 4
    for d in deltas:
        keep = np.zeros(N, bool)
        # candidate bins: own, and neighbours whose edge is within d (angle units)
        ft = bt - np.floor(bt); fp = bp - np.floor(bp)
        wt = 2 * np.pi / T; wp = np.pi / P
        for dtb in (-1, 0, 1):
            for dpb in (-1, 0, 1):
                okt = np.ones(N, bool) if dtb == 0 else ((ft * wt < d) if dtb < 0 else ((1 - ft) * wt < d))
                okp = np.ones(N, bool) if dpb == 0 else ((fp * wp < d) if dpb < 0 else ((1 - fp) * wp < d))
                ibt = (np.floor(bt).astype(int) + dtb) % T; ibp = np.floor(bp).astype(int) % P + dpb
                valid = (ibp >= 0) & (ibp < P)
                v = T * np.clip(ibp, 0, P - 1) + ibt
                keep |= okt & okp & valid & active[v]
        small = (rho1 < 1.0) | (r1 > 2 * rho1)
        keep |= small
        kp = np.zeros(G * 4, bool); kp[:N] = keep
        kg = kp.reshape(G, 4).any(axis=1)
        # runs of kept groups
        edges = np.count_nonzero(np.diff(kg.astype(int)) != 0)
        print("  delta %.3f rad: points kept %.3f  groups kept %.3f  (always-kept %.4f)  kept runs %d" % (d, keep.mean(), kg.mean(), small.mean(), edges // 2))

if __name__ == "__main__":
    for k in (0, 1, 7, 39, 232):
        s1, s2, X = ls.make_batch_pair(k)
        study(np.asarray(s1), np.asarray(s2), "bench pair %d" % k)
    for nm in ("scans_frame_804_805.npz", "scans_sample_pc_1_2.npz"):
        f = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", nm)
        if os.path.exists(f):
            d = np.load(f); study(d["scan1"], d["scan2"], nm)
