"""Independent NumPy model of the reference hot path -- TEST INFRASTRUCTURE (second reading of the
same reference code, used only to cross-check oracle/icet_oracle.cpp in tests/ and when generating
golden fixtures; never imported by the product).

Written from /root/reference/src/icet.cpp + src/utils.cpp with a different structure from the C++
oracle: membership decisions (angles, bins, bounds) are made in float32 exactly as the reference's
types dictate, but every mean / covariance / small matrix product runs in float64 through
numpy.linalg (eigh, pinv).  Agreement of the two on real scans is the acceptance gate of
SURVEY.md section 8(c)(3).

Eigenvector signs are implementation-defined (quirk Q9): ``fit_scan1`` accepts ``sign_ref`` (the
oracle's eigenvectors) and flips numpy's columns to match, and reports in ``sign_sensitive`` the
voxels whose L mask would change under the opposite signs.
"""
import numpy as np

F = np.float32
TWO_PI_D = 2.0 * np.pi


def c2s(p):                                             # utils.cpp:93-119
    p = np.asarray(p, F)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    r = np.sqrt((x * x + y * y + z * z).astype(F)).astype(F)
    th = np.arctan2(y, x).astype(F)
    neg = th < 0.0
    th = np.where(neg, (th.astype(np.float64) + TWO_PI_D).astype(F), th)
    with np.errstate(invalid="ignore", divide="ignore"):
        ph = np.arccos((z / r).astype(F)).astype(F)
    out = np.stack([r, th, ph], 1)
    out[np.isnan(out)] = F(1000.0)
    return out


def s2c(s):                                             # utils.cpp:121-142
    r, th, ph = s[:, 0], s[:, 1], s[:, 2]
    x = (r * np.sin(ph) * np.cos(th)).astype(F)
    y = (r * np.sin(ph) * np.sin(th)).astype(F)
    z = (r * np.cos(ph)).astype(F)
    return np.stack([x, y, z], 1)


def euler_R(phi, theta, psi):                           # utils.cpp:144-152
    c, s = np.cos, np.sin
    return np.array([
        [c(theta) * c(psi), s(psi) * c(phi) + s(phi) * s(theta) * c(psi), s(phi) * s(psi) - s(theta) * c(phi) * c(psi)],
        [-s(psi) * c(theta), c(phi) * c(psi) - s(phi) * s(theta) * s(psi), s(phi) * c(psi) + s(theta) * s(psi) * c(phi)],
        [s(theta), -s(phi) * c(theta), c(phi) * c(theta)]], dtype=np.float64)


def get_H(mu, angs):                                    # icet.cpp:494-532
    phi, theta, psi = [float(a) for a in angs]
    c, s = np.cos, np.sin
    Jx = np.array([[0., -s(psi) * s(phi) + c(phi) * s(theta) * c(psi), c(phi) * s(psi) + s(theta) * s(phi) * c(psi)],
                   [0., -s(phi) * c(psi) - c(phi) * s(theta) * s(psi), c(phi) * c(psi) - s(theta) * s(psi) * s(phi)],
                   [0., -c(phi) * c(theta), -s(phi) * c(theta)]])
    Jy = np.array([[-s(theta) * c(psi), c(theta) * s(phi) * c(psi), -c(theta) * c(phi) * c(psi)],
                   [s(psi) * s(theta), -c(theta) * s(phi) * s(psi), c(theta) * s(psi) * c(phi)],
                   [c(theta), s(phi) * s(theta), -s(theta) * c(phi)]])
    Jz = np.array([[-c(theta) * s(psi), c(psi) * c(phi) - s(phi) * s(theta) * s(psi), c(psi) * s(phi) + s(theta) * c(phi) * s(psi)],
                   [-c(psi) * c(theta), -s(psi) * c(phi) - s(phi) * s(theta) * c(psi), -s(phi) * s(psi) + s(theta) * c(psi) * c(phi)],
                   [0., 0., 0.]])
    H = np.zeros((3, 6))
    H[:, :3] = -np.eye(3)
    H[:, 3] = Jx @ mu; H[:, 4] = Jy @ mu; H[:, 5] = Jz @ mu
    return H


def scramble_rows(sph):                                 # icet.cpp:72-83 (quirk Q3), literal loop on indices
    N = sph.shape[0]
    index = np.argsort(sph[:, 0], kind="stable").tolist()
    src = list(range(N))
    for i in range(N):
        j = index[i]
        if j != i:
            src[i], src[j] = src[j], src[i]
            index[i], index[j] = index[j], index[i]
    return sph[np.asarray(src)]


def bin_ids(sph, T, P):                                 # icet.cpp:534-554 (double arithmetic on float32 angles)
    th = sph[:, 1].astype(np.float64); ph = sph[:, 2].astype(np.float64)
    bt = ((th / TWO_PI_D) * T).astype(np.int64) % T
    bp = ((ph / np.pi) * P).astype(np.int64) % P
    return (T * bp + bt).astype(np.int64)


def bounds_az_el(theta, phi, T, P):                     # icet.cpp:136-139 (float / int -> float, * double -> float)
    azmin = F(np.float64(F(theta) / F(T)) * TWO_PI_D); azmax = F(np.float64(F(theta + 1) / F(T)) * TWO_PI_D)
    elmin = F(np.float64(F(phi) / F(P)) * np.pi); elmax = F(np.float64(F(phi + 1) / F(P)) * np.pi)
    return azmin, azmax, elmin, elmax


def find_cluster(r, n, thresh, buff):                   # icet.cpp:557-607
    thresh = F(thresh); buff = F(buff)
    local = []
    for pr in r:
        if local and abs(F(local[-1] - pr)) <= thresh:
            local.append(pr)
        else:
            if len(local) >= n:
                return F(local[0] - buff), F(local[-1] + buff)
            local = [pr]
    if len(local) >= n:
        if local[0] != 0:
            return F(local[0] - buff), F(local[-1] + buff)
        return F(0), F(0)
    return F(0), F(0)


def inside(sph, lims):                                  # icet.cpp:632-634
    return ((sph[:, 1] >= lims[0]) & (sph[:, 1] <= lims[1]) & (sph[:, 2] >= lims[2]) & (sph[:, 2] <= lims[3])
            & (sph[:, 0] >= lims[4]) & (sph[:, 0] <= lims[5]))


def fit_scan1(scan1, T, P, n, thresh, buff, sign_ref=None):
    sph = scramble_rows(c2s(scan1))
    b = bin_ids(sph, T, P)
    V = T * P
    order = np.argsort(b, kind="stable")
    starts = np.searchsorted(b[order], np.arange(V + 1))
    tab = dict(n1=np.diff(starts).astype(np.int64), bounds=np.zeros((V, 6), F), has_fit=np.zeros(V, bool),
               mu1=np.zeros((V, 3)), sigma1=np.zeros((V, 3, 3)), Vec=np.zeros((V, 3, 3)), L=np.zeros((V, 3)),
               sign_sensitive=[])
    for phi in range(P):
        for theta in range(T):
            v = T * phi + theta
            idx = order[starts[v]:starts[v + 1]]
            az0, az1, el0, el1 = bounds_az_el(theta, phi, T, P)
            if len(idx) >= n:
                sel = sph[idx]
                inner, outer = find_cluster(sel[:, 0], n, thresh, buff)
                lims = np.array([az0, az1, el0, el1, inner, outer], F)
                tab["bounds"][v] = lims
                filt = sel[inside(sel, lims)]
                if float(outer) > 0.1 and filt.size >= n:
                    cart = s2c(filt).astype(np.float64)
                    mu = cart.mean(0)
                    cen = cart - mu
                    cov = cen.T @ cen / (cart.shape[0] - 1)
                    w, Vec = np.linalg.eigh(cov)
                    if sign_ref is not None:
                        for k in range(3):
                            if np.dot(Vec[:, k], sign_ref[v][:, k]) < 0:
                                Vec[:, k] = -Vec[:, k]

                    def lmask(Vm):
                        rot = (2.0 * np.sqrt(np.maximum(w, 0)))[:, None] * Vm        # diag(2 sqrt(lambda)) * V, rows (Q9)
                        sp = np.empty((6, 3))
                        for k in range(3):
                            sp[2 * k] = mu + rot[k]; sp[2 * k + 1] = mu - rot[k]
                        ss = c2s(sp.astype(F))
                        ins = np.zeros(6, bool)
                        for j in range(6):                                           # icet.cpp:669-686, early break
                            ins[j] = bool(inside(ss[j:j + 1], lims)[0])
                            if ss[j, 0] > lims[5]:
                                break
                        return np.array([ins[0] or ins[1], ins[2] or ins[3], ins[4] or ins[5]], float)
                    Lv = lmask(Vec)
                    if any((lmask(Vec * np.array(sg)) != Lv).any() for sg in ((-1, 1, 1), (1, -1, 1), (1, 1, -1), (-1, -1, 1), (-1, 1, -1), (1, -1, -1), (-1, -1, -1))):
                        tab["sign_sensitive"].append(v)
                    tab["has_fit"][v] = True; tab["mu1"][v] = mu; tab["sigma1"][v] = cov; tab["Vec"][v] = Vec; tab["L"][v] = Lv
            else:
                tab["bounds"][v] = np.array([az0, az1, el0, el1, 0, 0], F)
    return tab


def solve(scan1, scan2, runlen=7, x0=None, bins_phi=24, bins_theta=75, n=25, thresh=0.1, buff=0.1, sign_ref=None, pinv_rcond=None):
    T, P = bins_theta, bins_phi; V = T * P
    tab = fit_scan1(scan1, T, P, n, thresh, buff, sign_ref)
    og = s2c(scramble_rows(c2s(scan2)))                                              # prepScan2, icet.cpp:254-277
    X = np.zeros(6) if x0 is None else np.asarray(x0, np.float64).copy()
    hist = dict(X=[], HTWH=[], HTWdz=[], n2_raw=[], n2_in=[], used=[])
    pred = np.zeros(6)
    cov6 = np.zeros((6, 6))
    for _ in range(runlen):
        rot = euler_R(X[3], X[4], X[5])
        p2 = ((og.astype(np.float64) + X[:3]) @ rot).astype(F)                       # icet.cpp:375-378
        sph = c2s(p2)
        b = bin_ids(sph, T, P)
        order = np.argsort(b, kind="stable")
        starts = np.searchsorted(b[order], np.arange(V + 1))
        n2 = np.diff(starts)
        HTWH = np.zeros((6, 6)); HTWdz = np.zeros(6)
        n2_in = np.zeros(V, np.int64); used = np.zeros(V, bool)
        for v in np.nonzero((n2 > n) & (tab["n1"] > n) & (tab["bounds"][:, 5] > 1))[0]:  # icet.cpp:290
            sel = sph[order[starts[v]:starts[v + 1]]]
            filt = sel[inside(sel, tab["bounds"][v])]
            n2_in[v] = filt.shape[0]
            if filt.shape[0] > n and tab["has_fit"][v]:                              # icet.cpp:302
                cart = s2c(filt).astype(np.float64)
                mu = cart.mean(0); cen = cart - mu
                cov = cen.T @ cen / (cart.shape[0] - 1)
                Rn = tab["sigma1"][v] / (tab["n1"][v] - 1) + cov / (n2[v] - 1)       # icet.cpp:315
                LUt = np.diag(tab["L"][v]) @ tab["Vec"][v]                           # L * U^T with U = V^T (Q8)
                Rp = LUt @ Rn @ LUt.T
                W = np.linalg.pinv(Rp, rcond=3 * np.finfo(F).eps)
                Hz = LUt @ get_H(mu, X[3:])
                HTWH += Hz.T @ W @ Hz
                HTWdz += Hz.T @ W @ (LUt @ mu - LUt @ tab["mu1"][v])
                used[v] = True
        cov6 = np.linalg.pinv(HTWH, rcond=6 * np.finfo(F).eps)                       # icet.cpp:410-417
        pred = np.sqrt(np.abs(np.diag(cov6)))
        w, U2 = np.linalg.eigh(HTWH)                                                 # icet.cpp:443-492
        k = 0
        with np.errstate(divide="ignore", invalid="ignore"):
            while k < 5 and abs(w[5] / w[k]) > 1e6:
                pred = pred + U2[:, k]
                k += 1
        Us = U2[:, k:]
        with np.errstate(divide="ignore"):
            dx = Us @ ((Us.T @ HTWdz) / w[k:]) if HTWH.any() else np.zeros(6)        # pinv(L2 lam U2^T) L2 U2^T HTWdz
        X = X + dx
        hist["X"].append(X.copy()); hist["HTWH"].append(HTWH); hist["HTWdz"].append(HTWdz)
        hist["n2_raw"].append(n2); hist["n2_in"].append(n2_in); hist["used"].append(used)
    return dict(X=X, pred_stds=pred, cov=cov6, table=tab, hist={k: np.array(v) for k, v in hist.items()})
