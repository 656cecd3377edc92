"""ctypes loader for the CPU restatement (oracle/icet_oracle.cpp) -- TEST INFRASTRUCTURE.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this module;
nothing under ``icet_amd/`` does.  It never reads /root/reference.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libicet_oracle.so")

SERIAL, POOL4, TRUE_SORT, LIBMF, SKIP_RT2, RT2_THIN, REJECT_MOVING, HALF_GAP, DEVICE_ARITH = 0, 1, 2, 4, 8, 16, 32, 64, 128
PINV3_DOUBLE = 256


class Params(C.Structure):
    _fields_ = [("runlen", C.c_int32), ("bins_phi", C.c_int32), ("bins_theta", C.c_int32), ("n", C.c_int32),
                ("thresh", C.c_float), ("buff", C.c_float), ("mode", C.c_int32)]


_I32P = C.POINTER(C.c_int32)
_F32P = C.POINTER(C.c_float)


class Trace(C.Structure):
    _fields_ = [("max_iters", C.c_int32), ("n_ub_voxels", C.c_int32),
                ("n1_raw", _I32P), ("has_fit", _I32P), ("bounds", _F32P), ("mu1", _F32P), ("sigma1", _F32P),
                ("evecs1", _F32P), ("Ldiag", _F32P), ("sigma_points", _F32P),
                ("n2_raw", _I32P), ("n2_in", _I32P), ("used", _I32P), ("mu2", _F32P), ("sigma2", _F32P),
                ("HTWH", _F32P), ("HTWdz", _F32P), ("dx", _F32P), ("X", _F32P), ("eigvals", _F32P), ("pruned", _I32P)]


def build(force=False):
    """Compile the oracle with its Makefile (g++ only; no GPU, no reference sources)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("icet_oracle.cpp", "icet_nodes_oracle.cpp", "icet_oracle.h", "smalllinalg.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.icet_oracle_solve.restype = C.c_int
        L.icet_oracle_solve.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Trace)]
        L.icet_oracle_solve_batch.restype = C.c_int
        L.icet_oracle_solve_batch.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.icet_oracle_time_pair.restype = C.c_double
        L.icet_oracle_time_pair.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        L.icet_oracle_eig_sym.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.icet_oracle_pinv.restype = C.c_int
        L.icet_oracle_pinv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.icet_oracle_gn_tail.restype = C.c_int
        L.icet_oracle_gn_tail.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int32)]
        L.icet_oracle_c2s.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.icet_oracle_scramble.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.icet_oracle_get_H.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.icet_oracle_R.argtypes = [C.c_void_p, C.c_void_p]
        L.icet_oracle_solve_signed.restype = C.c_int
        L.icet_oracle_solve_signed.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        _lib = L
    return _lib


def colmajor(scan):
    """N x 3 array-like -> float32 buffer laid out x[N] | y[N] | z[N] (Eigen::MatrixXf::data())."""
    a = np.asarray(scan, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 3
    return np.ascontiguousarray(a.T)          # shape (3, N), C-order == column-major N x 3


def make_params(runlen=7, bins_phi=24, bins_theta=75, n=25, thresh=0.1, buff=0.1, mode=SERIAL):
    return Params(runlen, bins_phi, bins_theta, n, thresh, buff, mode)


def voxel_of(sph, bins_phi=24, bins_theta=75):
    """sortSphericalCoordinates' voxel T * binPhi + binTheta of every row of c2s() output (src/icet.cpp:545-549)."""
    bt = ((sph[:, 1].astype(np.float64) / (2 * np.pi)) * bins_theta).astype(np.int64) % bins_theta
    bp = ((sph[:, 2].astype(np.float64) / np.pi) * bins_phi).astype(np.int64) % bins_phi
    return bins_theta * bp + bt


def solve(scan1, scan2, x0=None, trace=False, sign_ref=None, **kw):
    """Run the restated ICET constructor.  Returns dict(X, pred_stds, cov[, trace arrays]).
    sign_ref (DIAGNOSTIC, scripts/ only -- no test uses it since the shared arithmetic rule made the signs agree): optional
    (V, 3, 3) eigenvectors (as columns) of another implementation; the oracle's eigenvector signs are aligned with them
    (icet_oracle_solve_signed) and the number of flipped columns is returned as ``n_sign_flips``."""
    p = make_params(**kw)
    s1, s2 = colmajor(scan1), colmajor(scan2)
    x0 = np.zeros(6, np.float32) if x0 is None else np.asarray(x0, np.float32).copy()
    X = np.zeros(6, np.float32); ps = np.zeros(6, np.float32); cov = np.zeros(36, np.float32)
    out = {}
    tr = None
    if trace:
        V = p.bins_phi * p.bins_theta; it = max(p.runlen, 1)
        arr = dict(n1_raw=np.zeros(V, np.int32), has_fit=np.zeros(V, np.int32), bounds=np.zeros((V, 6), np.float32),
                   mu1=np.zeros((V, 3), np.float32), sigma1=np.zeros((V, 3, 3), np.float32), evecs1=np.zeros((V, 3, 3), np.float32),
                   Ldiag=np.zeros((V, 3), np.float32), sigma_points=np.zeros((V, 6, 3), np.float32),
                   n2_raw=np.zeros((it, V), np.int32), n2_in=np.zeros((it, V), np.int32), used=np.zeros((it, V), np.int32),
                   mu2=np.zeros((it, V, 3), np.float32), sigma2=np.zeros((it, V, 3, 3), np.float32),
                   HTWH=np.zeros((it, 6, 6), np.float32), HTWdz=np.zeros((it, 6), np.float32), dx=np.zeros((it, 6), np.float32),
                   X=np.zeros((it, 6), np.float32), eigvals=np.zeros((it, 6), np.float32), pruned=np.zeros(it, np.int32))
        tr = Trace()
        tr.max_iters = it
        for k, v in arr.items():
            setattr(tr, k, v.ctypes.data_as(_I32P if v.dtype == np.int32 else _F32P))
        out["trace"] = arr
    if sign_ref is not None:
        ref = np.ascontiguousarray(np.asarray(sign_ref, np.float32).reshape(p.bins_phi * p.bins_theta, 9))
        flips = C.c_int32(0)
        rc = lib().icet_oracle_solve_signed(C.byref(p), s1.ctypes.data, s1.shape[1], s1.shape[1], s2.ctypes.data, s2.shape[1], s2.shape[1],
                                            x0.ctypes.data, ref.ctypes.data, X.ctypes.data, ps.ctypes.data, cov.ctypes.data,
                                            C.byref(tr) if tr is not None else None, C.byref(flips))
        out["n_sign_flips"] = int(flips.value)
    else:
        rc = lib().icet_oracle_solve(C.byref(p), s1.ctypes.data, s1.shape[1], s1.shape[1], s2.ctypes.data, s2.shape[1], s2.shape[1],
                                     x0.ctypes.data, X.ctypes.data, ps.ctypes.data, cov.ctypes.data, C.byref(tr) if tr is not None else None)
    if rc:
        raise ValueError("icet_oracle_solve: bad argument (rc=%d)" % rc)
    out.update(X=X, pred_stds=ps, cov=cov.reshape(6, 6))
    if tr is not None:
        out["n_ub_voxels"] = tr.n_ub_voxels
    return out


def solve_batch(scans1, scans2, x0=None, n_threads=1, **kw):
    p = make_params(**kw)
    k = len(scans1)
    s1 = [colmajor(s) for s in scans1]; s2 = [colmajor(s) for s in scans2]
    a1 = (C.c_void_p * k)(*[s.ctypes.data for s in s1]); a2 = (C.c_void_p * k)(*[s.ctypes.data for s in s2])
    n1 = np.array([s.shape[1] for s in s1], np.int64); n2 = np.array([s.shape[1] for s in s2], np.int64)
    x0 = np.zeros((k, 6), np.float32) if x0 is None else np.ascontiguousarray(x0, np.float32)
    X = np.zeros((k, 6), np.float32); ps = np.zeros((k, 6), np.float32); cov = np.zeros((k, 36), np.float32)
    rc = lib().icet_oracle_solve_batch(C.byref(p), k, a1, n1.ctypes.data, a2, n2.ctypes.data, x0.ctypes.data,
                                       X.ctypes.data, ps.ctypes.data, cov.ctypes.data, n_threads)
    if rc:
        raise ValueError("icet_oracle_solve_batch rc=%d" % rc)
    return dict(X=X, pred_stds=ps, cov=cov.reshape(k, 6, 6))


def time_pair(scan1, scan2, reps=3, x0=None, **kw):
    p = make_params(**kw)
    s1, s2 = colmajor(scan1), colmajor(scan2)
    x0 = np.zeros(6, np.float32) if x0 is None else np.asarray(x0, np.float32).copy()
    X = np.zeros(6, np.float32)
    sec = lib().icet_oracle_time_pair(C.byref(p), s1.ctypes.data, s1.shape[1], s2.ctypes.data, s2.shape[1], x0.ctypes.data, reps, X.ctypes.data)
    return sec, X


def eig_sym(A, fixed3=False):
    A = np.ascontiguousarray(A, np.float32); n = A.shape[0]
    w = np.zeros(n, np.float32); Q = np.zeros((n, n), np.float32)
    lib().icet_oracle_eig_sym(A.ctypes.data, n, int(fixed3), w.ctypes.data, Q.ctypes.data)
    return w, Q


def pinv(A):
    A = np.ascontiguousarray(A, np.float32); r, c = A.shape
    out = np.zeros((c, r), np.float32)
    rank = lib().icet_oracle_pinv(A.ctypes.data, r, c, out.ctypes.data)
    return out, rank


def gn_tail(HTWH, HTWdz, libmf=False):
    """The 6x6 tail of one Gauss-Newton iteration (src/icet.cpp:410-430) on its own: dict(cov, pred_stds, dx, eigvals, pruned, rank)."""
    H = np.ascontiguousarray(HTWH, np.float32).reshape(36); g = np.ascontiguousarray(HTWdz, np.float32).reshape(6)
    out = np.zeros(54, np.float32); rank = C.c_int32(0)
    pruned = lib().icet_oracle_gn_tail(H.ctypes.data, g.ctypes.data, int(libmf), out.ctypes.data, C.byref(rank))
    return dict(cov=out[:36].reshape(6, 6).copy(), pred_stds=out[36:42].copy(), dx=out[42:48].copy(), eigvals=out[48:54].copy(), pruned=int(pruned), rank=int(rank.value))


def c2s(scan):
    s = colmajor(scan); n = s.shape[1]
    out = np.zeros((3, n), np.float32)
    lib().icet_oracle_c2s(s.ctypes.data, n, n, out.ctypes.data)
    return out.T.copy()


def scramble(r):
    r = np.ascontiguousarray(r, np.float32)
    src = np.zeros(r.shape[0], np.int32)
    lib().icet_oracle_scramble(r.ctypes.data, r.shape[0], src.ctypes.data)
    return src


def get_H(mu, angs):
    mu = np.asarray(mu, np.float32); angs = np.asarray(angs, np.float32); H = np.zeros(18, np.float32)
    lib().icet_oracle_get_H(mu.ctypes.data, angs.ctypes.data, H.ctypes.data)
    return H.reshape(3, 6)


def euler_R(angs):
    angs = np.asarray(angs, np.float32); R = np.zeros(9, np.float32)
    lib().icet_oracle_R(angs.ctypes.data, R.ctypes.data)
    return R.reshape(3, 3)


# ---- the callers around the constructor (icet_nodes_oracle.cpp) ------------------------------------------------------
class NodeParams(C.Structure):
    _fields_ = [("solve", Params), ("min_range", C.c_float), ("seed_x0", C.c_int32), ("trans_thresh", C.c_float), ("rot_thresh", C.c_float),
                ("map_capacity", C.c_int32), ("map_downsample", C.c_int32), ("flags", C.c_int32)]


class NodeResult(C.Structure):
    _fields_ = [("solved", C.c_int32), ("diverged", C.c_int32), ("n_kept", C.c_int64), ("X", C.c_float * 6), ("pred_stds", C.c_float * 6),
                ("pose", C.c_float * 16), ("quat", C.c_float * 4), ("map_rows", C.c_int64)]


class Node:
    """CPU restatement of the per-frame body of odometry_node / map_maker_node (src/odometry.cpp:46-98,
    src/simpleMapMaker.cpp:86-172).  Same keyword arguments as icet_amd.api.Node."""

    def __init__(self, runlen=7, bins_phi=24, bins_theta=75, n=25, thresh=0.1, buff=0.1, min_range=2.0, seed_x0=1,
                 trans_thresh=0.0, rot_thresh=0.0, map_capacity=0, map_downsample=0, flags=0, mode=SERIAL):
        L = lib()
        L.icet_oracle_node_create.restype = C.c_void_p
        L.icet_oracle_node_create.argtypes = [C.POINTER(NodeParams)]
        L.icet_oracle_node_destroy.argtypes = [C.c_void_p]
        L.icet_oracle_node_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(NodeResult)]
        L.icet_oracle_node_map.restype = C.c_int64
        L.icet_oracle_node_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        for f in (L.icet_oracle_node_aligned, L.icet_oracle_node_snail_trail):
            f.restype = C.c_int64; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        self._p = NodeParams(make_params(runlen, bins_phi, bins_theta, n, thresh, buff, mode), min_range, seed_x0, trans_thresh, rot_thresh,
                             map_capacity, map_downsample, flags)
        self._h = C.c_void_p(L.icet_oracle_node_create(C.byref(self._p)))

    def close(self):
        if getattr(self, "_h", None):
            lib().icet_oracle_node_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, scan):
        a = colmajor(scan)
        r = NodeResult()
        rc = lib().icet_oracle_node_push(self._h, a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[1], C.byref(r))
        if rc != 0:
            raise RuntimeError("icet_oracle_node_push -> %d" % rc)
        return dict(solved=bool(r.solved), diverged=bool(r.diverged), n_kept=int(r.n_kept), X=np.array(r.X[:], np.float32),
                    pred_stds=np.array(r.pred_stds[:], np.float32), pose=np.array(r.pose[:], np.float32).reshape(4, 4),
                    quat=np.array(r.quat[:], np.float32), map_rows=int(r.map_rows))

    def map(self):
        rows = lib().icet_oracle_node_map(self._h, None, 0)
        out = np.zeros((3, max(rows, 1)), np.float32)
        if rows:
            lib().icet_oracle_node_map(self._h, out.ctypes.data_as(C.c_void_p), rows)
        return np.ascontiguousarray(out[:, :rows].T)

    def _rows(self, fn):
        rows = fn(self._h, None, 0)
        out = np.zeros((3, max(rows, 1)), np.float32)
        if rows:
            fn(self._h, out.ctypes.data_as(C.c_void_p), rows)
        return np.ascontiguousarray(out[:, :rows].T)

    def aligned(self):
        return self._rows(lib().icet_oracle_node_aligned)

    def snail_trail(self):
        return self._rows(lib().icet_oracle_node_snail_trail)
