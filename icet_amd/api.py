"""Host-side mirror of the reference interface for the accelerated path, on top of the C ABI.

The reference exposes exactly one operator for this path: the constructor of ``class ICET``
(/root/reference/include/icet.h:38-40, body src/icet.cpp:29-63), which *is* the solve; callers then
read the public members ``X`` and ``pred_stds`` (src/odometry.cpp:76-79, src/simpleMapMaker.cpp:119-122).
:class:`ICET` below keeps the same constructor arguments (same names, order, defaults and meaning) and
the same member names, and calls ``icet_solve`` in ``libicet_hip.so`` (include/icet_hip.h).

There is no CPU fallback: if the HIP library is missing or no GPU is usable this module raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICET_HIP_LIB") or os.path.join(_HERE, "lib", "libicet_hip.so")   # override: kernel experiments only

ICET_OK, ICET_ERR_BAD_ARG, ICET_ERR_NO_DEVICE, ICET_ERR_HIP, ICET_ERR_NOMEM, ICET_ERR_UNSUPPORTED = range(6)
_STATUS_NAMES = {0: "ICET_OK", 1: "ICET_ERR_BAD_ARG", 2: "ICET_ERR_NO_DEVICE", 3: "ICET_ERR_HIP", 4: "ICET_ERR_NOMEM", 5: "ICET_ERR_UNSUPPORTED"}
FLAG_TIMING = 1
FLAG_TRUE_SORT = 2      # non-parity extension, see include/icet_hip.h
FLAG_REJECT_MOVING = 4  # non-parity extension (moving-object rejection of the Python variant), see include/icet_hip.h
FLAG_ROUNDTRIP_SCAN2 = 16  # parity-study option: the reference's two spherical round trips of scan 2 (see include/icet_hip.h)
FLAG_DOUBLE_W = 32         # accuracy option: per-voxel W in double instead of the reference's float COD (see include/icet_hip.h)
FLAG_HALF_GAP_BOUNDS = 8  # non-parity extension (half-gap cluster buffers of the Python variant; implies TRUE_SORT), see include/icet_hip.h

# every symbol include/icet_hip.h, include/icet_nodes.h and include/icet_io.h declare
EXPORTED_SYMBOLS = ("icet_create", "icet_destroy", "icet_last_error", "icet_version", "icet_solve", "icet_solve_begin", "icet_solve_keyframe_tables", "icet_solve_end", "icet_solve_batch",
                    "icet_solve_batch_device", "icet_sync", "icet_reserve", "icet_last_timing", "icet_last_timing_iters", "icet_keep_stats", "icet_debug_fetch", "icet_debug_gn_tail", "icet_debug_pinv3", "icet_set_option", "icet_keyframe_device", "icet_register_device", "icet_keyframe_device_n", "icet_register_device_n", "icet_multi_create", "icet_multi_destroy", "icet_multi_last_error", "icet_multi_devices", "icet_multi_context",
                    "icet_multi_solve_batch", "icet_multi_solve_batch_device", "icet_multi_solve_batch_device_after", "icet_multi_solve_batch_device_async", "icet_multi_sync", "icet_multi_set_option",
                    "icet_node_create", "icet_node_destroy", "icet_node_last_error", "icet_node_push", "icet_node_push_device", "icet_node_push_many_device", "icet_node_map",
                    "icet_node_prev_scan", "icet_node_aligned", "icet_node_snail_trail", "icet_node_last_timing", "icet_stream", "icet_device",
                    "icet_load_scan", "icet_free_scan", "icet_save_scan_npy")
_NON_STATUS = ("icet_version", "icet_last_error", "icet_node_last_error", "icet_stream", "icet_device", "icet_free_scan", "icet_multi_last_error", "icet_multi_devices", "icet_multi_context")


class IcetError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__("%s%s" % (_STATUS_NAMES.get(status, str(status)), (": " + msg) if msg else ""))


class Params(C.Structure):
    _fields_ = [("runlen", C.c_int32), ("bins_phi", C.c_int32), ("bins_theta", C.c_int32), ("n", C.c_int32),
                ("thresh", C.c_float), ("buff", C.c_float), ("flags", C.c_int32)]


class NodeParams(C.Structure):
    """icet_node_params (include/icet_nodes.h)."""
    _fields_ = [("solve", Params), ("min_range", C.c_float), ("seed_x0", C.c_int32), ("trans_thresh", C.c_float), ("rot_thresh", C.c_float),
                ("map_capacity", C.c_int32), ("map_downsample", C.c_int32), ("flags", C.c_int32)]


class NodeResult(C.Structure):
    _fields_ = [("solved", C.c_int32), ("diverged", C.c_int32), ("n_kept", C.c_int64), ("X", C.c_float * 6), ("pred_stds", C.c_float * 6),
                ("pose", C.c_float * 16), ("quat", C.c_float * 4), ("map_rows", C.c_int64)]


class DevScan(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("n", C.c_int64), ("ld", C.c_int64)]


_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int32)


class Aux(C.Structure):
    _fields_ = [("cluster_bounds", _F), ("n1_raw", _I), ("has_fit", _I), ("mu1", _F), ("sigma1", _F), ("evecs1", _F), ("l_diag", _F),
                ("x_hist", _F), ("htwh", _F), ("htwdz", _F), ("n2_raw", _I), ("n2_in", _I), ("test_points", _F), ("points2", _F),
                ("points1_spherical", _F), ("point_index1", _I), ("bin_start1", _I), ("points2_spherical", _F), ("voxel2", _I), ("cond_info", _F)]


_lib = None


def load_library():
    """dlopen libicet_hip.so (built in-tree by ``__graft_entry__.build()`` / ``make -C icet_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IcetError(ICET_ERR_HIP, "HIP library not built: %s is missing (run `make -C icet_amd/csrc`); "
                                      "there is no CPU fallback for this path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.icet_version.restype = C.c_char_p
    L.icet_last_error.restype = C.c_char_p
    L.icet_last_error.argtypes = [C.c_void_p]
    L.icet_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    L.icet_destroy.argtypes = [C.c_void_p]
    L.icet_sync.argtypes = [C.c_void_p]
    L.icet_reserve.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.c_int64, C.c_int64]
    L.icet_solve.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Aux)]
    L.icet_solve_begin.argtypes = L.icet_solve.argtypes
    L.icet_solve_end.argtypes = [C.c_void_p]
    L.icet_solve_keyframe_tables.argtypes = [C.c_void_p]
    L.icet_solve_batch.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.icet_solve_batch_device.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.POINTER(DevScan), C.POINTER(DevScan), C.c_void_p, C.c_void_p]
    L.icet_last_timing.argtypes = [C.c_void_p, C.c_void_p]
    L.icet_last_timing_iters.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.icet_keep_stats.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.icet_debug_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    L.icet_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.icet_debug_gn_tail.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.icet_debug_pinv3.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.icet_keyframe_device.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.POINTER(DevScan)]
    L.icet_register_device.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.POINTER(DevScan), C.c_void_p, C.c_void_p]
    L.icet_keyframe_device_n.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.POINTER(DevScan), C.c_void_p]
    L.icet_register_device_n.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.POINTER(DevScan), C.c_void_p, C.c_void_p, C.c_void_p]
    L.icet_multi_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32]
    L.icet_multi_destroy.argtypes = [C.c_void_p]
    L.icet_multi_last_error.argtypes = [C.c_void_p]; L.icet_multi_last_error.restype = C.c_char_p
    L.icet_multi_devices.argtypes = [C.c_void_p]; L.icet_multi_devices.restype = C.c_int32
    L.icet_multi_context.argtypes = [C.c_void_p, C.c_int32]; L.icet_multi_context.restype = C.c_void_p
    L.icet_multi_solve_batch.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.icet_multi_solve_batch_device.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.POINTER(DevScan), C.POINTER(DevScan), C.c_void_p, C.c_void_p]
    L.icet_multi_solve_batch_device_after.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.POINTER(DevScan), C.POINTER(DevScan), C.c_void_p, C.c_void_p, C.c_void_p]
    L.icet_multi_solve_batch_device_async.argtypes = L.icet_multi_solve_batch_device_after.argtypes
    L.icet_multi_sync.argtypes = [C.c_void_p]
    L.icet_multi_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.icet_node_create.argtypes = [C.c_void_p, C.POINTER(NodeParams), C.POINTER(C.c_void_p)]
    L.icet_node_destroy.argtypes = [C.c_void_p]
    L.icet_node_last_error.argtypes = [C.c_void_p]; L.icet_node_last_error.restype = C.c_char_p
    L.icet_node_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(NodeResult)]
    L.icet_node_push_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(NodeResult)]
    L.icet_node_push_many_device.argtypes = [C.c_void_p, C.POINTER(DevScan), C.c_int32, C.POINTER(NodeResult)]
    L.icet_node_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.icet_node_last_timing.argtypes = [C.c_void_p, C.c_void_p]
    L.icet_node_prev_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.icet_node_aligned.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.icet_node_snail_trail.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.icet_load_scan.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int64)]
    L.icet_free_scan.argtypes = [C.POINTER(C.c_float)]; L.icet_free_scan.restype = None
    L.icet_save_scan_npy.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64]
    L.icet_stream.argtypes = [C.c_void_p]; L.icet_stream.restype = C.c_void_p
    L.icet_device.argtypes = [C.c_void_p]; L.icet_device.restype = C.c_int
    for name in EXPORTED_SYMBOLS:
        getattr(L, name)
        if name not in _NON_STATUS:
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _colmajor(scan):
    """N x 3 array-like -> float32 (3, N) C-contiguous buffer == column-major N x 3 (Eigen::MatrixXf::data())."""
    a = np.asarray(scan, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 3:
        raise IcetError(ICET_ERR_BAD_ARG, "scan must be N x 3")
    return np.ascontiguousarray(a.T)


class Context:
    """One device + stream + workspace (``icet_ctx``).  Not re-entrant; one per host thread."""

    def __init__(self, device=0, stream=None):
        L = load_library()
        h = C.c_void_p()
        st = L.icet_create(C.byref(h), int(device), C.c_void_p(stream) if stream else None)
        if st != ICET_OK:
            raise IcetError(st, "icet_create(device=%d)" % device)
        self._h = h
        self.device = device

    @classmethod
    def borrow(cls, handle, device=-1):
        """Wrap an icet_ctx* owned by someone else (e.g. icet_multi_context): never destroyed from here."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        self.device = device
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "_borrowed", False):
            self._h = None
            return
        for ref in getattr(self, "_nodes", []):          # nodes borrow this context: they go first
            nd = ref()
            if nd is not None:
                nd.close()
        self._nodes = []
        if getattr(self, "_h", None):
            load_library().icet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != ICET_OK:
            raise IcetError(st, load_library().icet_last_error(self._h).decode())

    def sync(self):
        self._check(load_library().icet_sync(self._h))

    def reserve(self, params, n_pairs, total_n1, total_n2):
        self._check(load_library().icet_reserve(self._h, C.byref(params), n_pairs, total_n1, total_n2))

    def debug_fetch(self, what, count):
        """Diagnostic: 'r' (float32, scan 1 in input order), 'bin' (uint16 per row: voxel id | literal-path flag << 14 | "r is exactly 0" << 15), 'src' (int32 scramble result), 'flags' (int32 per pair), 'lds_rank_ok' (count 1: the
        device passed the LDS-atomic order self-test of icet_create)."""
        code = {"r": 0, "bin": 1, "src": 3, "flags": 4, "rt2": 5, "lds_rank_ok": 6}[what]
        out = np.zeros(count, {0: np.float32, 1: np.uint16, 5: np.float32}.get(code, np.int32))
        self._check(load_library().icet_debug_fetch(self._h, code, out.ctypes.data, count))
        return out

    def keyframe_device(self, scan1_descs, params, d_rows_ptr=None):
        """Park the keyframe of the scans (device_ptr, n, ld) in this context (icet_keyframe_device[_n]: with d_rows_ptr -- a device int32
        array -- n is an upper bound and the actual row counts are read on the device)."""
        A = (DevScan * max(len(scan1_descs), 1))(*[DevScan(int(p), int(n), int(ld)) for (p, n, ld) in scan1_descs])
        self._check(load_library().icet_keyframe_device_n(self._h, C.byref(params), len(scan1_descs), A, C.c_void_p(d_rows_ptr) if d_rows_ptr else None))

    def register_device(self, scan2_descs, params, d_out_ptr, d_x0_ptr=None, d_rows_ptr=None):
        """Gauss-Newton loop of the scans against the parked keyframe (icet_register_device[_n])."""
        B = (DevScan * max(len(scan2_descs), 1))(*[DevScan(int(p), int(n), int(ld)) for (p, n, ld) in scan2_descs])
        self._check(load_library().icet_register_device_n(self._h, C.byref(params), len(scan2_descs), B, C.c_void_p(d_rows_ptr) if d_rows_ptr else None,
                                                          C.c_void_p(d_x0_ptr) if d_x0_ptr else None, C.c_void_p(d_out_ptr)))

    def set_option(self, name, value):
        """Launch-shape / diagnostic knob of this context (icet_set_option, include/icet_hip.h).  Launch-shape knobs leave the result
        bits alone; force_exact / guard_scale / lut_polar_quantile keep every decision but regroup float partial sums."""
        self._check(load_library().icet_set_option(self._h, name.encode(), float(value)))

    def last_timing(self):
        t = np.zeros(4, np.float32)
        self._check(load_library().icet_last_timing(self._h, t.ctypes.data))
        return dict(keyframe_ms=float(t[0]), gn_loop_ms=float(t[1]), accumulate_ms=float(t[2]), accumulate_launches=int(t[3]))

    def last_timing_iters(self, cap=64):
        """Per-iteration HIP-event times (ms) of the point-pass launches of the last ICET_FLAG_TIMING call (icet_last_timing_iters)."""
        t = np.zeros(cap, np.float32); n = C.c_int32(0)
        self._check(load_library().icet_last_timing_iters(self._h, t.ctypes.data, cap, C.byref(n)))
        return t[:n.value].copy()

    def keep_stats(self, n_pairs):
        """The keep list of the point pass after the last throughput batch (icet_keep_stats): (n_pairs, 4) int32 = mode, groups kept, list passes, lists built."""
        out = np.zeros((n_pairs, 4), np.int32)
        self._check(load_library().icet_keep_stats(self._h, n_pairs, out.ctypes.data))
        return out

    def debug_pinv3(self, mats):
        """icet_debug_pinv3 (test hook): the 3 x 3 float COD pseudo-inverse of ICET_FLAG_REFERENCE_W on the device for n matrices (n, 3, 3) -> (n, 3, 3)."""
        A = np.ascontiguousarray(mats, np.float32).reshape(-1, 9)
        out = np.zeros_like(A)
        self._check(load_library().icet_debug_pinv3(self._h, A.ctypes.data, A.shape[0], out.ctypes.data))
        return out.reshape(-1, 3, 3)

    def debug_gn_tail(self, htwh, htwdz):
        """icet_debug_gn_tail (test hook): the 6x6 tail of an iteration on the device for n (HTWH, HTWdz).  Returns dict of arrays with leading dimension n:
        cov (6, 6), pred_stds, dx, eigvals (NaN on the Cholesky route), pruned, route."""
        H = np.ascontiguousarray(htwh, np.float32).reshape(-1, 36); g = np.ascontiguousarray(htwdz, np.float32).reshape(-1, 6)
        n = H.shape[0]
        out = np.zeros((n, 56), np.float32)
        self._check(load_library().icet_debug_gn_tail(self._h, H.ctypes.data, g.ctypes.data, n, out.ctypes.data))
        return dict(cov=out[:, :36].reshape(n, 6, 6), pred_stds=out[:, 36:42], dx=out[:, 42:48], eigvals=out[:, 48:54], pruned=out[:, 54].astype(np.int32), route=out[:, 55].astype(np.int32))

    # -- single pair, host arrays ---------------------------------------------------------------
    def solve(self, scan1, scan2, runlen, X0, num_bins_phi, num_bins_theta, n=25, thresh=0.1, buff=0.1, aux=False, flags=0):
        """icet_solve.  scan1 / scan2: N x 3 (any layout numpy can view; an N x 3 array in Fortran order -- an Eigen::MatrixXf -- is
        passed without a copy).  aux=True also returns the side tables (icet_aux), among them ``points2`` (N2 x 3); aux="full" adds the
        per-point members of the reference object (points1_spherical, point_index1 / bin_start1, points2_spherical, voxel2)."""
        s1, s2 = _colmajor(scan1), _colmajor(scan2)
        p = Params(int(runlen), int(num_bins_phi), int(num_bins_theta), int(n), float(thresh), float(buff), int(flags))
        x0 = np.asarray(X0, np.float32).reshape(6).copy()
        X = np.zeros(6, np.float32); ps = np.zeros(6, np.float32); cov = np.zeros(36, np.float32)
        out = {}
        auxs = None
        if aux:
            V = int(num_bins_phi) * int(num_bins_theta); rl = max(int(runlen), 1)
            arr = dict(cluster_bounds=np.zeros((V, 6), np.float32), n1_raw=np.zeros(V, np.int32), has_fit=np.zeros(V, np.int32),
                       mu1=np.zeros((V, 3), np.float32), sigma1=np.zeros((V, 3, 3), np.float32), evecs1=np.zeros((V, 3, 3), np.float32),
                       l_diag=np.zeros((V, 3), np.float32), x_hist=np.zeros((rl, 6), np.float32), htwh=np.zeros((rl, 6, 6), np.float32),
                       htwdz=np.zeros((rl, 6), np.float32), n2_raw=np.zeros((rl, V), np.int32), n2_in=np.zeros((rl, V), np.int32),
                       test_points=np.zeros((V, 6, 3), np.float32), points2=np.zeros((3, s2.shape[1]), np.float32),
                       cond_info=np.zeros((rl, 8), np.float32))
            if aux == "full":
                arr.update(points1_spherical=np.zeros((3, s1.shape[1]), np.float32), point_index1=np.zeros(s1.shape[1], np.int32), bin_start1=np.zeros(V + 1, np.int32),
                           points2_spherical=np.zeros((3, s2.shape[1]), np.float32), voxel2=np.zeros(s2.shape[1], np.int32))
            auxs = Aux()
            for k, v in arr.items():
                setattr(auxs, k, v.ctypes.data_as(_I if v.dtype == np.int32 else _F))
            for k in ("points2", "points1_spherical", "points2_spherical"):
                if k in arr:
                    arr[k] = arr[k].T                         # (N, 3) views of the column-major buffers
            out["aux"] = arr
        st = load_library().icet_solve(self._h, C.byref(p), s1.ctypes.data, s1.shape[1], s1.shape[1], s2.ctypes.data, s2.shape[1], s2.shape[1],
                                       x0.ctypes.data, X.ctypes.data, ps.ctypes.data, cov.ctypes.data, C.byref(auxs) if auxs is not None else None)
        self._check(st)
        out.update(X=X, pred_stds=ps, cov=cov.reshape(6, 6))
        return out

    # -- batch, host arrays ------------------------------------------------------------------------
    def solve_batch(self, scans1, scans2, runlen, X0=None, num_bins_phi=24, num_bins_theta=75, n=25, thresh=0.1, buff=0.1):
        k = len(scans1)
        if len(scans2) != k:
            raise IcetError(ICET_ERR_BAD_ARG, "scans1 and scans2 differ in length")
        p = Params(int(runlen), int(num_bins_phi), int(num_bins_theta), int(n), float(thresh), float(buff), 0)
        s1 = [_colmajor(s) for s in scans1]; s2 = [_colmajor(s) for s in scans2]
        a1 = (C.c_void_p * max(k, 1))(*[s.ctypes.data for s in s1]); a2 = (C.c_void_p * max(k, 1))(*[s.ctypes.data for s in s2])
        n1 = np.array([s.shape[1] for s in s1], np.int64); n2 = np.array([s.shape[1] for s in s2], np.int64)
        x0 = None if X0 is None else np.ascontiguousarray(np.asarray(X0, np.float32).reshape(k, 6))
        X = np.zeros((k, 6), np.float32); ps = np.zeros((k, 6), np.float32); cov = np.zeros((k, 36), np.float32)
        st = load_library().icet_solve_batch(self._h, C.byref(p), k, a1, n1.ctypes.data, a2, n2.ctypes.data,
                                             x0.ctypes.data if x0 is not None else None, X.ctypes.data, ps.ctypes.data, cov.ctypes.data)
        self._check(st)
        return dict(X=X, pred_stds=ps, cov=cov.reshape(k, 6, 6))

    # -- batch, device-resident (raw device pointers; torch is only the allocator in callers) ---------
    def solve_batch_device(self, scan1_descs, scan2_descs, params, d_out_ptr, d_x0_ptr=None):
        """scan*_descs: sequences of (device_ptr, n, ld).  d_out_ptr: device pointer to n_pairs x 48 floats."""
        k = len(scan1_descs)
        A = (DevScan * max(k, 1))(*[DevScan(int(p), int(n), int(ld)) for (p, n, ld) in scan1_descs])
        B = (DevScan * max(k, 1))(*[DevScan(int(p), int(n), int(ld)) for (p, n, ld) in scan2_descs])
        st = load_library().icet_solve_batch_device(self._h, C.byref(params), k, A, B,
                                                    C.c_void_p(d_x0_ptr) if d_x0_ptr else None, C.c_void_p(d_out_ptr))
        self._check(st)


class MultiContext:
    """One context per GPU of this node (``icet_multi``): pair k of a batch runs on ``devices[k % len(devices)]``, results gathered."""

    def __init__(self, devices):
        L = load_library()
        ids = (C.c_int32 * max(len(devices), 1))(*[int(d) for d in devices])
        h = C.c_void_p()
        st = L.icet_multi_create(C.byref(h), ids, len(devices))
        if st != ICET_OK:
            raise IcetError(st, "icet_multi_create(%s)" % (list(devices),))
        self._h = h
        self.devices = list(devices)

    def close(self):
        if getattr(self, "_h", None):
            load_library().icet_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != ICET_OK:
            raise IcetError(st, load_library().icet_multi_last_error(self._h).decode())

    def solve_batch(self, scans1, scans2, runlen, X0=None, num_bins_phi=24, num_bins_theta=75, n=25, thresh=0.1, buff=0.1):
        k = len(scans1)
        p = Params(int(runlen), int(num_bins_phi), int(num_bins_theta), int(n), float(thresh), float(buff), 0)
        s1 = [_colmajor(s) for s in scans1]; s2 = [_colmajor(s) for s in scans2]
        a1 = (C.c_void_p * max(k, 1))(*[s.ctypes.data for s in s1]); a2 = (C.c_void_p * max(k, 1))(*[s.ctypes.data for s in s2])
        n1 = np.array([s.shape[1] for s in s1], np.int64); n2 = np.array([s.shape[1] for s in s2], np.int64)
        x0 = None if X0 is None else np.ascontiguousarray(np.asarray(X0, np.float32).reshape(k, 6))
        X = np.zeros((k, 6), np.float32); ps = np.zeros((k, 6), np.float32); cov = np.zeros((k, 36), np.float32)
        self._check(load_library().icet_multi_solve_batch(self._h, C.byref(p), k, a1, n1.ctypes.data, a2, n2.ctypes.data,
                                                          x0.ctypes.data if x0 is not None else None, X.ctypes.data, ps.ctypes.data, cov.ctypes.data))
        return dict(X=X, pred_stds=ps, cov=cov.reshape(k, 6, 6))

    def set_option(self, name, value):
        """"gather" (0 peer copies, 1 RCCL all-gather) or any per-context option, applied to every device (icet_multi_set_option)."""
        self._check(load_library().icet_multi_set_option(self._h, name.encode(), float(value)))

    def context_handle(self, i):
        return load_library().icet_multi_context(self._h, int(i))

    def context(self, i):
        """The context of devices[i] as a borrowed :class:`Context` (set_option / last_timing / reserve)."""
        return Context.borrow(self.context_handle(i), self.devices[i])

    def reserve(self, params, n_pairs, total_n1, total_n2):
        """Pre-size every device's workspace for its share of a batch (pairs round-robin: ceil(n_pairs / devices) each)."""
        L = load_library(); D = len(self.devices)
        for i in range(D):
            st = L.icet_reserve(C.c_void_p(self.context_handle(i)), C.byref(params), (n_pairs + D - 1) // D, (total_n1 + D - 1) // D + 1, (total_n2 + D - 1) // D + 1)
            if st != ICET_OK:
                raise IcetError(st, "icet_reserve on device entry %d" % i)

    def solve_batch_device(self, scan1_descs, scan2_descs, params, d_out_ptr, d_x0_ptr=None, producer_stream=None, asynchronous=False):
        """scan*_descs[k] = (device_ptr, n, ld) on devices[k % len(devices)]; d_out / d_x0 on devices[0].
        producer_stream: raw hipStream_t of devices[0] whose queued work (the writes of d_x0 / the scans) the solve must wait for.
        asynchronous=True returns as soon as the shares are handed to the device threads; call :meth:`sync` before reading d_out."""
        k = len(scan1_descs)
        A = (DevScan * max(k, 1))(*[DevScan(int(p), int(n), int(ld)) for (p, n, ld) in scan1_descs])
        B = (DevScan * max(k, 1))(*[DevScan(int(p), int(n), int(ld)) for (p, n, ld) in scan2_descs])
        fn = load_library().icet_multi_solve_batch_device_async if asynchronous else load_library().icet_multi_solve_batch_device_after
        self._check(fn(self._h, C.byref(params), k, A, B, C.c_void_p(d_x0_ptr) if d_x0_ptr else None, C.c_void_p(d_out_ptr),
                       C.c_void_p(producer_stream) if producer_stream else None))

    def sync(self):
        """icet_multi_sync: everything queued by asynchronous calls has completed on every device; raises the first failure."""
        self._check(load_library().icet_multi_sync(self._h))


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def euler_R(phi, theta, psi):
    """utils::R (src/utils.cpp:144-152): body-frame xyz Euler rotation; used only to rebuild the ``points2`` member."""
    c, s = np.cos, np.sin
    return np.array([
        [c(theta) * c(psi), s(psi) * c(phi) + s(phi) * s(theta) * c(psi), s(phi) * s(psi) - s(theta) * c(phi) * c(psi)],
        [-s(psi) * c(theta), c(phi) * c(psi) - s(phi) * s(theta) * s(psi), s(phi) * c(psi) + s(theta) * s(psi) * c(phi)],
        [s(theta), -s(phi) * c(theta), c(phi) * c(theta)]], dtype=np.float32)


class ICET:
    """Drop-in mirror of the reference's ``ICET`` object (include/icet.h:36-116).

    ``ICET(scan1, scan2, runlen, X0, num_bins_phi, num_bins_theta, n=25, thresh=0.1, buff=0.1)`` -- the
    constructor runs the whole registration on the GPU; read ``X`` (x, y, z, roll, pitch, yaw) and
    ``pred_stds`` afterwards, exactly as odometry_node / map_maker_node do.  Also filled:
    ``clusterBounds`` (V x 6), ``points1``, ``points2`` (scan 2 under the transform of the LAST
    iteration, i.e. X before the final update -- src/icet.cpp:375-378 precede :433), ``HTWH_i``,
    ``HTWdz_i``, ``ellipsoid1Means`` / ``ellipsoid1Covariances`` / ``ellipsoid1Alphas``, and ``cov``
    (the full 6x6 the reference keeps only as a local).
    """

    def __init__(self, scan1, scan2, runlen, X0, num_bins_phi, num_bins_theta, n=25, thresh=0.1, buff=0.1, *, device=0, side_tables=True):
        """side_tables: False = X / pred_stds only; True = every member a caller in the reference reads; "full" = also the per-point members
        points1Spherical, pointIndices1, points2Spherical, pointIndices2 (include/icet.h:79,82,95-96: several MB more per solve)."""
        self.rl, self.numBinsPhi, self.numBinsTheta, self.n, self.thresh, self.buff = runlen, num_bins_phi, num_bins_theta, n, thresh, buff
        ctx = default_context(device)
        res = ctx.solve(scan1, scan2, runlen, X0, num_bins_phi, num_bins_theta, n, thresh, buff, aux=side_tables)
        self.X = res["X"]
        self.pred_stds = res["pred_stds"]
        self.cov = res["cov"]
        self.points1 = np.asarray(scan1, np.float32)
        if side_tables:
            a = res["aux"]
            self.clusterBounds = a["cluster_bounds"]
            self.testPoints = a["test_points"].reshape(-1, 3)           # (V * 6) x 3, src/icet.cpp:41,213-231 (zeros where the reference leaves garbage)
            fit = a["has_fit"] == 1
            self.ellipsoid1Means = list(a["mu1"][fit])
            self.ellipsoid1Covariances = list(a["sigma1"][fit])
            self.ellipsoid1Alphas = [0.3] * int(fit.sum())
            self.ellipsoid2Means, self.ellipsoid2Covariances, self.ellipsoid2Alphas = [], [], []   # always empty in the reference
            self.side = a
            if runlen > 0:
                self.HTWH_i = a["htwh"][runlen - 1]
                self.HTWdz_i = a["htwdz"][runlen - 1].reshape(6, 1)
                xprev = np.asarray(X0, np.float32).reshape(6) if runlen == 1 else a["x_hist"][runlen - 2]
                self.dx = a["x_hist"][runlen - 1] - xprev
                self.points2 = a["points2"]                  # (p + t) * R of the last iteration (src/icet.cpp:375-378)
                if side_tables == "full":
                    T, P = num_bins_theta, num_bins_phi
                    self.points1Spherical = a["points1_spherical"]; self.points2Spherical = a["points2_spherical"]
                    bs, idx = a["bin_start1"], a["point_index1"]
                    # [theta][phi] -> ascending indices, as std::vector<std::vector<std::vector<int>>> (include/icet.h:95-96)
                    self.pointIndices1 = [[idx[bs[T * ph + th]:bs[T * ph + th + 1]] for ph in range(P)] for th in range(T)]
                    order = np.argsort(a["voxel2"], kind="stable"); cnt = np.bincount(a["voxel2"], minlength=T * P); st = np.concatenate([[0], np.cumsum(cnt)])
                    self.pointIndices2 = [[order[st[T * ph + th]:st[T * ph + th + 1]] for ph in range(P)] for th in range(T)]


# ---------------------------------------------------------------------------------------------------------------------
# The callers around the constructor (include/icet_nodes.h): the per-frame body of the reference's odometry_node and
# map_maker_node (src/odometry.cpp:46-98, src/simpleMapMaker.cpp:86-172), state kept in HBM between frames.
ODOMETRY_NODE = dict(runlen=7, bins_phi=24, bins_theta=75, n=25, thresh=0.1, buff=0.1, min_range=2.0, seed_x0=1,
                     trans_thresh=0.0, rot_thresh=0.0, map_capacity=0, map_downsample=0)              # src/odometry.cpp:58,73-82
MAP_MAKER_NODE = dict(runlen=12, bins_phi=24, bins_theta=75, n=25, thresh=0.1, buff=0.1, min_range=0.2, seed_x0=0,
                      trans_thresh=0.3, rot_thresh=0.3, map_capacity=600000, map_downsample=2000)      # src/simpleMapMaker.cpp:62,98,113-124,147,241-242
NODE_NO_RANGE_FILTER, NODE_ALIGNED_CLOUD, NODE_SNAIL_TRAIL = 1, 2, 4
NODE_NO_PIPELINE, NODE_SERIAL_ENQUEUE, NODE_DOUBLE_W, NODE_TIME_PHASES = 8, 16, 32, 64      # include/icet_nodes.h
SCAN_REGISTRATION_NODE = dict(runlen=7, bins_phi=24, bins_theta=75, n=25, thresh=0.1, buff=0.1, min_range=0.0, seed_x0=0,
                              trans_thresh=0.0, rot_thresh=0.0, map_capacity=0, map_downsample=0, flags=7)  # src/scanMatcher.cpp:44,55-64,76,79-84


def node_params(**kw):
    d = dict(ODOMETRY_NODE); d.update(kw)
    return NodeParams(Params(d["runlen"], d["bins_phi"], d["bins_theta"], d["n"], d["thresh"], d["buff"], 0), d["min_range"], d["seed_x0"],
                      d["trans_thresh"], d["rot_thresh"], d["map_capacity"], d["map_downsample"], d.get("flags", 0))


def _result_dict(r):
    return dict(solved=bool(r.solved), diverged=bool(r.diverged), n_kept=int(r.n_kept), X=np.array(r.X[:], np.float32),
                pred_stds=np.array(r.pred_stds[:], np.float32), pose=np.array(r.pose[:], np.float32).reshape(4, 4),
                quat=np.array(r.quat[:], np.float32), map_rows=int(r.map_rows))


class Node:
    """``icet_node``: feed lidar frames one by one; every frame after the first returns X, pred_stds and the chained pose."""

    def __init__(self, ctx=None, device=0, **kw):
        self._ctx = ctx if ctx is not None else Context(device)
        self._p = node_params(**kw)
        h = C.c_void_p()
        st = load_library().icet_node_create(self._ctx._h, C.byref(self._p), C.byref(h))
        if st != ICET_OK:
            raise IcetError(st, "icet_node_create")
        self._h = h
        import weakref
        if not hasattr(self._ctx, "_nodes"):
            self._ctx._nodes = []
        self._ctx._nodes.append(weakref.ref(self))

    def close(self):
        if getattr(self, "_h", None):
            load_library().icet_node_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, scan):
        """scan: N x 3 host array."""
        a = _colmajor(scan)
        r = NodeResult()
        st = load_library().icet_node_push(self._h, a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[1], C.byref(r))
        if st != ICET_OK:
            raise IcetError(st, "icet_node_push: " + (load_library().icet_node_last_error(self._h) or b"").decode())
        return _result_dict(r)

    def push_device(self, d_ptr, n, ld):
        """scan already in HBM on the context's device: column-major N x 3, leading dimension ld."""
        r = NodeResult()
        st = load_library().icet_node_push_device(self._h, C.c_void_p(int(d_ptr)), int(n), int(ld), C.byref(r))
        if st != ICET_OK:
            raise IcetError(st, "icet_node_push_device: " + (load_library().icet_node_last_error(self._h) or b"").decode())
        return _result_dict(r)

    def push_many_device(self, frames):
        """A burst of frames already in HBM, [(device_ptr, n, ld), ...]: icet_node_push_many_device -- pushed one after the other inside the library (one FFI call for all of them)."""
        k = len(frames)
        A = (DevScan * max(k, 1))(*[DevScan(int(p), int(n), int(ld)) for (p, n, ld) in frames])
        R = (NodeResult * max(k, 1))()
        st = load_library().icet_node_push_many_device(self._h, A, k, R)
        if st != ICET_OK:
            raise IcetError(st, "icet_node_push_many_device: " + (load_library().icet_node_last_error(self._h) or b"").decode())
        return [_result_dict(R[i]) for i in range(k)]

    def map(self):
        """``EigenQueue::getQueue()``: rows x 3, oldest first."""
        rows = C.c_int64()
        L = load_library()
        st = L.icet_node_map(self._h, None, 0, C.byref(rows))
        if st != ICET_OK:
            raise IcetError(st, "icet_node_map: " + (load_library().icet_node_last_error(self._h) or b"").decode())
        out = np.zeros((3, max(rows.value, 1)), np.float32)
        if rows.value:
            st = L.icet_node_map(self._h, out.ctypes.data_as(C.c_void_p), rows.value, C.byref(rows))
            if st != ICET_OK:
                raise IcetError(st, "icet_node_map: " + (load_library().icet_node_last_error(self._h) or b"").decode())
        return np.ascontiguousarray(out[:, :rows.value].T)

    def prev_scan(self):
        """The node's ``prev_pcl_matrix``: rows x 3."""
        rows = C.c_int64()
        L = load_library()
        st = L.icet_node_prev_scan(self._h, None, 0, C.byref(rows))
        if st != ICET_OK:
            raise IcetError(st, "icet_node_prev_scan: " + (load_library().icet_node_last_error(self._h) or b"").decode())
        out = np.zeros((3, max(rows.value, 1)), np.float32)
        if rows.value:
            st = L.icet_node_prev_scan(self._h, out.ctypes.data_as(C.c_void_p), rows.value, C.byref(rows))
            if st != ICET_OK:
                raise IcetError(st, "icet_node_prev_scan: " + (load_library().icet_node_last_error(self._h) or b"").decode())
        return np.ascontiguousarray(out[:, :rows.value].T)

    def _rows(self, fn, what):
        rows = C.c_int64()
        st = fn(self._h, None, 0, C.byref(rows))
        if st != ICET_OK:
            raise IcetError(st, what)
        out = np.zeros((3, max(rows.value, 1)), np.float32)
        if rows.value:
            st = fn(self._h, out.ctypes.data_as(C.c_void_p), rows.value, C.byref(rows))
            if st != ICET_OK:
                raise IcetError(st, what)
        return np.ascontiguousarray(out[:, :rows.value].T)

    def aligned(self):
        """``scan2_in_scan1_frame`` of the last frame (src/scanMatcher.cpp:76): rows x 3."""
        return self._rows(load_library().icet_node_aligned, "icet_node_aligned")

    def snail_trail(self):
        """``snailTrail`` (src/scanMatcher.cpp:79-84): rows x 3."""
        return self._rows(load_library().icet_node_snail_trail, "icet_node_snail_trail")

    def last_timing(self):
        t = (C.c_float * 3)()
        st = load_library().icet_node_last_timing(self._h, t)
        if st != ICET_OK:
            raise IcetError(st, "icet_node_last_timing: " + (load_library().icet_node_last_error(self._h) or b"").decode())
        return dict(filter_ms=t[0], solve_ms=t[1], map_ms=t[2])


# ---------------------------------------------------------------------------------------------------------------------
# Scan files (include/icet_io.h): what utils::loadPointCloudCSV (src/utils.cpp:12-91) and the Python side's np.load / KITTI
# readers hand to the constructor.
FMT_AUTO, FMT_NPY, FMT_OUSTER_CSV, FMT_XYZ_TSV, FMT_KITTI_BIN = range(5)


def load_scan(path, fmt=FMT_AUTO):
    """-> N x 3 float32 array (row-major view of the library's column-major buffer)."""
    L = load_library()
    p = C.POINTER(C.c_float)(); n = C.c_int64()
    st = L.icet_load_scan(os.fsencode(path), int(fmt), C.byref(p), C.byref(n))
    if st != ICET_OK:
        raise IcetError(st, "icet_load_scan(%s)" % path)
    try:
        a = np.ctypeslib.as_array(p, shape=(3, max(n.value, 1)))[:, :n.value].T.copy()
    finally:
        L.icet_free_scan(p)
    return a


def save_scan_npy(path, scan):
    a = _colmajor(scan)
    st = load_library().icet_save_scan_npy(os.fsencode(path), a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[1])
    if st != ICET_OK:
        raise IcetError(st, "icet_save_scan_npy(%s)" % path)
