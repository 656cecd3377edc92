"""Random draws from the PARAMETER space of the solve -- grid (bins_theta x bins_phi), minimum points n, thresh, buff, runlen, X0, scan sizes (random
stretches / strides of the sample scans, zero rows included, and a synthetic scan) -- and the GPU-against-oracle comparison of one draw.  Shared by
tests/test_gpu_parity.py::test_random_parameter_space and scripts/fuzz_params.py / fuzz_diag.py."""
import numpy as np


def pools():
    from tests.conftest import load_pair
    from icet_amd import lidar_sim as ls
    out = [load_pair("frame_804_805"), load_pair("sample_pc_1_2")]
    sa, sb, _ = ls.make_batch_pair(3)
    out.append((np.ascontiguousarray(sa.T.numpy()), np.ascontiguousarray(sb.T.numpy())))
    return out


# device flag -> oracle mode of its CPU twin (None: the reference's behaviour)
def _twins():
    from icet_amd import api
    from oracle import pyoracle as po
    return [(0, None), (api.FLAG_TRUE_SORT, po.TRUE_SORT), (api.FLAG_HALF_GAP_BOUNDS, po.HALF_GAP), (api.FLAG_REJECT_MOVING, po.REJECT_MOVING)]


def draw_case(rng, pools, with_flags=False):
    a0, b0 = pools[rng.integers(len(pools))]
    m = int(rng.choice([3000, 12000, 40000, a0.shape[0]]))
    if m < a0.shape[0]:
        if rng.random() < 0.5:                       # a contiguous stretch (storage order kept) ...
            s = int(rng.integers(0, a0.shape[0] - m)); a, b = a0[s:s + m], b0[s:s + m + int(rng.integers(0, 500))]
        else:                                        # ... or every k-th row
            k = a0.shape[0] // m; a, b = a0[::k], b0[int(rng.integers(0, k))::k]
    else:
        a, b = a0, b0
    T = int(rng.choice([7, 12, 25, 40, 64, 75, 90, 128, 150, 199])); P = int(rng.choice([3, 8, 11, 24, 32, 48]))
    while T * P > 10000: P = max(3, P // 2)
    kw = dict(n=int(rng.choice([3, 10, 25, 50, 120])), thresh=float(rng.choice([0.02, 0.1, 0.3, 1.0])), buff=float(rng.choice([0.0, 0.1, 0.5, 2.0])))
    runlen = int(rng.integers(1, 10))
    x0 = np.zeros(6, np.float32)
    if rng.random() < 0.5: x0 = (rng.normal(size=6) * np.array([0.2, 0.2, 0.05, 0.005, 0.005, 0.02])).astype(np.float32)
    if with_flags:                                   # (drawn last: the other draws of a seed are the same with and without)
        tw = _twins(); kw["_twin"] = tw[int(rng.integers(len(tw)))]
    return np.ascontiguousarray(a), np.ascontiguousarray(b), T, P, kw, runlen, x0


def run_case(ctx, a, b, T, P, kw, runlen, x0):
    from oracle import pyoracle as po
    kw = dict(kw)
    flag, mode = kw.pop("_twin", (0, None))
    okw = dict(kw) if mode is None else dict(kw, mode=mode)
    r = ctx.solve(a, b, runlen, x0, P, T, aux=True, flags=flag, **kw)
    ref = po.solve(a, b, x0=x0, runlen=runlen, bins_phi=P, bins_theta=T, trace=True, **okw)
    t, ax = ref["trace"], r["aux"]
    f = t["has_fit"] == 1
    bits = dict(n1_raw=np.array_equal(ax["n1_raw"], t["n1_raw"]), bounds=np.array_equal(ax["cluster_bounds"], t["bounds"]), has_fit=np.array_equal(ax["has_fit"], t["has_fit"]))
    for g, o in (("mu1", "mu1"), ("sigma1", "sigma1"), ("evecs1", "evecs1"), ("l_diag", "Ldiag")):
        x, y = ax[g][f], t[o][f]
        # a voxel whose cluster holds ONE point has a 0 / 0 covariance on both sides (has_fit stays 1, L = 0: src/icet.cpp:190-215 does not look): the NaNs must
        # sit in the same places; their sign / payload bits are not compared
        nx, ny = np.isnan(x), np.isnan(y)
        bits[g] = bool(np.array_equal(nx, ny) and np.array_equal(x.view(np.uint32)[~nx], y.view(np.uint32)[~ny]))
    act = f & (t["n1_raw"] > kw["n"]) & (t["bounds"][:, 5] > 1)
    # first iteration, per-voxel counts of scan 2: the oracle's exactly while the transform is the identity.  With X0 != 0 the device's transform is three FMAs
    # per coordinate (what Eigen's product kernel does on an FMA machine) and the oracle's is built with -ffp-contract=off: the two q differ in a last bit, and
    # a point within an ulp of a voxel edge -- about one per 500 k -- lands on the other side
    dn = ax["n2_raw"][0][act].astype(np.int64) - t["n2_raw"][0][act]
    bits["n2_raw0"] = bool((dn == 0).all()) if not x0.any() else bool((dn != 0).sum() <= 4 and np.abs(dn).max(initial=0) <= 2)
    d = np.abs(r["X"].astype(np.float64) - ref["X"])
    return bits, d, r, ref, int(f.sum())


