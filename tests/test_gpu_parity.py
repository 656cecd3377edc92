"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (icet_amd.api -> libicet_hip.so), against
the CPU oracle and the committed golden fixtures.  Nothing here reads /root/reference.

Device and oracle follow ONE arithmetic rule for everything that feeds the scan-1 Gaussians (correctly rounded
transcendentals, exact sums: oracle/icet_oracle.cpp header, icet_amd/csrc/icet_device_common.h), so the comparison is made
against the UNMODIFIED oracle -- natural eigenvector signs, nothing borrowed from the device:
  * the whole keyframe table is bit-exact: per-row r and voxel, the scramble, per-bin counts, cluster bounds, has_fit,
    mu1, sigma1, the eigenvectors (signs included) and the L masks;
  * the Gauss-Newton loop differs in two documented places -- the device does not round-trip scan 2 through spherical
    coordinates (<= 2 ulp per point) and forms the per-voxel moments in one pass about mu1 -- so X, pred_stds and cov agree to
    float tolerance.  MEASURED over the whole 256-pair bench batch (scripts/diag_batch_vs_oracle.py -> profiles/r03_parity_batch.txt,
    scripts/diag_rt2_all.py -> profiles/r03_diag_rt2_all.txt): |dX_t| median 4.4e-7 m, p99 7.5e-5; 255 of 256 pairs within SURVEY
    8(c)'s starting values (1e-4 m, 1e-5 rad), max 9.6e-5 m / 4.7e-6 rad among them; pred_stds within 1.2 %, cov within 2.5 %.
    What is left on those 255 pairs is the last bit of a single-ring voxel's mean (one ulp of one 27-point voxel's mu2 moves X by 4e-5 m inside the oracle); the
    skipped round trip is one of the things that flip it: against the oracle run with the same skip (ICET_ORACLE_SKIP_RT2) the six worst pairs drop from 6e-5..1e-4 m to
    3e-7..4e-6 m and the pred_stds outlier (pair 39, 1.2 %) disappears, but restoring the trips on the device (ICET_FLAG_ROUNDTRIP_SCAN2, test_scan2_round_trip_option)
    closes only pair 18 and opens 189 (profiles/r03_diag_rt2_all.txt).  The one
    exception, pair 232 (9.2e-4 m), is not explained by it (1.1e-3 m against the skipping oracle): the oracle's own answer on that
    pair moves by 1.8e-3 m under a 1-ulp perturbation of scan 2.  The bounds below are 2x the measured maxima -- no sign alignment --
    and test_many_pairs_parity_natural_signs holds them on EVERY pair of the batch, allowing at most one exception, which must lie
    within 1x the oracle's own 1-ulp sensitivity computed in the test.
"""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_T, TOL_R = 2e-4, 2e-5              # 2 x (8.1e-5 m, 9.7e-6 rad): measured maxima over the 256-pair batch without its one sensitivity-explained pair
RTOL_STD_LOOSE, RTOL_COV_LOOSE = 2.5e-2, 5e-2   # NAMED exception only: bench pair 39 (1.2 % / 2.5 %: the last bit of a single-ring voxel's mean)
RTOL_STD, RTOL_COV = 2e-3, 4e-3         # everything else: 2 x the largest value measured over EVERY _check_solution call of this file outside pair 39
                                        # (0.10 % / 0.20 %; gpurun_out/parity_margins.txt of round 4) -- the blanket 2.5 % / 5 % of round 3 is gone (review r3, weak 1e)
RTOL_STD_P99, RTOL_COV_P99 = 1.5e-3, 3e-3      # the 256 bench pairs: 2 x their p99 (0.07 % / 0.14 %, profiles/r03_parity_batch.txt)

_MARGINS = []                           # (test, |dX_t|, |dX_r|, rel pred_stds, rel cov) of every _check_solution call: written to gpurun_out/parity_margins.txt at exit


def _record(tag, res, ref):
    try:
        d = np.sqrt(np.abs(np.diag(ref["cov"]))) + 1e-30
        with np.errstate(all="ignore"):
            rs = float(np.nanmax(np.abs(res["pred_stds"] / np.where(ref["pred_stds"] == 0, np.nan, ref["pred_stds"]) - 1))) if np.any(ref["pred_stds"] != 0) else 0.0
        _MARGINS.append((tag, float(np.abs(res["X"][:3] - ref["X"][:3]).max()), float(np.abs(res["X"][3:] - ref["X"][3:]).max()), rs,
                         float((np.abs(res["cov"] - ref["cov"]) / np.outer(d, d)).max())))
    except Exception:
        pass


@pytest.fixture(scope="module", autouse=True)
def _write_margins():
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if _MARGINS and os.path.isdir(out):
        with open(os.path.join(out, "parity_margins.txt"), "w") as f:
            f.write("# test  |dX_t| m  |dX_r| rad  rel pred_stds  rel cov (every _check_solution call of tests/test_gpu_parity.py; bounds: %g m %g rad %g %g)\n" % (TOL_T, TOL_R, RTOL_STD, RTOL_COV))
            for m in _MARGINS:
                f.write("%s %.3g %.3g %.3g %.3g\n" % m)


def oracle_sensitivity(a, b, trials=32, scan1_too=False, x0=None, **kw):
    """How far the ORACLE's own answer moves when scan 2 (and, with scan1_too, scan 1) is perturbed by ~1 float32 ulp
    (relative 1e-7): the Gauss-Newton loop re-bins every iteration, so a point flipping across a voxel edge can move X by
    far more than rounding would; a 1-ulp change in scan 1 also reaches the per-voxel covariances whose smallest eigenvalues
    set weights of 1e7 (measured on the bench batch: typically 2e-4 .. 1e-3 m).  Used to calibrate the tolerance on
    ill-conditioned pairs.  32 trials (round 6; review r5: three trials give a LOWER bound of a heavy-tailed spread), solved on the host's
    cores in parallel; returns the per-component maximum, `oracle_sensitivity.last` also holds the 90 % quantile.  Users bound by 1.5 x the maximum."""
    from oracle import pyoracle as po
    base = po.solve(a, b, x0=x0, **kw)["X"]
    rng = np.random.default_rng(123)
    A, B = [], []
    for _ in range(trials):
        A.append((a.astype(np.float64) * (1.0 + rng.uniform(-1e-7, 1e-7, a.shape))).astype(np.float32) if scan1_too else a)
        B.append((b.astype(np.float64) * (1.0 + rng.uniform(-1e-7, 1e-7, b.shape))).astype(np.float32))
    X0 = None if x0 is None else np.tile(np.asarray(x0, np.float32).reshape(1, 6), (trials, 1))
    res = po.solve_batch(A, B, x0=X0, n_threads=max(1, min(16, os.cpu_count() or 1)), **kw)
    dev = np.abs(res["X"].astype(np.float64) - base)
    oracle_sensitivity.last = dict(max=dev.max(0), p90=np.quantile(dev, 0.9, axis=0), trials=trials)
    return dev.max(0)


HATCH = 1.5          # a bound "within the oracle's own 1-ulp spread" means: within HATCH x the maximum over 32 trials


def _replay_each_iteration(gpu_ctx, a, b, ref, P=24, T=75, x0=None, flags=0, **kw):
    """The non-chaotic comparison (review r5, weak 1): every Gauss-Newton iteration REPLAYED on the device from the ORACLE's X_{i-1} -- one step from the same
    state, no accumulated divergence.  Returns per iteration (raw-count differences on the active voxels, max |dHTWH| / max |HTWH|)."""
    t = ref["trace"]
    act = (t["has_fit"] == 1) & (t["n1_raw"] > kw.get("n", 25)) & (t["bounds"][:, 5] > 1)
    out = []
    for it in range(t["X"].shape[0]):
        x_in = (np.zeros(6, np.float32) if x0 is None else np.asarray(x0, np.float32)) if it == 0 else t["X"][it - 1]
        rp = gpu_ctx.solve(a, b, 1, x_in, P, T, aux=True, flags=flags, **kw)
        flips = int((rp["aux"]["n2_raw"][0][act] != t["n2_raw"][it][act]).sum())
        out.append((flips, float(np.abs(rp["aux"]["htwh"][0] - t["HTWH"][it]).max() / max(float(np.abs(t["HTWH"][it]).max()), 1e-30))))
    return out


def _check_solution(res, ref, tol_t=TOL_T, tol_r=TOL_R, rtol_std=RTOL_STD, rtol_cov=RTOL_COV):
    import inspect
    _record(inspect.stack()[1].function, res, ref)
    assert np.isfinite(res["X"]).all()
    assert np.abs(res["X"][:3] - ref["X"][:3]).max() <= tol_t, (res["X"], ref["X"])
    assert np.abs(res["X"][3:] - ref["X"][3:]).max() <= tol_r, (res["X"], ref["X"])
    assert np.allclose(res["pred_stds"], ref["pred_stds"], rtol=rtol_std, atol=1e-7)
    d = np.sqrt(np.abs(np.diag(ref["cov"])))
    assert (np.abs(res["cov"] - ref["cov"]) <= rtol_cov * np.outer(d, d) + 1e-12).all()


@pytest.mark.parametrize("name", ["frame_804_805", "sample_pc_1_2"])
def test_golden_pairs_keyframe_table_and_solution(gpu_ctx, name):
    from tests.conftest import load_pair, load_golden
    a, b = load_pair(name); g = load_golden(name)
    r = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
    ax = r["aux"]
    # keyframe build: exact where the arithmetic is integer / comparisons on identical floats
    assert np.array_equal(ax["n1_raw"], g["n1_raw"])
    assert np.array_equal(ax["cluster_bounds"], g["bounds"])
    assert np.array_equal(ax["has_fit"], g["has_fit"])
    f = g["has_fit"] == 1
    assert np.array_equal(ax["l_diag"][f], g["Ldiag"][f])
    # one arithmetic rule on both sides: the Gaussians and their eigenvectors -- signs included -- are the same BITS
    assert np.array_equal(ax["mu1"][f].view(np.uint32), g["mu1"][f].view(np.uint32))
    assert np.array_equal(ax["sigma1"][f].view(np.uint32), g["sigma1"][f].view(np.uint32))
    assert np.array_equal(ax["evecs1"][f].view(np.uint32), g["evecs1"][f].view(np.uint32))
    # Gauss-Newton loop
    act = f & (g["n1_raw"] > 25) & (g["bounds"][:, 5] > 1)
    assert np.array_equal(ax["n2_raw"][0][act], g["n2_raw"][0][act])
    assert (ax["n2_in"][0][act] != np.maximum(g["n2_in"][0][act], 0)).sum() <= 3
    for it in range(7):
        assert (ax["n2_raw"][it][act] != g["n2_raw"][it][act]).sum() <= 8
        assert np.abs(ax["x_hist"][it][:3] - g["X_hist"][it][:3]).max() <= TOL_T
        assert np.abs(ax["x_hist"][it][3:] - g["X_hist"][it][3:]).max() <= TOL_R
        assert np.abs(ax["htwh"][it] - g["HTWH"][it]).max() <= 2e-3 * np.abs(g["HTWH"][it]).max()
    _check_solution(r, g)


def test_first_iteration_is_tight(gpu_ctx, frames, frames_golden):
    """One iteration from X0 = 0: no accumulated boundary flips, so the update itself must agree closely."""
    a, b = frames
    r = gpu_ctx.solve(a, b, 1, np.zeros(6), 24, 75)
    assert np.abs(r["X"] - frames_golden["X_hist"][0]).max() < 5e-6


def test_synthetic_config2_pair_vs_oracle(gpu_ctx):
    from icet_amd import lidar_sim as ls
    from oracle import pyoracle as po
    for order in ("ring", "azimuth"):
        s1, s2, xt = ls.make_pair(order=order)
        a, b = s1.T.numpy(), s2.T.numpy()
        ref = po.solve(a, b, trace=True)
        r = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
        assert np.array_equal(r["aux"]["cluster_bounds"], ref["trace"]["bounds"])
        assert np.array_equal(r["aux"]["has_fit"], ref["trace"]["has_fit"])
        f = ref["trace"]["has_fit"] == 1
        assert np.array_equal(r["aux"]["l_diag"][f], ref["trace"]["Ldiag"][f])
        assert np.array_equal(r["aux"]["sigma1"][f].view(np.uint32), ref["trace"]["sigma1"][f].view(np.uint32))
        _check_solution(r, ref)
        assert np.abs(r["X"][:3] - xt[:3]).max() < 0.03       # and it registers the pair


def test_nonzero_x0_and_longer_run(gpu_ctx, sample_pc):
    from oracle import pyoracle as po
    a, b = sample_pc
    x0 = np.array([0.6, 0, 0, 0, 0, 0], np.float32)      # the reference demo seeds X0 = (1,0,0,0,0,0): icet_cpp_demo.cpp:34-36
    ref = po.solve(a, b, x0=x0, runlen=12)
    r = gpu_ctx.solve(a, b, 12, x0, 24, 75)
    _check_solution(r, ref)
    assert abs(r["X"][0] - 0.645) < 0.01


def test_highres_config5_grid(gpu_ctx, sample_pc):
    """150 x 48 voxels, 10 iterations (BASELINE config 5's grid) on the 131k-point real pair."""
    from oracle import pyoracle as po
    a, b = sample_pc
    ref = po.solve(a, b, runlen=10, bins_phi=48, bins_theta=150, trace=True)
    r = gpu_ctx.solve(a, b, 10, np.zeros(6), 48, 150, aux=True)
    assert np.array_equal(r["aux"]["n1_raw"], ref["trace"]["n1_raw"])
    assert np.array_equal(r["aux"]["cluster_bounds"], ref["trace"]["bounds"])
    assert np.array_equal(r["aux"]["has_fit"], ref["trace"]["has_fit"])
    _check_solution(r, ref)


def test_highres_config5_full_size(gpu_ctx):
    """BASELINE configs[4] at full size: one 128-channel pair (128 rings x 4096 steps, ~500 k points per scan), 150 x 48
    voxels, 10 iterations -- keyframe table bit-exact, X within the tolerance, against the oracle (~0.4 s of CPU)."""
    from icet_amd import lidar_sim as ls
    from oracle import pyoracle as po
    # a 150 x 48 grid has 2.4-degree voxels: from X0 = 0 the loop only converges for a frame-to-frame motion well below a voxel
    # (the bench's 0.5 m default motion leaves it wandering, and a wandering loop amplifies rounding like any chaotic map)
    s1, s2, xt = ls.make_pair(9000, 9001, motion=(0.12, 0.02, 0.005, 0.002, -0.001, 0.004), rings=128, steps=4096)
    a, b = s1.T.numpy(), s2.T.numpy()
    assert a.shape[0] > 450000
    ref = po.solve(a, b, runlen=10, bins_phi=48, bins_theta=150, trace=True)
    r = gpu_ctx.solve(a, b, 10, np.zeros(6), 48, 150, aux=True)
    t, ax = ref["trace"], r["aux"]
    f = t["has_fit"] == 1
    assert f.sum() > 300
    assert np.array_equal(ax["n1_raw"], t["n1_raw"]) and np.array_equal(ax["cluster_bounds"], t["bounds"]) and np.array_equal(ax["has_fit"], t["has_fit"])
    for name_g, name_o in (("mu1", "mu1"), ("sigma1", "sigma1"), ("evecs1", "evecs1"), ("l_diag", "Ldiag")):
        assert np.array_equal(ax[name_g][f].view(np.uint32), t[name_o][f].view(np.uint32)), name_g
    n = a.shape[0]
    assert np.array_equal(gpu_ctx.debug_fetch("src", n), po.scramble(po.c2s(a)[:, 0]))
    _check_solution(r, ref)
    assert np.abs(r["X"][:3] - xt[:3]).max() < 0.05


def test_highres_config5_exactly_as_bench_times_it(gpu_ctx):
    """BASELINE configs[4] EXACTLY as bench.py's `highres` sub-record generates it (review r3, next 1(i)): make_pair(9000, 9001,
    DEFAULT_MOTION, 128, 4096) generated ON THE DEVICE (torch's device generator draws other noise than the CPU's), 150 x 48 voxels,
    10 iterations, X0 = 0.  On this grid (2.4-degree voxels) the loop does not reach the true 0.5 m motion from X0 = 0 -- device and
    oracle both settle at x = 0.133 m -- but they settle at the SAME point: measured 1.1e-5 m apart (scripts/diag_highres.py ->
    profiles/r04_highres_parity.txt; DESIGN.md's 1.6e-3 m of round 3 was a CPU-generated pair whose loop ends in a limit cycle).
    Asserted: the keyframe table and the sort + swap loop bit for bit; every iteration REPLAYED from the oracle's own state (one
    Gauss-Newton step from the same X: no accumulated divergence) within the parity bound with at most a handful of voxels counting
    differently; the free-running result within the parity bound AND within 2 x the oracle's own sensitivity to a 1-ulp perturbation
    of scan 2 computed here (32 trials, x 1.5; floored at 2e-5 m); pred_stds / cov within 2 x the batch p99."""
    from icet_amd import lidar_sim as ls
    from oracle import pyoracle as po
    P, T, RL = 48, 150, 10
    dev = torch.device("cuda", 0)
    s1, s2, xt = ls.make_pair(9000, 9001, ls.DEFAULT_MOTION, 128, 4096, device=dev)
    a, b = s1.T.cpu().numpy(), s2.T.cpu().numpy()
    assert a.shape[0] > 450000
    ref = po.solve(a, b, runlen=RL, bins_phi=P, bins_theta=T, trace=True)
    g = gpu_ctx.solve(a, b, RL, np.zeros(6), P, T, aux=True)
    t, ax = ref["trace"], g["aux"]
    f = t["has_fit"] == 1
    assert f.sum() > 500
    assert np.array_equal(ax["n1_raw"], t["n1_raw"]) and np.array_equal(ax["cluster_bounds"], t["bounds"]) and np.array_equal(ax["has_fit"], t["has_fit"])
    for name_g, name_o in (("mu1", "mu1"), ("sigma1", "sigma1"), ("evecs1", "evecs1"), ("l_diag", "Ldiag")):
        assert np.array_equal(ax[name_g][f].view(np.uint32), t[name_o][f].view(np.uint32)), name_g
    assert np.array_equal(gpu_ctx.debug_fetch("src", a.shape[0]), po.scramble(po.c2s(a)[:, 0]))
    act = f & (t["n1_raw"] > 25) & (t["bounds"][:, 5] > 1)
    first = None
    for it in range(RL):
        x0 = np.zeros(6, np.float32) if it == 0 else t["X"][it - 1]
        rp = gpu_ctx.solve(a, b, 1, x0, P, T, aux=True)
        d = np.abs(rp["X"] - t["X"][it])
        flips = int((rp["aux"]["n2_raw"][0][act] != t["n2_raw"][it][act]).sum())
        assert d[:3].max() <= TOL_T and d[3:].max() <= TOL_R and flips <= 8, (it, d, flips)
        free = np.abs(ax["x_hist"][it] - t["X"][it])
        nd = int((ax["n2_raw"][it][act] != t["n2_raw"][it][act]).sum() + (ax["n2_in"][it][act] != np.maximum(t["n2_in"][it][act], 0)).sum())
        if first is None and nd:
            first = it
        print("iter %d: free-running |dX| %.2e m %.2e rad, %d voxel counts differ; replayed from the oracle's X: %.2e m %.2e rad, %d raw counts differ"
              % (it, free[:3].max(), free[3:].max(), nd, d[:3].max(), d[3:].max(), flips))
    print("first iteration with a differing decision: %s" % first)
    sens = oracle_sensitivity(a, b, runlen=RL, bins_phi=P, bins_theta=T)
    dX = np.abs(g["X"] - ref["X"])
    print("configs[4] as benchmarked: |dX_t| %.3g m, |dX_r| %.3g rad; oracle 1-ulp sensitivity %.3g m / %.3g rad" % (dX[:3].max(), dX[3:].max(), sens[:3].max(), sens[3:].max()))
    _check_solution(g, ref, rtol_std=RTOL_STD_P99, rtol_cov=RTOL_COV_P99)
    assert dX[:3].max() <= max(HATCH * sens[:3].max(), 2e-5) and dX[3:].max() <= max(HATCH * sens[3:].max(), 2e-6), (dX, sens)


def test_million_row_scans(gpu_ctx):
    """Past every size at which a launch shape switches: ~1 M rows per scan (128 rings x 8192 steps) -- the per-bucket sort at its
    largest LDS capacity (one block per CU) with buckets that overflow to the global-scratch radix sort, the swap-loop bit table
    too large for LDS (read from memory), accumulate blocks at 144 KB of LDS.  Same bar: sort + swap loop and the keyframe table
    bit-exact against the unmodified oracle, X within tolerance."""
    from icet_amd import lidar_sim as ls
    from oracle import pyoracle as po
    s1, s2, xt = ls.make_pair(9100, 9101, motion=(0.12, 0.02, 0.005, 0.002, -0.001, 0.004), rings=128, steps=8192)
    a, b = s1.T.numpy(), s2.T.numpy()
    assert a.shape[0] > 900000
    ref = po.solve(a, b, runlen=4, bins_phi=24, bins_theta=75, trace=True)
    r = gpu_ctx.solve(a, b, 4, np.zeros(6), 24, 75, aux=True)
    t, ax = ref["trace"], r["aux"]
    f = t["has_fit"] == 1
    assert f.sum() > 100
    assert np.array_equal(ax["n1_raw"], t["n1_raw"]) and np.array_equal(ax["cluster_bounds"], t["bounds"]) and np.array_equal(ax["has_fit"], t["has_fit"])
    for name_g, name_o in (("mu1", "mu1"), ("sigma1", "sigma1"), ("evecs1", "evecs1"), ("l_diag", "Ldiag")):
        assert np.array_equal(ax[name_g][f].view(np.uint32), t[name_o][f].view(np.uint32)), name_g
    n = a.shape[0]
    assert np.array_equal(gpu_ctx.debug_fetch("src", n), po.scramble(po.c2s(a)[:, 0]))
    assert gpu_ctx.debug_fetch("flags", 1)[0] == 0
    _check_solution(r, ref)


def test_other_parameters(gpu_ctx, frames):
    from oracle import pyoracle as po
    a, b = frames
    for kw in (dict(n=50, thresh=0.3, buff=0.5), dict(n=10, thresh=0.05, buff=0.0)):
        ref = po.solve(a, b, runlen=5, **kw)
        r = gpu_ctx.solve(a, b, 5, np.zeros(6), 24, 75, **kw)
        _check_solution(r, ref)


def test_batch_equals_singles_and_oracle(gpu_ctx, frames, sample_pc):
    """Ragged batch through icet_solve_batch (host pointers): each pair must equal its own single solve."""
    from oracle import pyoracle as po
    a, b = frames; c, d = sample_pc
    s1 = [a, c, a[:30000], c[5000:90000], b]
    s2 = [b, d, b[:31000], d[5000:91000], a]
    x0 = np.zeros((5, 6), np.float32); x0[3, 0] = 0.3
    out = gpu_ctx.solve_batch(s1, s2, 7, x0)
    for k in range(5):
        single = gpu_ctx.solve(s1[k], s2[k], 7, x0[k], 24, 75)
        # integer (fixed-point) atomics: the sums do not depend on arrival order or on how the pair was chunked,
        # so a pair solved inside a batch is BITWISE the pair solved alone
        assert np.array_equal(out["X"][k], single["X"]) and np.array_equal(out["pred_stds"][k], single["pred_stds"])
        ref = po.solve(s1[k], s2[k], x0=x0[k])
        # the truncated scans (k = 2, 3) are partial views with few voxels and a poorly constrained z / roll: held to the larger of the
        # parity bound and 5 x the oracle's own answer-to-answer spread under a 1-ulp perturbation of both scans
        sens = oracle_sensitivity(s1[k], s2[k], scan1_too=True, x0=x0[k]) if k in (2, 3) else np.zeros(6)
        _check_solution(dict(X=out["X"][k], pred_stds=out["pred_stds"][k], cov=out["cov"][k]), ref, max(TOL_T, HATCH * sens[:3].max()), max(TOL_R, HATCH * sens[3:].max()))


def test_identical_scans_within_oracle_sensitivity(gpu_ctx, frames):
    """scan2 == scan1 is the ill-conditioned case: every scan-2 point starts exactly on a scan-1 point, the
    spherical round trip flips some across voxel bounds and the loop wanders to a few-mm fixed point
    (tests/test_oracle.py::test_identical_scans_stay_near_zero_motion).  The first iteration must still agree
    tightly; the final X must agree to within a small multiple of the oracle's own 1-ulp sensitivity."""
    from oracle import pyoracle as po
    a, _ = frames
    # (the device path skips prepScan2's spherical->Cartesian round trip, an identity to 1-2 ulp: with scan2 == scan1
    # its first update is ~1e-7 while the oracle's is ~4e-5 -- both inside the 1-ulp sensitivity measured below)
    r1 = gpu_ctx.solve(a, a, 1, np.zeros(6), 24, 75)
    s1 = oracle_sensitivity(a, a, runlen=1)
    d1 = np.abs(r1["X"] - po.solve(a, a, runlen=1)["X"])
    assert d1[:3].max() <= max(HATCH * s1[:3].max(), 5e-6) and d1[3:].max() <= max(HATCH * s1[3:].max(), 5e-6), (r1["X"], s1)
    r = gpu_ctx.solve(a, a, 7, np.zeros(6), 24, 75)
    ref = po.solve(a, a)
    sens = oracle_sensitivity(a, a)
    d = np.abs(r["X"] - ref["X"])
    assert d[:3].max() <= max(HATCH * sens[:3].max(), TOL_T) and d[3:].max() <= max(HATCH * sens[3:].max(), TOL_R), (r["X"], ref["X"], sens)


def test_edge_cases(gpu_ctx, frames):
    from oracle import pyoracle as po
    a, b = frames
    z = np.zeros((0, 3), np.float32)
    x0 = np.array([0.1, 0, 0, 0, 0, 0.01], np.float32)
    for s1, s2 in ((z, z), (a, z), (z, b)):
        r = gpu_ctx.solve(s1, s2, 7, x0, 24, 75)
        ref = po.solve(s1, s2, x0=x0)
        assert np.allclose(r["X"], ref["X"], atol=1e-7) and np.allclose(r["X"], x0) and (r["pred_stds"] == 0).all()
    rng = np.random.default_rng(0)
    few = (rng.normal(size=(10, 3)) * 5).astype(np.float32)
    r = gpu_ctx.solve(few, few, 7, np.zeros(6), 24, 75)
    assert (r["X"] == 0).all() and (r["pred_stds"] == 0).all()
    # runlen 0: constructor returns X0 and zero pred_stds (src/icet.cpp:36-37,47)
    r = gpu_ctx.solve(a, b, 0, x0, 24, 75)
    assert np.array_equal(r["X"], x0) and (r["pred_stds"] == 0).all()
    # NaN / inf rows are binned at the 1000-sentinel like the reference (src/utils.cpp:116), not propagated
    bad = a.copy(); bad[100] = np.nan; bad[200, 2] = np.inf
    r = gpu_ctx.solve(bad, b, 3, np.zeros(6), 24, 75)
    ref = po.solve(bad, b, runlen=3)
    assert np.isfinite(r["X"]).all()
    _check_solution(r, ref)
    # all-zero scans (every point at the origin): nothing to fit
    zz = np.zeros((5000, 3), np.float32)
    r = gpu_ctx.solve(zz, zz, 3, np.zeros(6), 24, 75)
    assert (r["X"] == 0).all()
    # zero rows with every sign pattern (theta = atan2(+-0, +-0) puts them into four different voxels; their voxel comes from a host table, their scan-2
    # twins are counted per sign pattern, the rank sort's zero bucket is finished by its multi-split), next to tiny vectors whose squares underflow to
    # r = 0 but whose theta is a real angle, with X0 = 0, X0 with a NEGATIVE zero translation and an ordinary X0: per-voxel counts and the keyframe as the oracle's
    rng = np.random.default_rng(7)
    def salted(src):
        o = src.copy()
        idx = rng.choice(o.shape[0], 6000, replace=False)
        signs = rng.choice([0.0, -0.0], size=(6000, 3)).astype(np.float32)
        o[idx] = signs
        tiny = rng.choice(np.setdiff1d(np.arange(o.shape[0]), idx), 200, replace=False)
        o[tiny] = (rng.normal(size=(200, 3)) * 1e-25).astype(np.float32)
        return o
    sa, sb = salted(a), salted(b)
    for x0v in (np.zeros(6, np.float32), np.array([-0.0, 0.0, -0.0, 0, 0, 0], np.float32), np.array([0.05, -0.02, 0.01, 0.001, -0.002, 0.003], np.float32)):
        r = gpu_ctx.solve(sa, sb, 3, x0v, 24, 75, aux=True)
        ref = po.solve(sa, sb, x0=x0v, runlen=3, trace=True)
        t, ax = ref["trace"], r["aux"]
        f = t["has_fit"] == 1
        assert np.array_equal(ax["n1_raw"], t["n1_raw"]) and np.array_equal(ax["cluster_bounds"], t["bounds"]) and np.array_equal(ax["has_fit"], t["has_fit"])
        assert np.array_equal(ax["sigma1"][f].view(np.uint32), t["sigma1"][f].view(np.uint32))
        act = f & (t["n1_raw"] > 25) & (t["bounds"][:, 5] > 1)
        assert np.array_equal(ax["n2_raw"][0][act], t["n2_raw"][0][act]), x0v
        assert (ax["n2_in"][0][act] != np.maximum(t["n2_in"][0][act], 0)).sum() <= 3
        _check_solution(r, ref, tol_t=2 * TOL_T, tol_r=2 * TOL_R, rtol_std=RTOL_STD_LOOSE, rtol_cov=RTOL_COV_LOOSE)


def test_scan2_order_invariance_full_size(gpu_ctx):
    """Size-independent property at the BASELINE config-2 size: the per-voxel sums do not depend on the order
    in which scan-2 points are stored (only scan-1's order feeds findCluster), so shuffling scan 2 must leave
    X unchanged up to float summation order."""
    from icet_amd import lidar_sim as ls
    s1, s2, _ = ls.make_pair()
    a, b = s1.T.numpy(), s2.T.numpy()
    r0 = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
    perm = np.random.default_rng(5).permutation(b.shape[0])
    r1 = gpu_ctx.solve(a, b[perm], 7, np.zeros(6), 24, 75, aux=True)
    assert np.array_equal(r0["aux"]["n2_raw"][0], r1["aux"]["n2_raw"][0])
    assert np.array_equal(r0["aux"]["n2_in"][0], r1["aux"]["n2_in"][0])
    assert np.abs(r0["X"] - r1["X"]).max() < 5e-5
    # idempotence of the keyframe: scan 1 untouched -> identical table bits
    assert np.array_equal(r0["aux"]["cluster_bounds"], r1["aux"]["cluster_bounds"])
    assert np.array_equal(r0["aux"]["mu1"], r1["aux"]["mu1"])


def test_fast_classification_equals_literal_evaluation(gpu_ctx, frames, sample_pc):
    """The accumulate kernel classifies points through LUTs on transcendental-free coordinates and hands only the points within
    a guard band of a voxel edge to the literal formulas (correctly rounded atan2 / acos).  With the option force_exact EVERY
    point takes the literal path.  Every iteration of the fast run is replayed from the same X with force_exact: the integer
    counts of every voxel -- raw and inside the cluster bounds -- must be identical, i.e. the fast path never DECIDES
    differently, on real scans, synthetic scans, a cloud that covers the poles and the 150 x 48 grid.  (The sums agree to float
    rounding only: literal points enter as runs of one, fast points as runs of up to four.)"""
    from icet_amd import lidar_sim as ls
    s1, s2, _ = ls.make_pair()
    # a cloud that covers the WHOLE sphere, poles included: the polar look-up table is sized for the bins near the horizon, so the
    # narrow polar bins near the poles are marked "ask the literal path" cell by cell -- dense shells make every such voxel active
    rng = np.random.default_rng(11)
    dirs = rng.normal(size=(150000, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    shell = (dirs * (8.0 + 0.03 * rng.normal(size=(150000, 1)))).astype(np.float32)
    ang = np.float32(0.01)
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    shell2 = ((dirs * (8.0 + 0.03 * rng.normal(size=(150000, 1)))) @ Rz.T + np.array([0.05, -0.02, 0.03])).astype(np.float32)
    # garbage magnitudes: squares that over- / underflow float32 must take the literal path as well
    odd = frames[1].copy(); odd[::997] *= np.float32(1e25); odd[5::991] *= np.float32(1e-25); odd[7::983, 1] = np.float32(3e38)
    cases = [(frames[0], frames[1], 24, 75, 7), (sample_pc[0], sample_pc[1], 48, 150, 4), (s1.T.numpy(), s2.T.numpy(), 24, 75, 7),
             (shell, shell2, 24, 75, 4), (shell, shell2, 48, 150, 3), (frames[0], odd, 24, 75, 3)]
    for a, b, P, T, rl in cases:
        fast = gpu_ctx.solve(a, b, rl, np.zeros(6), P, T, aux=True)
        again = gpu_ctx.solve(a, b, rl, np.zeros(6), P, T)
        assert np.array_equal(again["X"], fast["X"]) and np.array_equal(again["cov"], fast["cov"])       # run-to-run reproducible
        gpu_ctx.set_option("force_exact", 1)
        try:
            for it in range(rl):
                x0 = np.zeros(6, np.float32) if it == 0 else fast["aux"]["x_hist"][it - 1]
                lit = gpu_ctx.solve(a, b, 1, x0, P, T, aux=True)
                assert np.array_equal(lit["aux"]["n2_raw"][0], fast["aux"]["n2_raw"][it]), (P, T, it)
                assert np.array_equal(lit["aux"]["n2_in"][0], fast["aux"]["n2_in"][it]), (P, T, it)
                # same decisions, other grouping of the float partial sums (runs of one): only rounding apart -- which a line-like voxel
                # can amplify (module docstring), hence the parity bound and not a few ulps
                dX = np.abs(lit["X"] - fast["aux"]["x_hist"][it])
                assert dX[:3].max() <= TOL_T and dX[3:].max() <= TOL_R, (P, T, it, dX)
        finally:
            gpu_ctx.set_option("force_exact", 0)


def test_sync_after_small_device_solves(gpu_ctx):
    """Round 6: icet_sync behind exactly ONE small icet_solve_batch_device call watches a word of pinned memory that the solve's last kernel raises behind its results
    (icet_capi.hip: armed_calls) instead of synchronising the stream; two calls in flight, or any other entry on the context in between, take the stream synchronisation.
    Whatever the path, the results behind an icet_sync must be complete: alternating one / two / three solves per sync on different pairs and X0s, with a keyframe / register
    pair of calls and a large batch in between, every output compared with the pair solved alone."""
    from icet_amd import lidar_sim as ls, api
    import icet_amd
    dev = torch.device("cuda", 0)
    pairs = [ls.make_batch_pair(k, device=dev)[:2] for k in range(3)]
    desc = [([(a.data_ptr(), a.shape[1], a.shape[1])], [(b.data_ptr(), b.shape[1], b.shape[1])]) for a, b in pairs]
    p = api.Params(4, 24, 75, 25, 0.1, 0.1, 0)
    ref = [gpu_ctx.solve(a.T.cpu().numpy(), b.T.cpu().numpy(), 4, np.zeros(6), 24, 75) for a, b in pairs]
    ctx = icet_amd.Context(0)
    outs = [torch.full((1, 48), float("nan"), dtype=torch.float32, device=dev) for _ in range(3)]
    pattern = [1, 1, 2, 1, 3, 1, 1, 2, 1]
    k = 0
    for rep, m in enumerate(pattern):
        used = []
        for _ in range(m):
            j = k % 3; k += 1
            outs[j].fill_(float("nan")); torch.cuda.synchronize()
            ctx.solve_batch_device(desc[j][0], desc[j][1], p, outs[j].data_ptr()); used.append(j)
        ctx.sync()
        for j in used:
            got = outs[j].cpu().numpy()[0]
            assert np.array_equal(got[:6], ref[j]["X"]) and np.array_equal(got[6:12], ref[j]["pred_stds"]), (rep, j)
        if rep == 3:                                                   # the two halves on the same context, then a batch too large for the graph path
            o = torch.zeros((1, 48), dtype=torch.float32, device=dev)
            ctx.keyframe_device(desc[0][0], p); ctx.register_device(desc[0][1], p, o.data_ptr()); ctx.sync()
            assert np.array_equal(o.cpu().numpy()[0, :6], ref[0]["X"])
            big1 = desc[1][0] * 12; big2 = desc[1][1] * 12
            ob = torch.zeros((12, 48), dtype=torch.float32, device=dev)
            ctx.solve_batch_device(big1, big2, p, ob.data_ptr()); ctx.sync()
            assert np.array_equal(ob.cpu().numpy()[7, :6], ref[1]["X"])
    # six hundred solve / sync cycles in a row (the watched word is reset and raised every time), the output cleared in between
    for k in range(600):
        j = k % 3
        outs[j].zero_(); torch.cuda.synchronize()
        ctx.solve_batch_device(desc[j][0], desc[j][1], p, outs[j].data_ptr()); ctx.sync()
        assert np.array_equal(outs[j].cpu().numpy()[0, :6], ref[j]["X"]), k
    ctx.close()


def test_ragged_batch_is_laid_out_xcd_balanced_with_the_callers_order_kept(gpu_ctx):
    """Round 6: a throughput batch (> 64 pairs) whose pairs differ in size by more than a quarter is sorted by size and dealt to the slots of every group of eight
    in snake order (every XCD the same share; icet_capi.hip solve_device_part) -- the caller's order comes back in k_init_state (X0) and k_gn_solve (results).  A pair's
    bits do not depend on its slot: every pair of a 100-pair batch of four sizes (alternating halves and fulls, as the reference's sample scans do, a few quarter-size
    and tiny ones), each with an X0 of its own, must give the bits of that pair solved alone; with batch parts (each part laid out by itself) as well."""
    from icet_amd import lidar_sim as ls, api
    import icet_amd
    dev = torch.device("cuda", 0)
    base = [ls.make_batch_pair(k, device=dev) for k in range(3)]
    n_pairs = 100
    def rows(j, full):
        return full if j % 2 else (full // 2 if j % 10 else (full // 4 if j % 20 else 5000))
    sel = [base[j % 3][:2] for j in range(n_pairs)]
    d1 = [(a.data_ptr(), rows(j, a.shape[1]), a.shape[1]) for j, (a, b) in enumerate(sel)]
    d2 = [(b.data_ptr(), rows(j, b.shape[1]), b.shape[1]) for j, (a, b) in enumerate(sel)]
    x0 = torch.zeros((n_pairs, 6), dtype=torch.float32, device=dev)
    x0[:, 0] = 0.004 * (torch.arange(n_pairs, device=dev) % 11).float(); x0[:, 5] = 0.0007 * (torch.arange(n_pairs, device=dev) % 5).float()
    ctx = icet_amd.Context(0)
    p = api.Params(5, 24, 75, 25, 0.1, 0.1, 0)
    outs = {}
    for parts in (0, 2):
        ctx.set_option("batch_parts", parts)
        o = torch.full((n_pairs, 48), float("nan"), dtype=torch.float32, device=dev)
        ctx.solve_batch_device(d1, d2, p, o.data_ptr(), x0.data_ptr()); ctx.sync()
        outs[parts] = o.cpu().numpy()
    ctx.set_option("batch_parts", 0)
    assert np.array_equal(outs[0], outs[2], equal_nan=True) and np.isfinite(outs[0]).all()
    for j in list(range(0, 24)) + [n_pairs - 1, n_pairs - 2, 40, 60, 61]:
        a, b = sel[j]
        single = gpu_ctx.solve(a[:, :d1[j][1]].T.cpu().numpy(), b[:, :d2[j][1]].T.cpu().numpy(), 5, x0[j].cpu().numpy(), 24, 75)
        assert np.array_equal(outs[0][j, :6], single["X"]) and np.array_equal(outs[0][j, 6:12], single["pred_stds"]), j
    # a uniform batch behind a ragged one on the same context (the layout is decided per call), and a ragged one again
    du1 = [(a.data_ptr(), a.shape[1], a.shape[1]) for a, b in sel]; du2 = [(b.data_ptr(), b.shape[1], b.shape[1]) for a, b in sel]
    o = torch.zeros((n_pairs, 48), dtype=torch.float32, device=dev)
    ctx.solve_batch_device(du1, du2, p, o.data_ptr(), x0.data_ptr()); ctx.sync()
    single = gpu_ctx.solve(sel[7][0].T.cpu().numpy(), sel[7][1].T.cpu().numpy(), 5, x0[7].cpu().numpy(), 24, 75)
    assert np.array_equal(o.cpu().numpy()[7, :6], single["X"])
    o2 = torch.zeros_like(o)
    ctx.solve_batch_device(d1, d2, p, o2.data_ptr(), x0.data_ptr()); ctx.sync()
    assert np.array_equal(o2.cpu().numpy(), outs[0])
    ctx.close()


def test_device_resident_batch_full_size(gpu_ctx):
    """icet_solve_batch_device with inputs resident in HBM (the bench path) at config-3 size with 4 distinct
    pairs cycled 64 x: every replica of a pair must give that pair's single-solve answer."""
    from icet_amd import lidar_sim as ls, api
    import icet_amd
    dev = torch.device("cuda", 0)
    pairs = [ls.make_batch_pair(k, device=dev) for k in range(4)]
    n_rep = 64
    d1 = [(pairs[j % 4][0].data_ptr(), pairs[j % 4][0].shape[1], pairs[j % 4][0].shape[1]) for j in range(4 * n_rep)]
    d2 = [(pairs[j % 4][1].data_ptr(), pairs[j % 4][1].shape[1], pairs[j % 4][1].shape[1]) for j in range(4 * n_rep)]
    out = torch.zeros((4 * n_rep, 48), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx = icet_amd.Context(0)
    p = api.Params(7, 24, 75, 25, 0.1, 0.1, 0)
    ctx.solve_batch_device(d1, d2, p, out.data_ptr())
    ctx.sync()
    res = out.cpu().numpy()
    from oracle import pyoracle as po
    for k in range(4):
        single = gpu_ctx.solve(pairs[k][0].T.cpu().numpy(), pairs[k][1].T.cpu().numpy(), 7, np.zeros(6), 24, 75)
        block = res[k::4]
        assert (block[:, :6] == single["X"]).all() and (block[:, 6:12] == single["pred_stds"]).all()   # bitwise, see above
    # A batch this large is cut into parts that run on helper streams (icet_capi.hip, batch_parts): the cut must not show.
    # Per-pair X0 checks that each part reads its own slice of x0 / writes its own slice of the output.
    x0 = torch.zeros((4 * n_rep, 6), dtype=torch.float32, device=dev)
    x0[:, 0] = 0.01 * (torch.arange(4 * n_rep, device=dev) % 7).float(); x0[:, 5] = 0.001 * (torch.arange(4 * n_rep, device=dev) % 3).float()
    outs = {}
    for parts in ("1", "2", "3"):
        ctx.set_option("batch_parts", int(parts))
        o = torch.zeros_like(out)
        ctx.solve_batch_device(d1, d2, p, o.data_ptr(), x0.data_ptr())
        ctx.sync()
        outs[parts] = o.cpu().numpy()
    ctx.set_option("batch_parts", 0)
    assert np.array_equal(outs["1"], outs["2"]) and np.array_equal(outs["1"], outs["3"])
    j = 4 * n_rep - 3                                      # a pair in the last part, with its own X0
    single = gpu_ctx.solve(pairs[j % 4][0].T.cpu().numpy(), pairs[j % 4][1].T.cpu().numpy(), 7, x0[j].cpu().numpy(), 24, 75)
    assert np.array_equal(outs["3"][j, :6], single["X"])
    ref = po.solve(pairs[0][0].T.cpu().numpy(), pairs[0][1].T.cpu().numpy())
    _check_solution(dict(X=res[0, :6], pred_stds=res[0, 6:12], cov=res[0, 12:].reshape(6, 6)), ref)
    ctx.close()


def _keyframe_vs_oracle(gpu_ctx, a, b):
    from oracle import pyoracle as po
    r = gpu_ctx.solve(a, b, 2, np.zeros(6), 24, 75, aux=True)
    n = a.shape[0]
    sph = po.c2s(a)
    assert np.array_equal(gpu_ctx.debug_fetch("r", n).view(np.uint32), sph[:, 0].view(np.uint32))      # r bit-exact
    word = gpu_ctx.debug_fetch("bin", n)
    assert np.array_equal((word & 0x3FFF).astype(np.int64), po.voxel_of(sph, 24, 75))                  # every row's voxel exact
    assert np.array_equal(gpu_ctx.debug_fetch("src", n), po.scramble(sph[:, 0]))                       # sort + swap loop exact
    ref = po.solve(a, b, runlen=2, trace=True)
    assert np.array_equal(r["aux"]["n1_raw"], ref["trace"]["n1_raw"])
    assert np.array_equal(r["aux"]["cluster_bounds"], ref["trace"]["bounds"])
    return r, ref


def test_rank_sort_and_scramble_corner_cases(gpu_ctx, frames):
    """The hand-written rank sort and the parallel form of the reference's swap loop on inputs built to hit their
    rare paths: (1) thousands of exactly equal keys (zero rows, more than an LDS bucket holds) -- stays in row order;
    (2) a sample that aliases with the data so that ONE bucket receives almost every row with differing keys -- sorted
    by the same code on global scratch; (3) a permutation whose descending chains are longer than the bounded walk --
    the one-lane replay of the literal loop takes over (flag bit 0)."""
    a, b = frames
    rng = np.random.default_rng(42)
    # (1) 9000 zero rows (signed zeros mixed) + real points
    z = a.copy(); z[rng.choice(z.shape[0], 9000, replace=False)] = 0.0; z[::7] *= np.float32(1.0)
    z[rng.choice(z.shape[0], 500, replace=False), 1] = -0.0
    _keyframe_vs_oracle(gpu_ctx, z, b)
    assert gpu_ctx.debug_fetch("flags", 1)[0] == 0
    # (2) every sampled position (stride = ceil(n / 2048)) holds the same radius -> all splitters equal
    n = a.shape[0]; stride = (n + 2047) // 2048
    c = a.copy()
    d = c[::stride]; nrm = np.linalg.norm(d, axis=1, keepdims=True); nrm[nrm == 0] = 1
    c[::stride] = (d / nrm * np.float32(5.0)).astype(np.float32)
    rr = np.linalg.norm(c[::stride].astype(np.float64), axis=1)
    assert np.unique(np.sqrt((c[::stride].astype(np.float32) ** 2).sum(1, dtype=np.float32))).size < 50      # (nearly) one key in the sample
    _keyframe_vs_oracle(gpu_ctx, c, b)
    # (3) radii that make s a long shift: rank(i) = i + 1 (mod n) -> descending chains of length ~n
    m = 20000
    ang = rng.uniform(0, 2 * np.pi, m); el = rng.uniform(-0.3, 0.3, m)
    rad = (5.0 + 1e-3 * ((np.arange(m) + 1) % m)).astype(np.float32)
    p = np.stack([rad * np.cos(el) * np.cos(ang), rad * np.cos(el) * np.sin(ang), rad * np.sin(el)], 1).astype(np.float32)
    _keyframe_vs_oracle(gpu_ctx, p, p)
    assert gpu_ctx.debug_fetch("flags", 1)[0] & 1          # the bounded walk overflowed and the serial replay ran
    # the same three inputs with the executed-step bits from the per-pair recurrence kernel (what a throughput batch runs): equal
    # keys, one giant bucket, and dependency chains as long as a chunk inside every chunk
    gpu_ctx.set_option("exec_pairwise", 1)
    try:
        _keyframe_vs_oracle(gpu_ctx, z, b)
        _keyframe_vs_oracle(gpu_ctx, c, b)
        _keyframe_vs_oracle(gpu_ctx, p, p)
    finally:
        gpu_ctx.set_option("exec_pairwise", -1)


def test_icet_class_mirrors_reference_members(frames, frames_golden):
    import icet_amd
    a, b = frames
    it = icet_amd.ICET(a, b, 7, np.zeros(6, np.float32), 24, 75)          # positional, like the reference call sites
    assert it.X.shape == (6,) and it.pred_stds.shape == (6,)
    assert it.clusterBounds.shape == (24 * 75, 6) and np.array_equal(it.clusterBounds, frames_golden["bounds"])
    assert len(it.ellipsoid1Means) == len(it.ellipsoid1Covariances) == len(it.ellipsoid1Alphas) == 86
    assert it.ellipsoid2Means == [] and it.points2.shape == b.shape and it.HTWH_i.shape == (6, 6) and it.HTWdz_i.shape == (6, 1)
    assert np.abs(it.X[:3] - frames_golden["X"][:3]).max() <= TOL_T


def test_reject_moving_extension_matches_its_oracle_twin(gpu_ctx):
    """ICET_FLAG_REJECT_MOVING is a labelled NON-PARITY extension (the Python variant's moving-object rejection on top of
    the C++ path).  It is held to a CPU twin -- the oracle with ICET_ORACLE_REJECT_MOVING -- on a pair in which half of the boxes
    moved: the same voxels dropped, the same answer within the usual bounds, a different answer than without the flag; off by default."""
    from icet_amd import lidar_sim as ls, api
    from oracle import pyoracle as po
    s1, s2, xt = ls.make_pair_with_moving_objects(shift=(0.6, 0.1), every=2)
    a, b = s1.T.numpy(), s2.T.numpy()
    g = gpu_ctx.solve(a, b, 9, np.zeros(6), 24, 75, aux=True, flags=api.FLAG_REJECT_MOVING)
    o = po.solve(a, b, runlen=9, trace=True, mode=po.REJECT_MOVING)
    _check_solution(g, o)
    for it in range(9):
        assert np.abs(g["aux"]["htwh"][it] - o["trace"]["HTWH"][it]).max() <= 5e-2 * np.abs(o["trace"]["HTWH"][it]).max(), it     # a dropped voxel would show as a missing term
    plain_g = gpu_ctx.solve(a, b, 9, np.zeros(6), 24, 75, aux=True)
    _check_solution(plain_g, po.solve(a, b, runlen=9))
    assert np.array_equal(g["aux"]["x_hist"][:4], plain_g["aux"]["x_hist"][:4])             # nothing happens before the 5th iteration
    assert np.abs(g["X"] - plain_g["X"]).max() > 1e-4                                        # then it is a different answer


def test_eigen_adapter_header_compiles_and_runs(tmp_path, gpu_ctx, frames):
    """include/icet.h -- the adapter with the reference's class name, constructor signature and members -- compiled with g++ against a
    minimal Eigen-API mock (tests/cpp/mock_eigen: this image has no Eigen; the mock pins no numerics) and run through the call
    pattern of src/odometry.cpp:73-82, two frames with X0 seeding: same X as the Python mirror, bit for bit."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a, b = frames
    (tmp_path / "s1.f32").write_bytes(np.ascontiguousarray(a.T).tobytes())
    (tmp_path / "s2.f32").write_bytes(np.ascontiguousarray(b.T).tobytes())
    exe = str(tmp_path / "adapter_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(root, "tests", "cpp", "mock_eigen"), "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "adapter_demo.cpp"), "-L", os.path.join(root, "icet_amd", "lib"), "-licet_hip",
                           "-Wl,-rpath," + os.path.join(root, "icet_amd", "lib"), "-o", exe])
    out = subprocess.run([exe, str(tmp_path / "s1.f32"), str(tmp_path / "s2.f32"), str(a.shape[0]), str(b.shape[0]), "0", "full"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    # level 2: every scan-1 position and every scan-2 row in exactly one voxel list, ascending inside each list, r summed over points1Spherical
    pp = [ln for ln in lines if ln.startswith("perpoint")][0].split()
    assert [int(pp[1]), int(pp[2]), int(pp[3]), int(pp[4]), int(pp[5])] == [a.shape[0], b.shape[0], 0, a.shape[0], b.shape[0]]
    from oracle import pyoracle as po_
    assert abs(float(pp[6]) - float(po_.c2s(a)[:, 0].astype(np.float64).sum())) <= 1e-6 * float(pp[6])
    X1 = np.array(lines[0].split()[1:], np.float32); X2 = np.array(lines[4].split()[1:], np.float32)
    r1 = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75)
    r2 = gpu_ctx.solve(a, b, 7, r1["X"], 24, 75)
    assert np.array_equal(X1, r1["X"]) and np.array_equal(X2, r2["X"])
    assert lines[2].split()[1:] == ["1800", "6", str(b.shape[0]), "10800", "86", "0", "6"]
    # the std::map members mu1 / sigma1 / U / L keyed [theta][phi] (/root/reference/include/icet.h:89-94) against the oracle's table:
    # one entry per fitted voxel, U = eigenvectors^T (orthonormal), L = the 0/1 diagonal, sigma2 / mu2 empty
    from oracle import pyoracle as po
    t = po.solve(a, b, trace=True)["trace"]
    f = t["has_fit"] == 1
    m = lines[3].split()
    thetas = np.nonzero(f)[0] % 75
    assert m[0] == "maps" and int(m[1]) == int(f.sum()) == 86 and int(m[2]) == int(m[3]) == int(m[4]) == len(set(thetas.tolist())) and m[5] == m[6] == "0"
    assert abs(float(m[8]) - float(np.trace(t["sigma1"][f].astype(np.float64), axis1=1, axis2=2).sum())) <= 1e-5 * float(m[8])
    assert float(m[10]) == float(t["Ldiag"][f].sum()) and float(m[12]) < 1e-9
    v0 = min(np.nonzero(f)[0], key=lambda v: (v % 75, v // 75))                  # first key of the nested maps: smallest theta, then phi
    assert np.array_equal(np.array(m[14:17], np.float32), t["mu1"][v0]) and [int(m[18]), int(m[19])] == [v0 % 75, v0 // 75]


def test_cross_lane_primitives_match_the_shuffles(tmp_path):
    """The DPP / permlane cross-lane helpers of icet_device_common.h (wave_xor<D>, wave_incl_sum, wave_reduce_max: what replaced the
    ds_bpermute shuffles in the splitter sort, the scans and the reductions) against __shfl_xor / a serial loop, on the device."""
    import shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "wave_xor")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(root, "icet_amd", "csrc"), "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "hip", "test_wave_xor.hip"), "-o", exe], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_test_points_member(gpu_ctx, frames):
    """`testPoints` (include/icet.h:84, src/icet.cpp:213-231): the sigma points of every pruned axis, bit for bit the oracle's."""
    from oracle import pyoracle as po
    a, b = frames
    g = gpu_ctx.solve(a, b, 1, np.zeros(6), 24, 75, aux=True)["aux"]
    t = po.solve(a, b, runlen=1, trace=True)["trace"]
    f = t["has_fit"] == 1
    pruned = np.repeat(t["Ldiag"] == 0, 2, axis=1) & f[:, None]                  # (V, 6): both sigma points of a pruned axis
    assert pruned.sum() > 20
    assert np.array_equal(g["test_points"][pruned].view(np.uint32), t["sigma_points"][pruned].view(np.uint32))
    assert not g["test_points"][~pruned].any()


def test_constructor_path_from_pageable_host_memory(gpu_ctx, frames):
    """The drop-in path (icet_solve: pageable host scans in, members out) since round 4: scan 2 uploaded on a copy stream beside the
    keyframe build, results + side tables in ONE block and one D2H copy, `points2` from the device, and the call in two halves
    (icet_solve_begin / icet_solve_end).  None of it may show in the bits: a padded leading dimension and the two-halves form give the
    result of the plain call; `points2` is scan 2 under the transform of the last iteration (src/icet.cpp:375-378 with the X BEFORE the
    final update); a second begin and the other host entry points are refused while a begin is pending."""
    import ctypes as C
    import icet_amd
    from icet_amd import api
    a, b = frames
    base = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
    xprev = base["aux"]["x_hist"][5].astype(np.float64)
    exp = (b.astype(np.float64) + xprev[:3]) @ api.euler_R(*xprev[3:]).astype(np.float64)
    assert base["aux"]["points2"].shape == b.shape and np.abs(base["aux"]["points2"] - exp).max() < 3e-5       # float products / FMAs against float64: a few ulp of ~100 m
    r1 = gpu_ctx.solve(a, b, 1, np.zeros(6), 24, 75, aux=True)
    assert np.array_equal(r1["aux"]["points2"], b)                                                                # one iteration: transformed by X0 = 0
    L = api.load_library()
    ctx = icet_amd.Context(0)
    p = api.Params(7, 24, 75, 25, 0.1, 0.1, 0)
    n1, n2 = a.shape[0], b.shape[0]
    ld1, ld2 = n1 + 37, n2 + 5                                             # leading dimensions that are neither n nor multiples of 4
    A = np.full((3, ld1), np.nan, np.float32); A[:, :n1] = a.T
    B = np.full((3, ld2), np.nan, np.float32); B[:, :n2] = b.T
    x0 = np.zeros(6, np.float32)

    def call(fn, aux=None):
        X = np.zeros(6, np.float32); ps = np.zeros(6, np.float32); cov = np.zeros(36, np.float32)
        st = fn(ctx._h, C.byref(p), A.ctypes.data, n1, ld1, B.ctypes.data, n2, ld2, x0.ctypes.data, X.ctypes.data, ps.ctypes.data, cov.ctypes.data, aux)
        return st, X, ps, cov

    st, X, ps, cov = call(L.icet_solve)
    assert st == 0 and np.array_equal(X, base["X"]) and np.array_equal(ps, base["pred_stds"]) and np.array_equal(cov.reshape(6, 6), base["cov"])
    # two halves: outputs arrive at end (the scans stay untouched until then)
    pts2 = np.zeros((3, n2), np.float32); bounds = np.zeros((24 * 75, 6), np.float32)
    ax = api.Aux(); ax.points2 = pts2.ctypes.data_as(api._F); ax.cluster_bounds = bounds.ctypes.data_as(api._F)
    st, X, ps, cov = call(L.icet_solve_begin, C.byref(ax))
    assert st == 0
    assert call(L.icet_solve_begin)[0] == api.ICET_ERR_BAD_ARG             # one begin at a time
    assert call(L.icet_solve)[0] == api.ICET_ERR_BAD_ARG
    with pytest.raises(icet_amd.IcetError):
        ctx.solve_batch([a], [b], 7)
    assert L.icet_solve_end(ctx._h) == 0
    assert np.array_equal(X, base["X"]) and np.array_equal(ps, base["pred_stds"])
    assert np.array_equal(pts2.T, base["aux"]["points2"]) and np.array_equal(bounds, base["aux"]["cluster_bounds"])
    assert L.icet_solve_end(ctx._h) == api.ICET_ERR_BAD_ARG                # nothing pending any more
    A[:] = 0; B[:] = 0                                                      # (the next call must not see the old uploads)
    # the batch entry through the same ring: ragged pairs, bitwise the single solves
    got = ctx.solve_batch([a, b[:40000], a], [b, a[:41000], b], 7)
    assert np.array_equal(got["X"][0], base["X"]) and np.array_equal(got["X"][2], base["X"])
    assert np.array_equal(got["X"][1], gpu_ctx.solve(b[:40000], a[:41000], 7, np.zeros(6), 24, 75)["X"])
    ctx.close()


def test_per_point_members_of_the_reference_object(gpu_ctx, frames):
    """points1Spherical / pointIndices1 / points2Spherical / pointIndices2 (/root/reference/include/icet.h:79,82,95-96): produced on request
    (aux="full").  Scan-1 members bit for bit the oracle's: points1Spherical = cartesianToSpherical of the rows in the order the sort + swap loop
    leaves them (src/icet.cpp:69-83), pointIndices1 = the stable grouping of those positions by voxel (:534-554).  Scan-2 members: the
    spherical coordinates and voxels of `points2` as returned (caller's row order), bit for bit the oracle's c2s / voxel rule on those floats."""
    from oracle import pyoracle as po
    import icet_amd
    a, b = frames
    T, P = 75, 24
    g = gpu_ctx.solve(a, b, 7, np.zeros(6), P, T, aux="full")
    base = gpu_ctx.solve(a, b, 7, np.zeros(6), P, T, aux=True)
    ax = g["aux"]
    assert np.array_equal(g["X"], base["X"]) and np.array_equal(ax["cluster_bounds"], base["aux"]["cluster_bounds"])
    sph = po.c2s(a); src = po.scramble(sph[:, 0])
    assert np.array_equal(ax["points1_spherical"].view(np.uint32), sph[src].view(np.uint32))
    vox = po.voxel_of(sph[src], P, T)                                  # voxel of the row at every position
    assert np.array_equal(ax["point_index1"], np.argsort(vox, kind="stable"))
    assert np.array_equal(np.diff(ax["bin_start1"]), np.bincount(vox, minlength=T * P)) and np.array_equal(np.diff(ax["bin_start1"]), ax["n1_raw"])
    p2 = np.ascontiguousarray(ax["points2"])
    assert np.abs(p2 - base["aux"]["points2"]).max() < 2e-5           # device transform (FMA) against the host pass of the default level
    s2 = po.c2s(p2)
    assert np.array_equal(ax["points2_spherical"].view(np.uint32), s2.view(np.uint32))
    assert np.array_equal(ax["voxel2"], po.voxel_of(s2, P, T))
    it = icet_amd.ICET(a, b, 7, np.zeros(6, np.float32), P, T, side_tables="full")
    v = int(np.argmax(ax["n1_raw"])); th, ph = v % T, v // T
    assert len(it.pointIndices1) == T and len(it.pointIndices1[0]) == P and np.array_equal(it.pointIndices1[th][ph], np.nonzero(vox == v)[0])
    assert sum(len(x) for row in it.pointIndices2 for x in row) == b.shape[0] and np.array_equal(it.pointIndices2[th][ph], np.nonzero(ax["voxel2"] == v)[0])


def test_error_behaviour(gpu_ctx, frames):
    import icet_amd
    a, b = frames
    with pytest.raises(icet_amd.IcetError) as e:
        gpu_ctx.solve(a, b, 7, np.zeros(6), 0, 75)
    assert e.value.status == icet_amd.api.ICET_ERR_BAD_ARG
    with pytest.raises(icet_amd.IcetError):
        gpu_ctx.solve(a, b, -1, np.zeros(6), 24, 75)
    with pytest.raises(icet_amd.IcetError) as e:
        gpu_ctx.solve(a, b, 7, np.zeros(6), 400, 400)
    assert e.value.status == icet_amd.api.ICET_ERR_UNSUPPORTED
    with pytest.raises(icet_amd.IcetError):
        icet_amd.Context(99)
    # the context is still usable after errors
    r = gpu_ctx.solve(a, b, 1, np.zeros(6), 24, 75)
    assert np.isfinite(r["X"]).all()


def test_cpp_host_class_demo(tmp_path, gpu_ctx, frames, frames_golden):
    """The Eigen-free C++ class of include/icet_host.hpp (what include/icet.h adapts to Eigen), compiled with plain g++
    against libicet_hip.so and run like the reference's demo harness: same answer as the Python mirror, bit for bit."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a, b = frames
    (tmp_path / "s1.f32").write_bytes(np.ascontiguousarray(a.T).tobytes())
    (tmp_path / "s2.f32").write_bytes(np.ascontiguousarray(b.T).tobytes())
    exe = str(tmp_path / "host_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "host_demo.cpp"),
                           "-L", os.path.join(root, "icet_amd", "lib"), "-licet_hip", "-Wl,-rpath," + os.path.join(root, "icet_amd", "lib"), "-o", exe])
    out = subprocess.run([exe, str(tmp_path / "s1.f32"), str(tmp_path / "s2.f32"), str(a.shape[0]), str(b.shape[0]), "7", "24", "75"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    X = np.array(lines["X"].split(), np.float32)
    ref = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75)
    assert np.array_equal(X, ref["X"])
    assert np.abs(X[:3] - frames_golden["X"][:3]).max() <= TOL_T
    assert lines["ellipsoids"].split()[0] == "86" and lines["bad_status"].strip() == "1"


# Named exceptions of the 256-pair headline batch (review r3, weak 1c / 1e): measured in profiles/r03_parity_batch.txt, r03_pair232.txt.
#   232: corridor scene, lambda_min(HtWH) 100x below the next eigenvalue; 9.2e-4 m from the oracle = 0.32 sigma of its own predicted std; held to
#        1x the oracle's own sensitivity to a 1-ulp perturbation of scan 2, computed in the test (1.8e-3 m measured).
#   39:  pred_stds 1.2 % / cov 2.5 % off while X agrees (a single-ring voxel's last bit): held to RTOL_STD_LOOSE / RTOL_COV_LOOSE.
BATCH_EXCEPTIONS_X = (232,)
BATCH_EXCEPTIONS_STD = (39, 232)


def test_many_pairs_parity_natural_signs(gpu_ctx):
    """ALL 256 DISTINCT pairs of the headline batch (BASELINE configs[2]) pushed through ONE icet_solve_batch_device call -- the call
    bench.py times -- and THAT output compared with the UNMODIFIED oracle (review r3, next 1(ii)); every pair is also solved alone
    through icet_solve for the keyframe tables, and must give the batch's bits.  The reference's result depends on the signs of the
    scan-1 eigenvectors (rows-of-V sigma points; `L*U^T` with U = V^T applies V, Q8), and on near-degenerate covariances
    the QR iteration's signs turn on the last bits of sigma1; with one arithmetic rule on both sides those bits are the same, so nothing
    is borrowed from the device.  Asserted: keyframe table bit-exact on EVERY pair; X within 2e-4 m / 2e-5 rad, pred_stds / cov within
    2 x their measured p99 (0.15 % / 0.3 %) on every pair but the NAMED exceptions above -- which are held to their own stated bounds
    (232: 1x the oracle's 1-ulp sensitivity computed here; 39: 2.5 % / 5 %) -- and the distribution: median below 5e-6 m,
    at least 97 % of the pairs within SURVEY 8(c)'s starting values of 1e-4 m / 1e-5 rad."""
    from concurrent.futures import ThreadPoolExecutor
    import icet_amd
    from icet_amd import lidar_sim as ls, api
    from oracle import pyoracle as po
    dev = torch.device("cuda", 0)
    N = 256
    pairs = [ls.make_batch_pair(k, device=dev) for k in range(N)]
    d1 = [(p[0].data_ptr(), p[0].shape[1], p[0].shape[1]) for p in pairs]; d2 = [(p[1].data_ptr(), p[1].shape[1], p[1].shape[1]) for p in pairs]
    out = torch.zeros((N, 48), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    bctx = icet_amd.Context(0)
    bctx.solve_batch_device(d1, d2, api.Params(7, 24, 75, 25, 0.1, 0.1, 0), out.data_ptr()); bctx.sync()
    batch = out.cpu().numpy()
    bctx.close()
    dts, drs, worst, outside = np.zeros(N), np.zeros(N), np.zeros(4), []
    d_lit, d_modes = np.zeros((N, 6)), np.zeros((N, 6))
    with ThreadPoolExecutor(min(os.cpu_count() or 1, 16)) as ex:
        for k0 in range(0, N, 32):                                  # 32 pairs at a time: the oracle runs on the host cores while the device solves
            ks = list(range(k0, min(k0 + 32, N)))
            host = [(pairs[k][0].T.cpu().numpy(), pairs[k][1].T.cpu().numpy()) for k in ks]
            futs = [ex.submit(po.solve, a, b, trace=True) for a, b in host]
            futs_lit = [ex.submit(po.solve, a, b, mode=po.LIBMF) for a, b in host]      # the oracle with the reference's literal expression types (reported, not bounded: below)
            gpus = [gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True) for a, b in host]
            for k, (a, b), g, fu, ful in zip(ks, host, gpus, futs, futs_lit):
                ref = fu.result()
                lit = ful.result()
                d_lit[k] = np.abs(batch[k, :6] - lit["X"]); d_modes[k] = np.abs(ref["X"] - lit["X"])
                t, ax = ref["trace"], g["aux"]
                f = t["has_fit"] == 1
                assert np.array_equal(ax["n1_raw"], t["n1_raw"]) and np.array_equal(ax["cluster_bounds"], t["bounds"]) and np.array_equal(ax["has_fit"], t["has_fit"]), k
                for name_g, name_o in (("mu1", "mu1"), ("sigma1", "sigma1"), ("evecs1", "evecs1"), ("l_diag", "Ldiag")):
                    assert np.array_equal(ax[name_g][f].view(np.uint32), t[name_o][f].view(np.uint32)), (k, name_g)
                # the pair inside the 256-pair call is bit for bit the pair solved alone (integer accumulation)
                assert np.array_equal(batch[k, :6], g["X"]) and np.array_equal(batch[k, 6:12], g["pred_stds"]) and np.array_equal(batch[k, 12:], g["cov"].reshape(36)), k
                gb = dict(X=batch[k, :6], pred_stds=batch[k, 6:12], cov=batch[k, 12:].reshape(6, 6))       # what the batch entry returned
                dts[k] = np.abs(gb["X"][:3] - ref["X"][:3]).max(); drs[k] = np.abs(gb["X"][3:] - ref["X"][3:]).max()
                if k in BATCH_EXCEPTIONS_X:
                    sens = oracle_sensitivity(a, b)
                    print("pair %d (named exception): |dX_t| %.3g m, |dX_r| %.3g rad; oracle 1-ulp sensitivity %.3g m / %.3g rad" % (k, dts[k], drs[k], sens[:3].max(), sens[3:].max()))
                    if dts[k] > TOL_T or drs[k] > TOL_R:
                        outside.append(k)
                    assert dts[k] <= max(sens[:3].max(), TOL_T) and drs[k] <= max(sens[3:].max(), TOL_R), (k, dts[k], drs[k], sens)
                    continue
                loose = k in BATCH_EXCEPTIONS_STD
                _check_solution(gb, ref, rtol_std=RTOL_STD_LOOSE if loose else RTOL_STD_P99, rtol_cov=RTOL_COV_LOOSE if loose else RTOL_COV_P99)
                if not loose:
                    dd = np.sqrt(np.abs(np.diag(ref["cov"])))
                    worst = np.maximum(worst, [dts[k], drs[k], np.abs(gb["pred_stds"] / ref["pred_stds"] - 1).max(), (np.abs(gb["cov"] - ref["cov"]) / np.outer(dd, dd)).max()])
    within = ((dts <= 1e-4) & (drs <= 1e-5)).mean()
    print("%d pairs through one icet_solve_batch_device call, natural signs: outside 2e-4 m / 2e-5 rad: %s; among the unexceptional pairs max |dX_t| %.3g m, |dX_r| %.3g rad, rel pred_stds %.3g, rel cov %.3g; median |dX_t| %.3g; within 1e-4 m / 1e-5 rad: %.1f %%"
          % (N, outside, *worst, np.median(dts), 100 * within))
    assert np.median(dts) < 5e-6 and np.median(drs) < 5e-7 and within >= 0.97, (np.median(dts), np.median(drs), within)
    # GPU vs the oracle in its LITERAL mode (glibc float atan2 / acos / sin / cos, sequential float sums, std::hypot: the expression types of
    # src/utils.cpp:103-108,134-136 and src/icet.cpp:160-162 -- the closest thing to the reference binary that exists here).  No bound is asserted:
    # the two modes of the SAME oracle differ by as much from each other (eigenvector signs of thin voxels turning on last bits); the figures an
    # integrator needs are printed, written to gpurun_out/parity_literal.txt and quoted in INTEGRATION.md.
    def stats(d):
        dt_, dr_ = d[:, :3].max(1), d[:, 3:].max(1)
        big = np.nonzero((dt_ > 3e-4) | (dr_ > 1e-4))[0]
        return "median %.3g m / %.3g rad, p99 %.3g m / %.3g rad, max %.3g m / %.3g rad, over 3e-4 m or 1e-4 rad: %d pairs %s" % (
            np.median(dt_), np.median(dr_), np.percentile(dt_, 99), np.percentile(dr_, 99), dt_.max(), dr_.max(), big.size, big.tolist())
    lines = ["|dX| over the %d bench pairs (one icet_solve_batch_device call):" % N, "GPU vs literal-mode oracle:          " + stats(d_lit),
             "shared-rule oracle vs literal oracle: " + stats(d_modes), "GPU vs shared-rule oracle:           " + stats(np.abs(np.concatenate([dts[:, None]] * 3 + [drs[:, None]] * 3, 1)))]
    print("\n".join(lines))
    outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(outdir):
        with open(os.path.join(outdir, "parity_literal.txt"), "w") as fh:
            fh.write("\n".join(lines) + "\n")
    assert np.isfinite(d_lit).all()


def test_two_stage_solve_gives_the_bits_of_the_one_block_solve(gpu_ctx, sample_pc):
    """Fine grids (V > 4096) in small batches (<= 4 pairs) reduce H^T W H in two stages -- three blocks per pair reduce their share of the
    active voxels to 27 partial sums, one block adds them and runs the 6 x 6 part (BASELINE north_star's "two-stage reduction") -- larger
    batches in one block per pair.  Every form adds in ONE canonical tree (virtual blocks of 512 slots), so a pair gives the same bits
    alone (two stages), inside a batch of 6 (one block), and when undecided points wait in the overflow list (force_exact: stage 1
    declines, stage 2 drains and solves alone)."""
    import icet_amd
    from icet_amd import api
    dev = torch.device("cuda", 0)
    a, b = sample_pc
    ta = torch.from_numpy(np.ascontiguousarray(a.T)).to(dev); tb = torch.from_numpy(np.ascontiguousarray(b.T)).to(dev)
    d1 = [(ta.data_ptr(), a.shape[0], a.shape[0])]; d2 = [(tb.data_ptr(), b.shape[0], b.shape[0])]
    prm = api.Params(6, 48, 150, 25, 0.1, 0.1, 0)
    for exact in (0, 1):
        ctx = icet_amd.Context(0); ctx.set_option("force_exact", exact); ctx.set_option("graph", 0)
        one = torch.zeros((1, 48), dtype=torch.float32, device=dev); six = torch.zeros((6, 48), dtype=torch.float32, device=dev)
        ctx.solve_batch_device(d1, d2, prm, one.data_ptr()); ctx.sync()
        ctx.solve_batch_device(d1 * 6, d2 * 6, prm, six.data_ptr()); ctx.sync()
        assert bool(torch.isfinite(one).all()) and all(torch.equal(six[k], one[0]) for k in range(6)), exact
        if exact == 0:
            host = gpu_ctx.solve(a, b, 6, np.zeros(6), 48, 150)             # icet_solve: one pair, two stages
            assert np.array_equal(host["X"], one[0, :6].cpu().numpy())
        ctx.close()


def test_graph_replay_of_small_device_batches(gpu_ctx):
    """A device-resident batch of <= 8 pairs whose launch key (geometry, sizes, workspace and output pointers) repeats is captured into a
    hipGraph at its second occurrence and replayed afterwards (option "graph", on by default).  The replay must be indistinguishable:
    same bits as the eager launches; the descriptor table and X0 are re-read on every replay, so OTHER scans of the same size at other
    addresses and another X0 give their own answers; a call with another key falls back to eager launches and back again."""
    import icet_amd
    from icet_amd import lidar_sim as ls, api
    dev = torch.device("cuda", 0)
    s1, s2, _ = ls.make_batch_pair(3, device=dev)
    t2 = s2.clone(); t2[0] += 0.01                                   # another scan 2 of the same size at another address
    d1 = [(s1.data_ptr(), s1.shape[1], s1.shape[1])]; d2 = [(s2.data_ptr(), s2.shape[1], s2.shape[1])]; e2 = [(t2.data_ptr(), t2.shape[1], t2.shape[1])]
    prm = api.Params(7, 24, 75, 25, 0.1, 0.1, 0)
    x0 = torch.zeros((1, 6), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def run(ctx, scans2, out):
        ctx.solve_batch_device(d1, scans2, prm, out.data_ptr(), x0.data_ptr()); ctx.sync()
        return out.cpu().numpy().copy()

    eager = icet_amd.Context(0); eager.set_option("graph", 0)
    o = torch.zeros((1, 48), dtype=torch.float32, device=dev)
    ref_a = run(eager, d2, o); ref_b = run(eager, e2, o)
    x0[0, 0] = 0.05; torch.cuda.synchronize()
    ref_c = run(eager, d2, o)
    x0[0, 0] = 0.0; torch.cuda.synchronize()
    assert not np.array_equal(ref_a, ref_b) and not np.array_equal(ref_a, ref_c)
    g = icet_amd.Context(0)                                          # default: graph replay on
    outs = [run(g, d2, o) for _ in range(5)]                         # eager, capture + replay, replay ...
    assert all(np.array_equal(v, ref_a) for v in outs)
    assert np.array_equal(run(g, e2, o), ref_b)                      # same key, other scan: the replay reads the new descriptor
    x0[0, 0] = 0.05; torch.cuda.synchronize()
    assert np.array_equal(run(g, d2, o), ref_c)                      # same key, other X0 contents
    x0[0, 0] = 0.0; torch.cuda.synchronize()
    o2 = torch.zeros((1, 48), dtype=torch.float32, device=dev)
    assert np.array_equal(run(g, d2, o2), ref_a)                     # another output pointer = another key: eager again
    assert np.array_equal(run(g, d2, o), ref_a) and np.array_equal(run(g, d2, o), ref_a) and np.array_equal(run(g, d2, o), ref_a)
    two = torch.zeros((2, 48), dtype=torch.float32, device=dev)      # and a different batch size in between
    g.solve_batch_device(d1 + d1, d2 + e2, prm, two.data_ptr()); g.sync()
    assert np.array_equal(two[0].cpu().numpy(), ref_a[0]) and np.array_equal(two[1].cpu().numpy(), ref_b[0])
    assert np.array_equal(run(g, d2, o), ref_a)
    eager.close(); g.close()


def test_keyframe_and_register_halves_equal_the_whole_solve(gpu_ctx):
    """icet_keyframe_device + icet_register_device (the solve in two halves, for the sequential callers) give the bits of
    icet_solve_batch_device; a parked keyframe can be registered against more than once; a mismatched call is refused."""
    import icet_amd
    from icet_amd import lidar_sim as ls, api
    dev = torch.device("cuda", 0)
    pairs = [ls.make_batch_pair(k, device=dev) for k in range(3)]
    d1 = [(p[0].data_ptr(), p[0].shape[1], p[0].shape[1]) for p in pairs]; d2 = [(p[1].data_ptr(), p[1].shape[1], p[1].shape[1]) for p in pairs]
    xd = torch.zeros((3, 6), dtype=torch.float32, device=dev); xd[:, 0] = torch.tensor([0.0, 0.02, -0.01], device=dev)
    whole = torch.zeros((3, 48), dtype=torch.float32, device=dev); halves = torch.zeros_like(whole); again = torch.zeros_like(whole)
    torch.cuda.synchronize()
    prm = api.Params(7, 24, 75, 25, 0.1, 0.1, 0)
    ctx = icet_amd.Context(0)
    ctx.solve_batch_device(d1, d2, prm, whole.data_ptr(), xd.data_ptr()); ctx.sync()
    ctx.keyframe_device(d1, prm)
    ctx.register_device(d2, prm, halves.data_ptr(), xd.data_ptr())
    ctx.register_device(d2, prm, again.data_ptr(), xd.data_ptr()); ctx.sync()
    assert torch.equal(whole, halves) and torch.equal(whole, again) and bool(torch.isfinite(whole).all())
    with pytest.raises(icet_amd.IcetError):                      # other grid than the parked keyframe's
        ctx.register_device(d2, api.Params(7, 48, 150, 25, 0.1, 0.1, 0), again.data_ptr())
    with pytest.raises(icet_amd.IcetError):                      # other number of pairs
        ctx.register_device(d2[:2], prm, again.data_ptr())
    ctx.solve_batch_device(d1[:1], d2[:1], prm, again.data_ptr()); ctx.sync()
    with pytest.raises(icet_amd.IcetError):                      # a whole solve un-parks the keyframe
        ctx.register_device(d2, prm, again.data_ptr())
    ctx.close()


def test_rccl_gather_on_one_rank_is_the_identity():
    """icet_amd.dist.gather_results through RCCL (torch.distributed backend "nccl") with a process group of ONE rank on cuda:0: the
    all-gather + de-interleave must return the rank's own rows.  This is as much of the N > 1 path as a 1-GPU box can run; the
    world-size-2 logic is covered by the gloo test on CPU (tests/test_abi_and_host.py)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from icet_amd.dist import gather_results, solve_sharded
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(7 * 48, dtype=torch.float32, device=dev).reshape(7, 48) * 0.5
y = gather_results(x, 7)
assert y.shape == x.shape and torch.equal(x, y) and y.data_ptr() != x.data_ptr()
z = solve_sharded(5, lambda ids: torch.full((len(ids), 48), 3.0, device=dev) * torch.tensor(ids, device=dev, dtype=torch.float32)[:, None], dev)
assert torch.equal(z[:, 0].cpu(), torch.tensor([0., 3., 6., 9., 12.]))
dist.barrier(); dist.destroy_process_group()
print("rccl one-rank gather ok")
"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "rccl one-rank gather ok" in out.stdout, out.stderr[-2000:]


def test_multi_device_entry_with_one_device_equals_the_single_context(gpu_ctx, frames, sample_pc):
    """icet_multi_* with n_devices = 1 (all this box has): the host-pointer entry must reproduce icet_solve_batch bit for bit, the
    device-resident entry icet_solve_batch_device -- X0 scatter and result gather included -- and unknown device ids are refused.
    (No run on N > 1 DEVICES has happened anywhere yet: DESIGN.md section 9; the N > 1 sharding itself runs in the next test.)"""
    import icet_amd
    from icet_amd import lidar_sim as ls, api
    a, b = frames; c, d = sample_pc
    s1 = [a, c, a[:30000], b]; s2 = [b, d, b[:31000], a]
    x0 = np.zeros((4, 6), np.float32); x0[1, 0] = 0.3; x0[3, 5] = 0.01
    ref = gpu_ctx.solve_batch(s1, s2, 7, x0)
    m = icet_amd.MultiContext([0])
    got = m.solve_batch(s1, s2, 7, x0)
    for key in ("X", "pred_stds", "cov"):
        assert np.array_equal(got[key], ref[key]), key
    dev = torch.device("cuda", 0)
    pairs = [ls.make_batch_pair(k, device=dev) for k in range(3)]
    d1 = [(p[0].data_ptr(), p[0].shape[1], p[0].shape[1]) for p in pairs]; d2 = [(p[1].data_ptr(), p[1].shape[1], p[1].shape[1]) for p in pairs]
    xd = torch.zeros((3, 6), dtype=torch.float32, device=dev); xd[:, 0] = torch.tensor([0.0, 0.02, -0.01], device=dev)
    o_multi = torch.zeros((3, 48), dtype=torch.float32, device=dev); o_single = torch.zeros_like(o_multi)
    torch.cuda.synchronize()
    prm = api.Params(7, 24, 75, 25, 0.1, 0.1, 0)
    m.solve_batch_device(d1, d2, prm, o_multi.data_ptr(), xd.data_ptr())
    ctx = icet_amd.Context(0); ctx.solve_batch_device(d1, d2, prm, o_single.data_ptr(), xd.data_ptr()); ctx.sync(); ctx.close()
    torch.cuda.synchronize()
    assert torch.equal(o_multi, o_single) and bool(torch.isfinite(o_multi).all())
    # X0 produced on a side stream and handed over as `producer_stream`: the library's streams must wait for it (no host sync here)
    side = torch.cuda.Stream(device=dev)
    o_after = torch.zeros_like(o_multi); xs = torch.empty_like(xd)
    with torch.cuda.stream(side):
        big = torch.randn(4096, 4096, device=dev); big = big @ big                 # keeps the stream busy for a while
        xs.copy_(xd + big[0, 0] * 0.0)
    m.solve_batch_device(d1, d2, prm, o_after.data_ptr(), xs.data_ptr(), producer_stream=side.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(o_after, o_single)
    # the library's own RCCL gather (ncclCommInitAll + ncclAllGather behind icet_multi, north_star's wording) with the one rank this box has
    m.set_option("gather", 1)
    o_rccl = torch.full_like(o_multi, float("nan"))
    m.solve_batch_device(d1, d2, prm, o_rccl.data_ptr(), xd.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(o_rccl, o_single)
    m.set_option("gather", 0)
    m.close()
    m2 = icet_amd.MultiContext([0, 0])
    with pytest.raises(icet_amd.IcetError) as e:                   # one rank per GPU: a repeated device cannot form a communicator
        m2.set_option("gather", 1)
    assert e.value.status == api.ICET_ERR_UNSUPPORTED
    m2.close()
    with pytest.raises(icet_amd.IcetError) as e:
        icet_amd.MultiContext([0, 63])
    assert e.value.status == api.ICET_ERR_NO_DEVICE


@pytest.mark.parametrize("shards", [2, 3])
def test_multi_device_sharding_with_shards_sharing_one_gpu(gpu_ctx, frames, sample_pc, shards):
    """The N > 1 logic of icet_multi_* on the one GPU of this box: device id 0 listed `shards` times gives that many contexts, host
    threads and result buffers; pair k goes to shard k mod N, X0 rows are scattered to the shards and the 48 result floats per pair
    gathered back (strided copies).  Ragged pairs, more pairs than shards and fewer pairs than shards; results must be the single
    context's, bit for bit, through the host-pointer entry and through the device-resident one."""
    import icet_amd
    from icet_amd import lidar_sim as ls, api
    a, b = frames; c, d = sample_pc
    s1 = [a, c, a[:30000], b, c[:50000]]; s2 = [b, d, b[:31000], a, d[:47000]]
    x0 = np.zeros((5, 6), np.float32); x0[1, 0] = 0.3; x0[3, 5] = 0.01; x0[4, 1] = -0.05
    m = icet_amd.MultiContext([0] * shards)
    for npairs in (5, 1):                                            # 1 pair: shards 1.. have nothing to do
        ref = gpu_ctx.solve_batch(s1[:npairs], s2[:npairs], 7, x0[:npairs])
        got = m.solve_batch(s1[:npairs], s2[:npairs], 7, x0[:npairs])
        for key in ("X", "pred_stds", "cov"):
            assert np.array_equal(got[key], ref[key]), (key, npairs)
    dev = torch.device("cuda", 0)
    n = 7
    pairs = [ls.make_batch_pair(k, device=dev) for k in range(n)]
    d1 = [(p[0].data_ptr(), p[0].shape[1], p[0].shape[1]) for p in pairs]; d2 = [(p[1].data_ptr(), p[1].shape[1], p[1].shape[1]) for p in pairs]
    xd = torch.zeros((n, 6), dtype=torch.float32, device=dev); xd[:, 0] = torch.linspace(-0.02, 0.03, n, device=dev)
    o_multi = torch.full((n, 48), float("nan"), dtype=torch.float32, device=dev); o_single = torch.zeros((n, 48), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    prm = api.Params(7, 24, 75, 25, 0.1, 0.1, 0)
    m.solve_batch_device(d1, d2, prm, o_multi.data_ptr(), xd.data_ptr())
    ctx = icet_amd.Context(0); ctx.solve_batch_device(d1, d2, prm, o_single.data_ptr(), xd.data_ptr()); ctx.sync(); ctx.close()
    torch.cuda.synchronize()
    assert torch.equal(o_multi, o_single) and bool(torch.isfinite(o_multi).all())
    # the asynchronous form: three calls queued back to back (other output buffers, another X0, fewer pairs), ONE sync at the end
    outs = [torch.full((n, 48), float("nan"), dtype=torch.float32, device=dev) for _ in range(3)]
    xd2 = xd + 0.01
    torch.cuda.synchronize()
    m.solve_batch_device(d1, d2, prm, outs[0].data_ptr(), xd.data_ptr(), asynchronous=True)
    m.solve_batch_device(d1, d2, prm, outs[1].data_ptr(), xd2.data_ptr(), asynchronous=True)
    m.solve_batch_device(d1[:3], d2[:3], prm, outs[2].data_ptr(), xd.data_ptr(), asynchronous=True)
    m.sync()
    ctx = icet_amd.Context(0); o2 = torch.zeros_like(o_single)
    ctx.solve_batch_device(d1, d2, prm, o2.data_ptr(), xd2.data_ptr()); ctx.sync(); ctx.close()
    assert torch.equal(outs[0], o_single) and torch.equal(outs[1], o2) and torch.equal(outs[2][:3], o_single[:3]) and bool(torch.isnan(outs[2][3:]).all())
    with pytest.raises(icet_amd.IcetError):                          # a failing call (no such grid) is reported by the sync, and the handle stays usable
        m.solve_batch_device(d1, d2, api.Params(7, 400, 400, 25, 0.1, 0.1, 0), outs[0].data_ptr(), asynchronous=True)
        m.sync()
    m.solve_batch_device(d1, d2, prm, outs[0].data_ptr(), xd.data_ptr(), asynchronous=True); m.sync()
    assert torch.equal(outs[0], o_single)
    m.close()


def test_config3_2048_pairs_over_8_shards_on_one_card(gpu_ctx):
    """BASELINE configs[3] at FULL size on the one card of this box: 2048 pairs (the 256 distinct bench pairs aliased x 8, every alias with its own
    X0) round-robin over EIGHT shards -- device 0 listed eight times: eight contexts, host threads, workspaces and result buffers, i.e. everything of
    `icet_multi_*` but the peer link -- result rows gathered by strided copies into one buffer, bit-equal to ONE context solving the 2048 pairs; the
    RCCL gather refuses repeated device ids instead of hanging; the eight workspaces' high-water mark is asserted (what `--gpus 8` needs per card is
    an eighth of it).  No run on more than one device has happened anywhere: this keeps the day hardware appears a formality."""
    import icet_amd
    from icet_amd import lidar_sim as ls, api
    dev = torch.device("cuda", 0)
    N, D = 2048, 256
    pairs = [ls.make_batch_pair(k, device=dev) for k in range(D)]
    d1 = [(pairs[k % D][0].data_ptr(), pairs[k % D][0].shape[1], pairs[k % D][0].shape[1]) for k in range(N)]
    d2 = [(pairs[k % D][1].data_ptr(), pairs[k % D][1].shape[1], pairs[k % D][1].shape[1]) for k in range(N)]
    x0 = torch.zeros((N, 6), dtype=torch.float32, device=dev)
    x0[:, 0] = 0.002 * (torch.arange(N, device=dev) // D).float(); x0[:, 5] = 0.0005 * (torch.arange(N, device=dev) % 5).float()
    prm = api.Params(7, 24, 75, 25, 0.1, 0.1, 0)
    o_single = torch.zeros((N, 48), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx = icet_amd.Context(0)
    ctx.solve_batch_device(d1, d2, prm, o_single.data_ptr(), x0.data_ptr()); ctx.sync(); ctx.close()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    m = icet_amd.MultiContext([0] * 8)
    o_multi = torch.full((N, 48), float("nan"), dtype=torch.float32, device=dev)
    m.solve_batch_device(d1, d2, prm, o_multi.data_ptr(), x0.data_ptr())
    torch.cuda.synchronize()
    used_gb = (free0 - torch.cuda.mem_get_info(0)[0]) / 2 ** 30
    assert torch.equal(o_multi, o_single) and bool(torch.isfinite(o_multi).all())
    # aliases of one pair differ through their X0 (every slot really got ITS row of x0): most land on the same registration, none is a copy of another's slot
    r = o_multi.cpu().numpy().reshape(8, D, 48)
    assert not np.array_equal(r[0], r[1]) and np.median(np.abs(r[1:, :, :3] - r[0:1, :, :3]).max(2)) < 1e-3
    # asynchronous form, twice back to back into two buffers, one sync
    o_a = torch.full((N, 48), float("nan"), dtype=torch.float32, device=dev); o_b = torch.full((N, 48), float("nan"), dtype=torch.float32, device=dev)
    m.solve_batch_device(d1, d2, prm, o_a.data_ptr(), x0.data_ptr(), asynchronous=True)
    m.solve_batch_device(d1, d2, prm, o_b.data_ptr(), x0.data_ptr(), asynchronous=True)
    m.sync()
    assert torch.equal(o_a, o_single) and torch.equal(o_b, o_single)
    print("configs[3] on one card: 2048 pairs over 8 shards bit-equal to one context; the eight workspaces hold %.1f GB of HBM (%.2f GB per shard of 256 pairs)" % (used_gb, used_gb / 8))
    assert used_gb < 48.0, used_gb                                    # 8 x (256 pairs x ~12 MB of tables + scratch): measured ~26 GB; a leak or a per-call reallocation would show here
    with pytest.raises(icet_amd.IcetError):                          # one ncclAllGather needs distinct devices: refused (at the option or at the call), not hung
        m.set_option("gather", 1)
        m.solve_batch_device(d1, d2, prm, o_a.data_ptr(), x0.data_ptr())
    m.close()


def test_real_scan_batch_keyframe_bits_and_solution(gpu_ctx, frames, sample_pc):
    """The REAL-data throughput batch of bench.py (`sample_batch`; review r4, item 4): 64 pairs built from the reference's two sample pairs, pair k turned by
    its own small rotation, thousands of exact-zero rows per scan kept, through ONE icet_solve_batch_device call.  Every pair must give its single-solve bits;
    on a sample of them the keyframe table is the oracle's bit for bit (zero rows: the voxel of r = 0 from the two sign bits, the rank sort's bucket of equal
    keys finished by its multi-split, the zero-row voxel's cluster) and X / pred_stds / cov within the usual bounds."""
    import icet_amd
    from icet_amd import lidar_sim as ls, api
    from oracle import pyoracle as po
    dev = torch.device("cuda", 0)
    base = [tuple(torch.from_numpy(np.ascontiguousarray(x.T)).to(dev) for x in frames), tuple(torch.from_numpy(np.ascontiguousarray(x.T)).to(dev) for x in sample_pc)]
    N = 64
    pairs = []
    for k in range(N):
        R = torch.as_tensor(ls.real_batch_rotation(k), device=dev)
        pairs.append(((R @ base[k % 2][0]).contiguous(), (R @ base[k % 2][1]).contiguous()))
    d1 = [(p[0].data_ptr(), p[0].shape[1], p[0].shape[1]) for p in pairs]; d2 = [(p[1].data_ptr(), p[1].shape[1], p[1].shape[1]) for p in pairs]
    out = torch.zeros((N, 48), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx = icet_amd.Context(0)
    ctx.solve_batch_device(d1, d2, api.Params(7, 24, 75, 25, 0.1, 0.1, 0), out.data_ptr()); ctx.sync()
    batch = out.cpu().numpy()
    ctx.close()
    assert np.isfinite(batch).all()
    for k in (0, 1, 2, 3, 30, 63):
        a = np.ascontiguousarray(pairs[k][0].T.cpu().numpy()); b = np.ascontiguousarray(pairs[k][1].T.cpu().numpy())
        assert int((np.abs(a).max(1) == 0).sum()) > 4000                 # the invalid returns are still exact-zero rows after the rotation
        g = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
        assert np.array_equal(batch[k, :6], g["X"]) and np.array_equal(batch[k, 6:12], g["pred_stds"]) and np.array_equal(batch[k, 12:], g["cov"].reshape(36)), k
        ref = po.solve(a, b, trace=True)
        t, ax = ref["trace"], g["aux"]
        f = t["has_fit"] == 1
        assert np.array_equal(ax["n1_raw"], t["n1_raw"]) and np.array_equal(ax["cluster_bounds"], t["bounds"]) and np.array_equal(ax["has_fit"], t["has_fit"]), k
        for name_g, name_o in (("mu1", "mu1"), ("sigma1", "sigma1"), ("evecs1", "evecs1"), ("l_diag", "Ldiag")):
            assert np.array_equal(ax[name_g][f].view(np.uint32), t[name_o][f].view(np.uint32)), (k, name_g)
        act = f & (t["n1_raw"] > 25) & (t["bounds"][:, 5] > 1)
        assert np.array_equal(ax["n2_raw"][0][act], t["n2_raw"][0][act]), k
        # a real scan turned by an arbitrary small rotation is a less forgiving input than the fixtures as given (rings no longer aligned with the grid, ~100 active
        # voxels): held to the larger of the parity bounds and 1x the oracle's own answer-to-answer spread under a 1-ulp perturbation of both scans
        sens = oracle_sensitivity(a, b, scan1_too=True)
        dt, dr = float(np.abs(g["X"][:3] - ref["X"][:3]).max()), float(np.abs(g["X"][3:] - ref["X"][3:]).max())
        # (round 6) the two non-chaotic checks behind the spread bound: every iteration replayed from the oracle's state -- same decisions, H^T W H to 2e-3 -- and the
        # ATTRIBUTION: against the oracle run with the device's documented deviations (scan 2 not round-tripped, FMA transform, moments about mu1:
        # ICET_ORACLE_DEVICE_ARITH) the millimetres are gone (measured: pair 0 2.3e-3 m -> 5e-8 m)
        rep = _replay_each_iteration(gpu_ctx, a, b, ref)
        attr = po.solve(a, b, mode=po.DEVICE_ARITH)
        da = float(np.abs(g["X"][:3] - attr["X"][:3]).max()), float(np.abs(g["X"][3:] - attr["X"][3:]).max())
        print("real batch pair %d: |dX| %.3g m %.3g rad (oracle 1-ulp spread, 32 trials: max %.3g m %.3g rad, p90 %.3g m); replayed per iteration: count flips %s, dH %s; against the oracle in device arithmetic: %.3g m %.3g rad"
              % (k, dt, dr, sens[:3].max(), sens[3:].max(), oracle_sensitivity.last["p90"][:3].max(), [f for f, _ in rep], ["%.1e" % h for _, h in rep], da[0], da[1]))
        assert max(f for f, _ in rep) <= 2 and max(h for _, h in rep) <= 2e-3, rep
        assert da[0] <= TOL_T / 10 and da[1] <= TOL_R / 10, da
        _check_solution(g, ref, tol_t=max(TOL_T, HATCH * float(sens[:3].max())), tol_r=max(TOL_R, HATCH * float(sens[3:].max())), rtol_std=RTOL_STD_LOOSE, rtol_cov=RTOL_COV_LOOSE)


def test_coarse_grid_long_range_takes_the_wide_fixed_point_path(gpu_ctx):
    """the two-instruction float -> fixed-point conversion of k_gn_accumulate holds for |v| < 2^15 m^2 only.  A
    4 x 2 grid (90-degree voxels) over ranges of 150-220 m puts single squared distances to mu1 above 10^4 m^2 and 4-point partial
    sums above 2^15: such flushes must take the exact wide conversion.  Decisions (per-voxel counts of every iteration) must equal
    the oracle's, the first update and X must agree -- a corrupted accumulator would move them by metres."""
    from icet_amd import lidar_sim as ls
    from oracle import pyoracle as po
    s1, s2, _ = ls.make_pair(rings=32, steps=1024)
    a = (s1.T.numpy() * 40.0).astype(np.float32); b = (s2.T.numpy() * 40.0).astype(np.float32)
    kw = dict(runlen=5, bins_phi=2, bins_theta=4, n=25, thresh=4.0, buff=4.0)
    ref = po.solve(a, b, trace=True, **kw)
    t = ref["trace"]; f = t["has_fit"] == 1
    sph = po.c2s(b); vox = po.voxel_of(sph, 2, 4)
    d2max = 0.0
    for v in np.nonzero(f)[0]:
        m = (vox == v) & (sph[:, 0] >= t["bounds"][v, 4]) & (sph[:, 0] <= t["bounds"][v, 5])
        if m.any():
            d2max = max(d2max, float(np.square(b[m] - t["mu1"][v]).max()))
    assert d2max > 8192.0 + 1.0, d2max                     # four such points in one lane overflow the fast conversion's range (2^15)
    g = gpu_ctx.solve(a, b, kw["runlen"], np.zeros(6), 2, 4, 25, 4.0, 4.0, aux=True)
    ax = g["aux"]
    assert np.array_equal(ax["cluster_bounds"], t["bounds"]) and np.array_equal(ax["has_fit"], t["has_fit"])
    assert np.array_equal(ax["mu1"][f].view(np.uint32), t["mu1"][f].view(np.uint32))
    act = f & (t["n1_raw"] > 25) & (t["bounds"][:, 5] > 1)
    assert np.array_equal(ax["n2_in"][0][act], t["n2_in"][0][act]) and int(t["n2_in"][0][act].max()) > 100
    h0 = t["HTWH"][0]; sc = np.sqrt(np.abs(np.diag(h0))) + 1e-30
    assert (np.abs(ax["htwh"][0] - h0) / np.outer(sc, sc)).max() < 2e-3
    assert np.abs(ax["x_hist"][0] - t["X"][0]).max() < 40 * 5e-6 + 1e-5
    sens = oracle_sensitivity(a, b, **kw)
    assert np.abs(g["X"][:3] - ref["X"][:3]).max() <= max(40 * TOL_T, HATCH * sens[:3].max()) and np.abs(g["X"][3:] - ref["X"][3:]).max() <= max(TOL_R, HATCH * sens[3:].max()), (g["X"], ref["X"], sens)


def test_scan2_round_trip_option(gpu_ctx, frames):
    """ICET_FLAG_ROUNDTRIP_SCAN2 restores the reference's two spherical round trips of scan 2 (src/icet.cpp:275, :303), the device's one
    documented deviation in the loop, as a parity-study option (about twice the loop time).  Checked here: the round-tripped copy of scan 2
    is BIT FOR BIT sphericalToCartesian(cartesianToSpherical(.)) under the shared rule (NumPy restatement on the oracle's angles, zero rows
    of a real scan included); decisions are untouched (same keyframe, same first-iteration counts); the result stays within the parity
    bounds; on bench pair 18 -- where the skipped trips ARE the difference to the oracle (9.6e-5 m) -- the flag brings the device to
    within 1e-5 m.  It is not a cure-all and the test does not claim one: over the 256 bench pairs it repairs pair 18, leaves 176 / 121 / 39
    where they were and opens pair 189 (profiles/r03_diag_rt2_all.txt) -- those differences are the last bit of a single-ring voxel's mean,
    which the round trip is only one of several things to flip."""
    import icet_amd
    from icet_amd import lidar_sim as ls, api
    from oracle import pyoracle as po

    def numpy_round_trip(p):
        sph = po.c2s(p); r, th, ph = sph[:, 0], sph[:, 1], sph[:, 2]
        sp = np.sin(ph.astype(np.float64)).astype(np.float32); cp = np.cos(ph.astype(np.float64)).astype(np.float32)
        st = np.sin(th.astype(np.float64)).astype(np.float32); ct = np.cos(th.astype(np.float64)).astype(np.float32)
        return np.stack([(r * sp) * ct, (r * sp) * st, r * cp], 1).astype(np.float32)

    dev = torch.device("cuda", 0)
    s1, s2, _ = ls.make_batch_pair(18, device=dev)
    a, b = s1.T.cpu().numpy(), s2.T.cpu().numpy()
    ref = po.solve(a, b)
    ctx = icet_amd.Context(0)
    g0 = ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
    g1 = ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True, flags=api.FLAG_ROUNDTRIP_SCAN2)
    n = b.shape[0]; ld = (n + 63) // 64 * 64
    og = ctx.debug_fetch("rt2", 3 * ld).reshape(3, ld)[:, :n].T
    assert np.array_equal(og.view(np.uint32), numpy_round_trip(b).view(np.uint32)) and int((og != b).any(1).sum()) > n // 2
    d0 = np.abs(g0["X"][:3] - ref["X"][:3]).max(); d1 = np.abs(g1["X"][:3] - ref["X"][:3]).max()
    print("pair 18: |dX_t| %.3g m without, %.3g m with the round trips" % (d0, d1))
    assert np.array_equal(g0["aux"]["cluster_bounds"], g1["aux"]["cluster_bounds"]) and np.array_equal(g0["aux"]["n2_in"][0], g1["aux"]["n2_in"][0])
    assert d1 <= 1e-5 and d0 <= 1e-5      # (until round 5 d0 was 9.6e-5 m on this pair and the trips closed it to 2.5e-6: with W taken as the reference takes it -- float COD -- the
                                           # pair agrees to 1.2e-6 m WITHOUT the trips: what looked like the round trips' doing was mostly the double-precision W)
    # a real scan with thousands of exact zero rows (the sentinel branch of the round trip: r = 0, phi = NaN -> 1000), and the batched pre-pass
    fa, fb = frames
    r = ctx.solve(fa, fb, 7, np.zeros(6), 24, 75, flags=api.FLAG_ROUNDTRIP_SCAN2)
    n = fb.shape[0]; ld = (n + 63) // 64 * 64
    og = ctx.debug_fetch("rt2", 3 * ld).reshape(3, ld)[:, :n].T
    assert np.array_equal(og.view(np.uint32), numpy_round_trip(fb).view(np.uint32)) and int((fb == 0).all(1).sum()) > 1000
    _check_solution(r, po.solve(fa, fb))
    pairs = [ls.make_batch_pair(k, device=dev) for k in (18, 39)]
    d1 = [(p[0].data_ptr(), p[0].shape[1], p[0].shape[1]) for p in pairs]; d2 = [(p[1].data_ptr(), p[1].shape[1], p[1].shape[1]) for p in pairs]
    out = torch.zeros((2, 48), dtype=torch.float32, device=dev)
    ctx.solve_batch_device(d1, d2, api.Params(7, 24, 75, 25, 0.1, 0.1, api.FLAG_ROUNDTRIP_SCAN2), out.data_ptr()); ctx.sync()
    for j, p in enumerate(pairs):
        pa, pb = p[0].T.cpu().numpy(), p[1].T.cpu().numpy()
        one = ctx.solve(pa, pb, 7, np.zeros(6), 24, 75, flags=api.FLAG_ROUNDTRIP_SCAN2)
        assert np.array_equal(out[j, :6].cpu().numpy(), one["X"])
        loose = j == 1                                       # bench pair 39: the named pred_stds / cov exception
        _check_solution(one, po.solve(pa, pb), rtol_std=RTOL_STD_LOOSE if loose else RTOL_STD, rtol_cov=RTOL_COV_LOOSE if loose else RTOL_COV)
    ctx.close()


def test_rarely_taken_paths_give_the_same_bits(gpu_ctx, frames, sample_pc):
    """Every launch-shape knob selects a code path that exists for unusual inputs; on ordinary inputs they must all produce
    the same bits as the default: accumulator rows that do not fit LDS (HBM-atomic spill path), rank-sort buckets that do not
    fit LDS (global-scratch radix sort), buckets whose keys pile up in one cell of the counting sort (LDS radix sort), the swap-loop
    bit table read from memory instead of LDS (scans above ~0.75 M rows), the executed-step bits from the per-pair recurrence kernel (the
    kernel of throughput batches) and from the chain walks (small batches), small keyframe tiles, few / many accumulate blocks, and the
    library (rocPRIM) sort that the hand-written rank sort replaced (diagnostic builds only); the stable multi-splits with ranks from ballots instead of the values the
    LDS atomics hand back (the default on a device that passed the order self-test -- which an MI355X must)."""
    a, b = frames; c, d = sample_pc
    assert gpu_ctx.debug_fetch("lds_rank_ok", 1)[0] == 1, "LDS atomics of one wave not served in lane order on this device?"
    base1 = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
    base2 = gpu_ctx.solve(c, d, 7, np.zeros(6), 48, 150)
    knobs = [("lds_slots", 32, 0), ("rs_cap", 128, 0), ("rs_max_cell", 0, 24), ("rs_max_cell", 1, 24), ("exec_bits_lds", 0, 1), ("lds_rank", 0, -1), ("exec_pairwise", 1, -1), ("exec_pairwise", 0, -1), ("kf_pts", 1, 8), ("kf_pts", 3, 8), ("acc_blocks", 7, 1536),
             ("acc_pts", 64, 4), ("batch_stage", 0, 4)]
    import icet_amd
    try:                                                     # the rocPRIM A/B backend exists only in a diagnostic build (make EXTRA=-DICET_DIAG_LIBSORT): the shipped library has one sort
        gpu_ctx.set_option("library_sort", 1)
        knobs.append(("library_sort", 1, 0))
    except icet_amd.IcetError as e:
        assert e.status == icet_amd.api.ICET_ERR_UNSUPPORTED
    gpu_ctx.set_option("library_sort", 0)
    for key, val, default in knobs:
        gpu_ctx.set_option(key, val)
        try:
            r1 = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
            r2 = gpu_ctx.solve(c, d, 7, np.zeros(6), 48, 150)
        finally:
            gpu_ctx.set_option(key, default)
        assert np.array_equal(r1["X"], base1["X"]) and np.array_equal(r1["pred_stds"], base1["pred_stds"]), (key, val)
        assert np.array_equal(r1["aux"]["n2_in"], base1["aux"]["n2_in"]) and np.array_equal(r1["aux"]["htwh"], base1["aux"]["htwh"]), (key, val)
        assert np.array_equal(r1["aux"]["sigma1"], base1["aux"]["sigma1"]) and np.array_equal(r1["aux"]["l_diag"], base1["aux"]["l_diag"]), (key, val)
        assert np.array_equal(r2["X"], base2["X"]), (key, val)
    # wider guard bands / a coarser polar table send more rows and points through the literal path and must not change a single
    # decision, in the keyframe or in the loop (the tables are rebuilt by the next call)
    import icet_amd
    for key, val in (("guard_scale", 16), ("lut_polar_quantile", 0.6), ("lut_polar_quantile", 0.0)):
        ctx = icet_amd.Context(0)
        ctx.set_option(key, val)
        k1 = ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
        assert np.array_equal(k1["aux"]["n1_raw"], base1["aux"]["n1_raw"]) and np.array_equal(k1["aux"]["sigma1"], base1["aux"]["sigma1"]), (key, val)   # keyframe: same bits
        assert np.array_equal(k1["aux"]["l_diag"], base1["aux"]["l_diag"]) and np.array_equal(k1["aux"]["cluster_bounds"], base1["aux"]["cluster_bounds"]), (key, val)
        # loop: a point that changes from the fast to the literal path enters the sums as a run of one instead of inside its lane's
        # run, so the float partial sums round differently; the DECISIONS must not change: every iteration replayed from the same X
        for it in range(7):
            x0 = np.zeros(6, np.float32) if it == 0 else base1["aux"]["x_hist"][it - 1]
            r1 = ctx.solve(a, b, 1, x0, 24, 75, aux=True)
            assert np.array_equal(r1["aux"]["n2_raw"][0], base1["aux"]["n2_raw"][it]) and np.array_equal(r1["aux"]["n2_in"][0], base1["aux"]["n2_in"][it]), (key, val, it)
        ctx.close()
        assert np.abs(k1["X"][:3] - base1["X"][:3]).max() <= TOL_T and np.abs(k1["X"][3:] - base1["X"][3:]).max() <= TOL_R, (key, val)
    with pytest.raises(icet_amd.IcetError):
        gpu_ctx.set_option("no_such_knob", 1)


def test_true_sort_extension_matches_its_oracle_twin(gpu_ctx, frames, sample_pc):
    """ICET_FLAG_TRUE_SORT is a labelled NON-PARITY extension (rows really sorted by range before clustering).  It is still held to
    a CPU twin: the oracle with ICET_ORACLE_TRUE_SORT -- keyframe decisions bit-exact, final X within the usual tolerance."""
    from oracle import pyoracle as po
    from icet_amd import api
    for a, b in (frames, sample_pc):
        g = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True, flags=api.FLAG_TRUE_SORT)
        o = po.solve(a, b, trace=True, mode=po.TRUE_SORT)
        n = a.shape[0]
        r = po.c2s(a)[:, 0]
        assert np.array_equal(gpu_ctx.debug_fetch("src", n), np.argsort(r, kind="stable"))      # src = the stable sorted order itself
        assert np.array_equal(g["aux"]["n1_raw"], o["trace"]["n1_raw"]) and np.array_equal(g["aux"]["cluster_bounds"], o["trace"]["bounds"])
        assert np.array_equal(g["aux"]["has_fit"], o["trace"]["has_fit"])
        f = o["trace"]["has_fit"] == 1
        assert np.array_equal(g["aux"]["evecs1"][f].view(np.uint32), o["trace"]["evecs1"][f].view(np.uint32))
        _check_solution(dict(X=g["X"], pred_stds=g["pred_stds"], cov=g["cov"]), o)
        plain = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
        assert g["aux"]["has_fit"].sum() > 2 * plain["aux"]["has_fit"].sum()                   # and it is a different answer by design


def test_half_gap_bounds_extension_matches_its_oracle_twin(gpu_ctx, frames, sample_pc):
    """ICET_FLAG_HALF_GAP_BOUNDS (the cluster buffers of the Python variant, python/utils.py:92-119) is a labelled NON-PARITY
    extension on top of the really-sorted rows.  Held to its CPU twin (ICET_ORACLE_HALF_GAP): cluster bounds and the whole keyframe table
    bit-exact, X within the usual tolerance; and it does what it says -- no bound is wider than with the fixed buffer, some are tighter."""
    from oracle import pyoracle as po
    from icet_amd import api
    for a, b in (frames, sample_pc):
        g = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True, flags=api.FLAG_HALF_GAP_BOUNDS)
        o = po.solve(a, b, trace=True, mode=po.HALF_GAP)
        assert np.array_equal(g["aux"]["n1_raw"], o["trace"]["n1_raw"]) and np.array_equal(g["aux"]["cluster_bounds"], o["trace"]["bounds"])
        assert np.array_equal(g["aux"]["has_fit"], o["trace"]["has_fit"])
        f = o["trace"]["has_fit"] == 1
        for name_g, name_o in (("mu1", "mu1"), ("sigma1", "sigma1"), ("evecs1", "evecs1"), ("l_diag", "Ldiag")):
            assert np.array_equal(g["aux"][name_g][f].view(np.uint32), o["trace"][name_o][f].view(np.uint32)), name_g
        _check_solution(dict(X=g["X"], pred_stds=g["pred_stds"], cov=g["cov"]), o)
        srt = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True, flags=api.FLAG_TRUE_SORT)       # same clusters, fixed buffers
        bh, bs = g["aux"]["cluster_bounds"], srt["aux"]["cluster_bounds"]
        has = bs[:, 5] > 0
        assert np.array_equal(bh[:, 5] > 0, has)
        assert (bh[has, 4] >= bs[has, 4]).all() and (bh[has, 5] <= bs[has, 5]).all()
        assert (bh[has, 4] > bs[has, 4]).any() or (bh[has, 5] < bs[has, 5]).any()


# ---- ICET::checkCondition's ill-conditioned route (src/icet.cpp:410-430, 443-492): review r4, row a15 -----------------------------------

def _cond_sweep_matrices():
    """(HTWH, HTWdz) pairs whose condition number crosses checkCondition's cutoff: the oracle's per-iteration matrices of the degenerate
    golden scenes, rescaled along their weakest eigenvector so that cond sweeps [5e4, 3e7] (where pruning starts, and where the rank
    threshold of the pseudo-inverse, 1 / (6 eps) = 1.4e6, sits), plus rank-deficient, zero, NaN and well-conditioned ones."""
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_degenerate.npz")))
    names = sorted({k.split("/")[0] for k in g})
    Hs, gs = [], []
    conds = np.geomspace(5e4, 3e7, 140)                                 # crosses the Cholesky route's bound (2.5e5), checkCondition's cutoff (1e6) and the rank threshold (1.4e6)
    for nm in names:
        for it in (0, 3, 6):
            H = g[nm + "/HTWH"][it].astype(np.float64); gv = g[nm + "/HTWdz"][it]
            Hs.append(H.astype(np.float32)); gs.append(gv)
            w, Q = np.linalg.eigh(H)
            for c in conds:
                w2 = w.copy(); w2[0] = w[5] / c
                H2 = ((Q * w2) @ Q.T).astype(np.float32)
                Hs.append((H2 + H2.T) / 2); gs.append(gv)
            for drop in (1, 2, 3, 5):                                   # exactly rank-deficient
                w2 = w.copy(); w2[:drop] = 0.0
                H2 = ((Q * w2) @ Q.T).astype(np.float32)
                Hs.append((H2 + H2.T) / 2); gs.append(gv)
    rng = np.random.default_rng(5)
    for _ in range(64):                                                 # well conditioned
        A = rng.standard_normal((6, 12)); Hs.append((A @ A.T).astype(np.float32)); gs.append(rng.standard_normal(6).astype(np.float32))
    Hs.append(np.zeros((6, 6), np.float32)); gs.append(np.zeros(6, np.float32))                  # no voxel matched: HTWH = 0 (cond = NaN: no pruning)
    Hs.append(np.zeros((6, 6), np.float32)); gs.append(np.ones(6, np.float32))
    Hn = np.eye(6, dtype=np.float32); Hn[2, 2] = np.nan; Hs.append(Hn); gs.append(np.ones(6, np.float32))
    Hs.append(np.full((6, 6), np.nan, np.float32)); gs.append(np.ones(6, np.float32))
    Hs.append(np.diag([1, 1, 1, 1, 1, -1e-3]).astype(np.float32)); gs.append(np.ones(6, np.float32))    # indefinite
    return np.stack(Hs), np.stack(gs)


def _same_bits(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def test_gn_tail_literal_bits(gpu_ctx):
    """The device's restatement of src/icet.cpp:410-430 (pinv by column-pivoted QR, pred_stds, checkCondition's pruning with
    `pred_stds += U2.col(k)`, dx through the pseudo-inverse of L2 lam U2^T) against the oracle's, on the SAME matrices: every output BIT FOR
    BIT -- rank decision, pruned-axis count and eigenvector signs included -- over condition numbers 3e5 .. 3e7, rank-deficient, zero and NaN
    input.  (The rule of rounds 2-4, |lambda_k| > 6 eps lambda_max, disagrees with the pivot rule on ~0.4 % of such matrices:
    scripts/rank_rule_study.py.)  Then the default routing: a matrix sent down the Cholesky route is one the oracle does not prune, and its
    inverse agrees to rounding x condition number."""
    import icet_amd
    from oracle import pyoracle as po
    H, g = _cond_sweep_matrices()
    ctx = icet_amd.Context(0)
    ctx.set_option("gn_cond_bound", 0)                                  # everything through the literal route
    dev = ctx.debug_gn_tail(H, g)
    n_pruned = np.zeros(7, int); n_def = 0
    for i in range(H.shape[0]):
        ref = po.gn_tail(H[i], g[i])
        assert dev["route"][i] == 2
        assert dev["pruned"][i] == ref["pruned"], (i, dev["pruned"][i], ref["pruned"])
        for k in ("cov", "pred_stds", "dx", "eigvals"):
            assert _same_bits(dev[k][i], ref[k]), (i, k, dev[k][i], ref[k])
        n_pruned[ref["pruned"]] += 1; n_def += ref["rank"] < 6
    print("gn_tail literal route: %d matrices bit-identical to the oracle; pruned-axis histogram %s; %d rank-deficient by the pivot rule" % (H.shape[0], n_pruned.tolist(), n_def))
    assert n_pruned[1] > 100 and n_pruned[2] > 10 and n_pruned[3] > 10 and n_def > 100      # the sweep does exercise what it claims to
    ctx.set_option("gn_cond_bound", 2.5e5)                              # the default routing (a factor 4 below the cutoff)
    dflt = ctx.debug_gn_tail(H, g)
    n_chol = 0
    for i in range(H.shape[0]):
        ref = po.gn_tail(H[i], g[i])
        if dflt["route"][i] == 0:
            n_chol += 1
            assert ref["pruned"] == 0 and ref["rank"] == 6 and dflt["pruned"][i] == 0
            cond = ref["eigvals"][5] / ref["eigvals"][0]
            d = np.sqrt(np.abs(np.diag(ref["cov"])))
            assert (np.abs(dflt["cov"][i] - ref["cov"]) <= 64 * cond * 6e-8 * np.outer(d, d)).all(), (i, cond)
        else:
            for k in ("cov", "pred_stds", "dx", "eigvals"):
                assert _same_bits(dflt[k][i], ref[k]), (i, k)
    assert n_chol >= 64
    ctx.close()


def _degenerate_golden():
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_degenerate.npz")))
    return {nm: {k.split("/")[1]: v for k, v in g.items() if k.startswith(nm + "/")} for nm in sorted({k.split("/")[0] for k in g})}


def _degenerate_scans(name):
    from icet_amd import lidar_sim as ls
    a, b, _ = ls.make_degenerate_named(name)
    return np.ascontiguousarray(a.T.numpy()), np.ascontiguousarray(b.T.numpy())


def _oracle_spread(a, b, ref, trials=32):
    """How far the ORACLE's own per-iteration tables move under a 1-ulp perturbation of scan 2 (relative 1e-7): on scenes with millimetre noise the
    thin direction of a voxel's covariance is a few float ulps of the coordinates wide, H^T W H moves by per cents and the SIGN of the pruned
    eigenvector -- hence of the +-1 added to pred_stds -- can flip (measured, profiles/r05_diag_degenerate.txt).  32 trials on the host's cores
    (round 6; three were a lower bound of a heavy-tailed spread): maxima, and the 90 % quantile of the H^T W H spread."""
    from oracle import pyoracle as po
    from concurrent.futures import ThreadPoolExecutor
    t = ref["trace"]
    rng = np.random.default_rng(123)
    perturbed = [(b.astype(np.float64) * (1.0 + rng.uniform(-1e-7, 1e-7, b.shape))).astype(np.float32) for _ in range(trials)]
    with ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as ex:          # (the C call releases the GIL)
        sols = list(ex.map(lambda bp: po.solve(a, bp, trace=True), perturbed))
    Hs = [float((np.abs(o2["trace"]["HTWH"] - t["HTWH"]).reshape(7, -1).max(1) / np.abs(t["HTWH"]).reshape(7, -1).max(1)).max()) for o2 in sols]
    return dict(H=max(Hs), H_p90=float(np.quantile(Hs, 0.9)), ps=max(float(np.abs(o2["pred_stds"] - ref["pred_stds"]).max()) for o2 in sols),
                Xt=max(float(np.abs(o2["X"][:3] - ref["X"][:3]).max()) for o2 in sols), Xr=max(float(np.abs(o2["X"][3:] - ref["X"][3:]).max()) for o2 in sols),
                pruned_same=all(bool(np.array_equal(o2["trace"]["pruned"], t["pruned"])) for o2 in sols), trials=trials)


@pytest.mark.parametrize("name", ["tunnel_s05", "tunnel_s10", "tunnel_s10_m", "wall_s10", "wall_s30", "ground_s02", "ground_s05", "ground_s10_m"])
def test_degenerate_scenes_take_the_pruning_route(gpu_ctx, name):
    """Tunnel / single wall / open ground (icet_amd/lidar_sim.DEGENERATE_SCENES): HTWH has condition numbers of 1e6 .. 1e9, the reference
    prunes one to three solution axes and ADDS THE PRUNED EIGENVECTOR to pred_stds (src/icet.cpp:479) -- pred_stds[0] = +1.0 in
    tunnel_s05, -1.0 in tunnel_s10.  Two layers:
    (1) EXACT: inside the real solve, every iteration's 6x6 tail is the oracle's function of the device's own (HTWH_i, HTWdz_i): the pruned-axis
        count, the eigenvalues checkCondition saw, X_i = X_{i-1} + dx and the final pred_stds / covariance are the oracle's gn_tail of the
        device's tables BIT FOR BIT -- sign of the pruned eigenvector included.
    (2) against the oracle's solve of the same scans, where the two sides' HTWH differ by the usual last bits of the per-voxel moments:
        the keyframe table bit-exact; X within the usual bounds or 1x the oracle's own 1-ulp sensitivity; where the oracle's answer is itself
        stable under a 1-ulp perturbation of scan 2 (its pruned counts do not change / its pred_stds move by < 1e-3), the device prunes the same
        axes in every iteration and pred_stds agree INCLUDING SIGN to 1e-3 absolute; on the others (millimetre-noise scenes at the resolution of
        float32 coordinates: the oracle flips the sign of its own +-1.0 under that perturbation) the difference is held to the oracle's spread."""
    from oracle import pyoracle as po
    from icet_amd import api
    a, b = _degenerate_scans(name)
    gold = _degenerate_golden()[name]
    assert gold["checksum"][0] == a.shape[0] and gold["checksum"][1] == b.shape[0] and np.isclose(gold["checksum"][4], np.abs(a.astype(np.float64)).sum(), rtol=1e-12), "scene generator drifted"
    ref = po.solve(a, b, trace=True)
    t = ref["trace"]
    assert np.array_equal(t["pruned"], gold["pruned"]) and np.allclose(ref["pred_stds"], gold["pred_stds"], rtol=1e-4, atol=1e-6)
    r = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True)
    ax = r["aux"]
    assert np.array_equal(ax["cluster_bounds"], t["bounds"]) and np.array_equal(ax["has_fit"], t["has_fit"])
    f = t["has_fit"] == 1
    assert np.array_equal(ax["evecs1"][f].view(np.uint32), t["evecs1"][f].view(np.uint32))
    ci = ax["cond_info"]
    # (1) the tail inside the solve kernel == the oracle's tail of the device's own tables, bit for bit
    xprev = np.zeros(6, np.float32)
    for it in range(7):
        tail = po.gn_tail(ax["htwh"][it], ax["htwdz"][it])
        assert int(ci[it, 7]) == 2, "a degenerate scene must leave the Cholesky route"
        assert int(ci[it, 6]) == tail["pruned"], (it, ci[it], tail["pruned"])
        assert _same_bits(ci[it, :6], tail["eigvals"]), (it, ci[it, :6], tail["eigvals"])
        assert _same_bits(ax["x_hist"][it], (xprev + tail["dx"]).astype(np.float32)), (it, ax["x_hist"][it], xprev + tail["dx"])
        xprev = ax["x_hist"][it]
    assert _same_bits(r["pred_stds"], tail["pred_stds"]) and _same_bits(r["cov"], tail["cov"]), (r["pred_stds"], tail["pred_stds"])
    # (2) against the oracle's own solve
    sp = _oracle_spread(a, b, ref)
    dH = float((np.abs(ax["htwh"] - t["HTWH"]).reshape(7, -1).max(1) / np.abs(t["HTWH"]).reshape(7, -1).max(1)).max())
    dps = float(np.abs(r["pred_stds"] - ref["pred_stds"]).max())
    dt, dr = float(np.abs(r["X"][:3] - ref["X"][:3]).max()), float(np.abs(r["X"][3:] - ref["X"][3:]).max())
    print("%s: pruned oracle %s device %s | pred_stds oracle %s device %s | dH %.2e (oracle 1-ulp spread %.2e) d pred_stds %.2e (%.2e) |dX| %.2e m %.2e rad (%.2e / %.2e) oracle pruning stable: %s"
          % (name, t["pruned"].tolist(), ci[:, 6].astype(int).tolist(), np.round(ref["pred_stds"], 4).tolist(), np.round(r["pred_stds"], 4).tolist(), dH, sp["H"], dps, sp["ps"], dt, dr, sp["Xt"], sp["Xr"], sp["pruned_same"]))
    # (round 6) H^T W H against the UNMODIFIED oracle: within 1.5 x the oracle's own 32-trial spread, and every iteration replayed from the oracle's state takes the
    # oracle's decisions.  Until round 5 this stood at 0.14 (wall_s10) / 0.17 (ground_s10_m) against spreads of 0.09 / 0.07; the ATTRIBUTION (scripts/diag_attribution.py,
    # profiles/r06_attribution.txt): the per-voxel weight W = pinv(R_noise) was taken in double, while the reference's float CompleteOrthogonalDecomposition of a cond
    # 1e6 .. 1e7 voxel carries a relative error of cond x eps and decides its rank by pivots.  The device now runs the reference's float COD (bit-identical to the
    # restatement's function: test_pinv3_reference_bits); the double path is ICET_FLAG_DOUBLE_W, shown below against the oracle run the same way.
    rep_o = _replay_each_iteration(gpu_ctx, a, b, ref)
    r2 = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True, flags=api.FLAG_DOUBLE_W)
    dH2 = float((np.abs(r2["aux"]["htwh"] - t["HTWH"]).reshape(7, -1).max(1) / np.abs(t["HTWH"]).reshape(7, -1).max(1)).max())
    ta = po.solve(a, b, trace=True, mode=po.DEVICE_ARITH | po.PINV3_DOUBLE)["trace"]
    dH2_attr = float((np.abs(r2["aux"]["htwh"] - ta["HTWH"]).reshape(7, -1).max(1) / np.abs(ta["HTWH"]).reshape(7, -1).max(1)).max())
    print("   %s: dH %.2e (oracle 32-trial spread max %.2e p90 %.2e); replayed from the oracle's state: count flips %s dH %s | with ICET_FLAG_DOUBLE_W: dH %.2e, against the oracle in device arithmetic + double W %.2e, pruned %s"
          % (name, dH, sp["H"], sp["H_p90"], [f for f, _ in rep_o], ["%.1e" % h for _, h in rep_o], dH2, dH2_attr, r2["aux"]["cond_info"][:, 6].astype(int).tolist()))
    assert max(f for f, _ in rep_o) == 0, rep_o                            # one step from the same state: the same decisions
    assert dH <= max(2e-3, HATCH * sp["H"]), (dH, sp["H"])
    assert float(np.median([h for _, h in rep_o])) <= max(2e-3, sp["H"]), (rep_o, sp["H"])
    assert dH2_attr <= 0.1, (dH2, dH2_attr)
    cond_o = np.abs(t["eigvals"][:, 5] / t["eigvals"][:, 0])
    straddles = bool(((cond_o > 1e6 / 1.5) & (cond_o < 1.5e6)).any())      # a condition number within the tables' own spread of the cutoff: pruning there is a coin toss on either side
    if straddles and not np.array_equal(ci[:, 6].astype(int), t["pruned"]):
        # (ground_s10_m: cond 1.1e6.)  The two sides pruned the weakest axis in different iterations: what can be compared is X off that axis
        assert np.abs(ci[:, 6].astype(int) - t["pruned"]).max() <= 1
        w, Q = po.eig_sym(t["HTWH"][-1])
        dX = (r["X"] - ref["X"]).astype(np.float64)
        dX_kept = dX - Q[:, 0].astype(np.float64) * (Q[:, 0].astype(np.float64) @ dX)
        assert np.abs(dX_kept[:3]).max() <= max(TOL_T, sp["Xt"]) and np.abs(dX_kept[3:]).max() <= max(TOL_R, sp["Xr"]), (dX, dX_kept)
        print("   %s straddles the cutoff (oracle cond %s): pruning differs by iteration; |dX| off the weakest axis %.2e m %.2e rad" % (name, np.round(cond_o).tolist(), np.abs(dX_kept[:3]).max(), np.abs(dX_kept[3:]).max()))
        return
    assert dt <= max(TOL_T, sp["Xt"]) and dr <= max(TOL_R, sp["Xr"]), (dt, dr, sp)
    if sp["pruned_same"]:
        assert np.array_equal(ci[:, 6].astype(int), t["pruned"]), (ci[:, 6], t["pruned"])
    else:
        assert np.abs(ci[:, 6].astype(int) - t["pruned"]).max() <= 1
    if sp["ps"] < 1e-3:
        assert dps <= 1e-3, (r["pred_stds"], ref["pred_stds"])            # signs included: negative entries are the reference's behaviour
    else:
        # an unstable scene: the oracle's own pred_stds moves under the 1-ulp perturbation, and the SIGN of a pruned axis' +-1 (the eigen-solver's last bits) is then a
        # coin toss that three trials may or may not show -- layer (1) above pins it exactly to the device's own matrix; here the magnitudes are compared
        dabs = float(np.abs(np.abs(r["pred_stds"]) - np.abs(ref["pred_stds"])).max())
        assert dps <= HATCH * sp["ps"] + 1e-3 or dabs <= HATCH * sp["ps"] + 1e-3, (dps, dabs, sp["ps"], r["pred_stds"], ref["pred_stds"])
    assert name not in ("tunnel_s10", "tunnel_s10_m", "wall_s30") or sp["ps"] < 1e-3          # these three are the stable ones: the tight branch must be the one that ran


def test_degenerate_pairs_inside_a_batch_leave_their_neighbours_alone(gpu_ctx):
    """The same scenes through ONE icet_solve_batch_device call between well-conditioned pairs: a block that takes the literal route must
    not disturb its neighbours' bits, and must give its own single-solve bits."""
    import icet_amd
    from icet_amd import lidar_sim as ls, api
    dev = torch.device("cuda", 0)
    names = ["tunnel_s05", "ground_s02", "wall_s10", "ground_s10_m"]
    pairs = []
    for k in range(4):
        pairs.append(ls.make_batch_pair(k, device=dev)[:2])
        a, b = _degenerate_scans(names[k])
        pairs.append((torch.from_numpy(np.ascontiguousarray(a.T)).to(dev), torch.from_numpy(np.ascontiguousarray(b.T)).to(dev)))
    pairs.append(ls.make_batch_pair(4, device=dev)[:2])
    d1 = [(p[0].data_ptr(), p[0].shape[1], p[0].shape[1]) for p in pairs]; d2 = [(p[1].data_ptr(), p[1].shape[1], p[1].shape[1]) for p in pairs]
    out = torch.zeros((len(pairs), 48), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx = icet_amd.Context(0)
    ctx.solve_batch_device(d1, d2, api.Params(7, 24, 75, 25, 0.1, 0.1, 0), out.data_ptr()); ctx.sync()
    res = out.cpu().numpy()
    ctx.close()
    n_literal = 0
    for k, p in enumerate(pairs):
        single = gpu_ctx.solve(p[0].T.cpu().numpy(), p[1].T.cpu().numpy(), 7, np.zeros(6), 24, 75, aux=True)
        assert np.array_equal(res[k, :6], single["X"]) and np.array_equal(res[k, 6:12], single["pred_stds"]) and np.array_equal(res[k, 12:], single["cov"].reshape(36)), k
        n_literal += int((single["aux"]["cond_info"][:, 7] == 2).any())
        assert (single["aux"]["cond_info"][:, 7] == 2).any() == (k % 2 == 1), k
    assert n_literal == 4


def test_no_matching_voxel_gives_zero_update(gpu_ctx, frames):
    """HTWH = 0 (scan 2 far away from every scan-1 voxel): cond = 0/0 = NaN, so nothing is pruned (src/icet.cpp:469 compares false), the
    pseudo-inverse of the zero matrix is zero: X stays X0, pred_stds = 0 -- on the device as in the oracle."""
    from oracle import pyoracle as po
    a, _ = frames
    b = (a + np.array([500.0, 0, 300.0], np.float32)).astype(np.float32)
    x0 = np.array([0.1, 0, 0, 0, 0, 0.01], np.float32)
    ref = po.solve(a, b, x0=x0, trace=True)
    r = gpu_ctx.solve(a, b, 7, x0, 24, 75, aux=True)
    assert (ref["trace"]["HTWH"] == 0).all() and (r["aux"]["htwh"] == 0).all()
    assert np.array_equal(r["X"], ref["X"]) and np.array_equal(r["X"], x0) and (r["pred_stds"] == 0).all() and (ref["pred_stds"] == 0).all()
    assert (r["aux"]["cond_info"][:, 6] == 0).all() and (r["aux"]["cond_info"][:, 7] == 2).all()


@pytest.mark.parametrize("seed,with_flags", [(1, False), (2, False), (3, False), (4, False), (5, False), (6, False), (11, True), (12, True), (13, True), (14, True), (15, True), (16, True)])
def test_random_parameter_space(gpu_ctx, seed, with_flags):
    """(Twelve seeds in the suite since round 6: 480 draws.)  40 seeded draws from the parameter space (tests/param_sweep.py: grids from 7 x 3 to 199 x 48, minimum points 3 .. 120, thresh 0.02 .. 1, buff 0 .. 2, runlen 1 .. 9,
    zero and non-zero X0, stretches / strides of the sample scans with their zero rows, 3 k .. 131 k rows; 600 draws over five seeds were run by hand,
    scripts/fuzz_params.py -> profiles/r05_fuzz_params.txt).  In EVERY draw the whole keyframe table is the oracle's bits (NaN covariances of one-point clusters in
    the same places) and so are the first iteration's per-voxel counts (X0 = 0; see param_sweep.run_case for X0 != 0).  The solution: within the parity bound, or within
    1.5 x the oracle's own spread under a 1-ulp perturbation of the scans; a draw whose spread exceeds 10 x the parity bound is an ill-posed problem (a handful of voxels,
    an axis at the condition cutoff, pruning that comes and goes: the oracle's own answer moves by centimetres, scripts/fuzz_diag.py -> profiles/r05_fuzz_diag.txt) and
    only its first update is compared.  (Round 6: the spread is the maximum over 2 x 16 trials and the factor is 1.5, and every draw admitted through it is also REPLAYED
    iteration by iteration from the oracle's state.)  with_flags: each draw also picks one of the opt-in extensions (ICET_FLAG_TRUE_SORT, _HALF_GAP_BOUNDS, _REJECT_MOVING) or none,
    and is held to that extension's CPU twin."""
    from tests.param_sweep import draw_case, run_case, pools
    rng = np.random.default_rng(seed)
    pl = pools()
    beyond, ill, replays, nan_both = [], [], [], []
    for c in range(40):
        a, b, T, P, kw, runlen, x0 = draw_case(rng, pl, with_flags=with_flags)
        bits, d, r, ref, fits = run_case(gpu_ctx, a, b, T, P, kw, runlen, x0)
        assert all(bits.values()), (c, T, P, kw, runlen, {k: v for k, v in bits.items() if not v})
        okw = {k: v for k, v in kw.items() if k != "_twin"}
        if kw.get("_twin", (0, None))[1] is not None: okw["mode"] = kw["_twin"][1]
        # (n = 3: a cluster of a few collinear points has a 0 / 0 in its fit, and the reference's loop carries the NaN into X -- src/icet.cpp has no guard; the
        # restatement does the same and the device must agree on WHERE that happens: found by seed 4, draws 10 and 30)
        assert bool(np.isfinite(r["X"]).all()) == bool(np.isfinite(ref["X"]).all()), (c, T, P, kw, runlen, r["X"], ref["X"])
        if not np.isfinite(ref["X"]).all():
            first_bad = int(np.argmax(~np.isfinite(ref["trace"]["X"]).all(1)))
            assert int(np.argmax(~np.isfinite(r["aux"]["x_hist"]).all(1))) == first_bad, (c, "NaN enters in another iteration")
            nan_both.append(c)
            continue
        if d[:3].max() > TOL_T or d[3:].max() > TOL_R:
            okw = dict(x0=x0, runlen=runlen, bins_phi=P, bins_theta=T, **okw)
            sens = np.maximum(oracle_sensitivity(a, b, trials=16, scan1_too=True, **okw), oracle_sensitivity(a, b, trials=16, **okw))
            beyond.append((c, d[:3].max(), d[3:].max(), sens[:3].max(), sens[3:].max()))
            if sens[:3].max() > 10 * TOL_T or sens[3:].max() > 10 * TOL_R:
                ill.append(c)
                d0 = np.abs(r["aux"]["x_hist"][0] - ref["trace"]["X"][0])
                if d0.max() > 1e-3:                                       # (seven used voxels with weights of 1e9: even ONE update moves by a centimetre inside the oracle)
                    o1 = dict(okw, runlen=1)
                    s0 = np.maximum(oracle_sensitivity(a, b, trials=16, scan1_too=True, **o1), oracle_sensitivity(a, b, trials=16, **o1))
                    assert d0.max() <= HATCH * s0.max(), (c, T, P, kw, d0, s0)
            else:
                assert d[:3].max() <= max(TOL_T, HATCH * sens[:3].max()) and d[3:].max() <= max(TOL_R, HATCH * sens[3:].max()), (c, T, P, kw, runlen, d, sens)
            # (round 6) whoever is admitted through the spread also passes the non-chaotic check: every iteration replayed from the oracle's state takes the
            # oracle's decisions (with X != 0 the FMA transform may put a point or two per 100 k across an edge) and builds its H^T W H
            rkw = {k: v for k, v in kw.items() if k != "_twin"}
            rep = _replay_each_iteration(gpu_ctx, a, b, ref, P=P, T=T, x0=x0, flags=kw.get("_twin", (0, None))[0], **rkw)
            replays.append((c, max(f for f, _ in rep), max(h for _, h in rep)))
            assert max(f for f, _ in rep) <= 4, (c, rep)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_param_sweep_%d.txt" % seed), "w") as fh:
            fh.write("# test_random_parameter_space (seed %d, flags %s): 40 draws, keyframe bits equal in all; beyond the parity bound: %d, of them ill-posed: %d\n# case |dX_t| |dX_r| oracle spread t / r\n" % (seed, with_flags, len(beyond), len(ill)))
            for w in beyond:
                fh.write("%d %.3g %.3g %.3g %.3g\n" % w)
            fh.write("# replayed from the oracle's state (draws beyond the bound): case, max raw-count differences per iteration, max rel dHTWH\n")
            for w in replays:
                fh.write("%d %d %.3g\n" % w)
    assert len(beyond) <= 14 and len(ill) <= 12 and len(nan_both) <= 4, (beyond, ill, nan_both)            # the draws are deliberately extreme


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_batches_carry_their_single_solve_bits(gpu_ctx, seed):
    """(Three seeds in the suite since round 6.)  12 seeded random batches (tests/param_sweep.run_batch; 134 more were run by hand, scripts/fuzz_batch.py -> profiles/r05_fuzz_batch.txt): 1 .. 48 ragged pairs --
    empty scans, scans of a few rows, strided and whole real / synthetic scans -- random grid, minimum points, thresh, buff, runlen and X0, once through icet_solve_batch
    and once through icet_solve_batch_device with padded leading dimensions.  Every pair of every batch: the BITS of its own single solve in another context (integer
    accumulation: no dependence on batch size, chunking, launch shape or neighbours)."""
    from icet_amd import api
    from tests.param_sweep import run_batch, pools
    rng = np.random.default_rng(seed)
    pl = pools()
    single = api.Context()
    for _ in range(12):
        desc, diffs = run_batch(gpu_ctx, single, rng, pl)
        assert not diffs, (desc, diffs[:6])
    single.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_adversarial_scans(gpu_ctx, seed):
    """(Four seeds in the suite since round 6: 120 pairs, each on both paths.)  30 seeded adversarial pairs (tests/param_sweep.spoil: thousands of equal sort keys, lattice points EXACTLY on voxel edges, duplicated rows, NaN / +-inf entries,
    rows scaled by 1e-20 / 1e+18, signed zeros; 180 more by hand, scripts/fuzz_adversarial.py -> profiles/r05_fuzz_adversarial.txt).  The keyframe table is the oracle's
    bits in every one.  The first iteration's per-voxel counts of scan 2 are the oracle's in every one when the device round-trips scan 2 through spherical coordinates
    as the reference does (ICET_FLAG_ROUNDTRIP_SCAN2); on the default path -- which skips that round trip, DESIGN.md section 7 -- a fifth of these pairs differ in a few
    counts: a lattice point that sits ON an edge moves off it in the round trip, and a vector whose squares are denormal comes back 1e-5 away from where it was."""
    from icet_amd import api
    from tests.param_sweep import draw_adversarial, run_case, pools
    pl = pools()
    n_default_diff = 0
    for flag in (api.FLAG_ROUNDTRIP_SCAN2, 0):
        rng = np.random.default_rng(seed)
        for c in range(30):
            a, b, T, P, kw, runlen, x0, what = draw_adversarial(rng, pl)
            if flag: kw["_twin"] = (flag, None)
            bits, d, r, ref, fits = run_case(gpu_ctx, a, b, T, P, kw, runlen, x0)
            counts_equal = bits.pop("n2_raw0")
            assert all(bits.values()), (c, what, T, P, kw, {k: v for k, v in bits.items() if not v})
            assert np.isfinite(r["X"]).all() == np.isfinite(ref["X"]).all()
            if flag: assert counts_equal, (c, what, T, P, kw)
            else: n_default_diff += 0 if counts_equal else 1
    assert n_default_diff <= 12


def test_extreme_grids_up_to_the_voxel_limit(gpu_ctx, frames):
    """One voxel, one ring, one column, and the 10 000-voxel limit in every shape -- 200 x 50, 100 x 100, 10 000 x 1, 1 x 10 000, 5000 x 2, 2 x 5000 (scripts/fuzz_grids.py ->
    profiles/r05_fuzz_grids.txt).  A grid that is extremely fine in one direction wants larger look-up tables than a block's LDS holds: until round 5 such a grid failed at the
    first launch with hipErrorInvalidValue; the tables are now capped (coarser cells hold two edges and send their points through the literal formulas).  Keyframe table and
    first-iteration counts: the oracle's bits; one voxel more than the limit, or a non-positive dimension: refused before anything runs."""
    from icet_amd import api
    from tests.param_sweep import run_case
    a, b = frames
    for (T, P) in ((1, 1), (1, 24), (75, 1), (2, 2), (200, 50), (100, 100), (10000, 1), (1, 10000), (5000, 2), (2, 5000), (2500, 4), (8, 1250)):
        bits, d, r, ref, fits = run_case(gpu_ctx, a, b, T, P, dict(n=25, thresh=0.1, buff=0.1), 3, np.zeros(6, np.float32))
        assert all(bits.values()), (T, P, {k: v for k, v in bits.items() if not v})
        assert np.isfinite(r["X"]).all()
    for (T, P, st) in ((10001, 1, "ICET_ERR_UNSUPPORTED"), (101, 100, "ICET_ERR_UNSUPPORTED"), (1, 10001, "ICET_ERR_UNSUPPORTED"), (0, 5, "ICET_ERR_BAD_ARG"), (5, 0, "ICET_ERR_BAD_ARG"), (-1, 3, "ICET_ERR_BAD_ARG")):
        with pytest.raises(api.IcetError) as ei:
            gpu_ctx.solve(a, b, 3, np.zeros(6), P, T)
        assert st in str(ei.value), (T, P, str(ei.value))
    r = gpu_ctx.solve(a, b, 3, np.zeros(6), 24, 75)                     # the context is usable after a refusal
    assert np.isfinite(r["X"]).all()


def test_massive_sort_ties(gpu_ctx):
    """The rank sort under ties by the ten thousand (tests/param_sweep.tie_cases; scripts/fuzz_ties.py -> profiles/r05_fuzz_ties.txt): 1, 5, 38 and 930 distinct rows
    repeated up to 300 000 times, 60 000 unit directions scaled to ONE range, a scan whose first half is one row.  The reference's order among equal ranges is the
    row order (stable sort, oracle header); the swap loop, the clusters and the Gaussians built on it must be the oracle's bits.  (Crowded counting-sort cells take
    the general path of the bucket sort: such a scan costs up to 30 ms instead of 0.3 -- correct, not fast.)"""
    from tests.param_sweep import run_case, tie_cases
    for name, a, b in tie_cases():
        bits, d, r, ref, fits = run_case(gpu_ctx, np.ascontiguousarray(a), np.ascontiguousarray(b), 75, 24, dict(n=25, thresh=0.1, buff=0.1), 3, np.zeros(6, np.float32))
        assert all(bits.values()), (name, {k: v for k, v in bits.items() if not v})
        n = a.shape[0]
        from oracle import pyoracle as po
        assert np.array_equal(gpu_ctx.debug_fetch("src", n), po.scramble(po.c2s(np.ascontiguousarray(a))[:, 0])), name


def test_random_launch_knobs_do_not_show_in_the_bits():
    """25 random settings of the launch-shape / path-selection options (tests/param_sweep.KNOBS: LDS slot rows incl. more than fit, points and blocks of the point pass,
    keyframe tile size, batch parts, rank-sort capacity and cell limit, bit table in LDS or memory, LDS-rank or ballot multi-split, pairwise or tiled swap-loop flags,
    graph replay on / off; 60 more by hand, profiles/r05_fuzz_knobs.txt): single solve and 40-pair device batch, first call and repeat, carry the default settings' bits."""
    from tests.param_sweep import run_knob_draws
    failed = run_knob_draws(25, 1)
    assert not failed, failed[:3]


def test_multi_million_row_scan_and_five_thousand_pairs(gpu_ctx, sample_pc):
    """A 2.2 M-row pair (past the LDS bit table of the swap loop at 0.75 M rows and past the 2^20 rows a block of the point pass may count; 0.8 - 4.3 M by hand:
    scripts/fuzz_sizes.py -> profiles/r05_fuzz_sizes.txt) against the oracle, and ONE device batch of 5000 small pairs whose sampled members carry their single-solve bits."""
    from icet_amd import api
    from tests.param_sweep import run_case
    a0, b0 = sample_pc
    rows = 2_200_000
    rep = -(-rows // len(a0))
    a = np.concatenate([a0 * np.float32(1 + 1e-4 * k) for k in range(rep)])[:rows]      # every copy a slightly different range: no ties, the same directions
    b = np.concatenate([b0 * np.float32(1 + 1e-4 * k) for k in range(rep)])[:rows + 17]
    bits, d, r, ref, fits = run_case(gpu_ctx, np.ascontiguousarray(a), np.ascontiguousarray(b), 75, 24, dict(n=25, thresh=0.1, buff=0.1), 3, np.zeros(6, np.float32))
    assert all(bits.values()), {k: v for k, v in bits.items() if not v}
    assert d[:3].max() <= TOL_T and d[3:].max() <= TOL_R
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    n_pairs, k = 5000, 400
    starts = rng.integers(0, len(a0) - k, n_pairs)
    A = torch.from_numpy(np.ascontiguousarray(a0.T)).to(dev); B = torch.from_numpy(np.ascontiguousarray(b0.T)).to(dev)
    ld = A.shape[1]
    d1 = [(A.data_ptr() + 4 * int(s), k, ld) for s in starts]; d2 = [(B.data_ptr() + 4 * int(s), k + 3, ld) for s in starts]
    out = torch.zeros(n_pairs, 48, device=dev)
    gpu_ctx.solve_batch_device(d1, d2, api.Params(4, 8, 16, 10, 0.1, 0.1, 0), out.data_ptr()); torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert np.isfinite(o).all()
    single = api.Context()
    for j in list(range(0, n_pairs, 125)) + [n_pairs - 1]:
        s = int(starts[j])
        rs = single.solve(a0[s:s + k], b0[s:s + k + 3], 4, np.zeros(6), 8, 16, n=10)
        assert np.array_equal(o[j, :6].view(np.uint32), rs["X"].view(np.uint32)) and np.array_equal(o[j, 12:48].view(np.uint32), rs["cov"].reshape(36).view(np.uint32)), j
    single.close()


def _keep_batch(dev, frames, sample_pc, n_syn=40, n_real=8):
    from icet_amd import lidar_sim as ls
    pairs = [ls.make_batch_pair(k, device=dev)[:2] for k in list(range(n_syn - 2)) + [39, 232]]       # 39 / 232: the pairs whose X keeps moving by decimetres (fall back, rebuild)
    base = [tuple(torch.from_numpy(np.ascontiguousarray(x.T)).to(dev) for x in frames), tuple(torch.from_numpy(np.ascontiguousarray(x.T)).to(dev) for x in sample_pc)]
    for k in range(n_real):
        R = torch.as_tensor(ls.real_batch_rotation(k), device=dev)
        pairs.append(((R @ base[k % 2][0]).contiguous(), (R @ base[k % 2][1]).contiguous()))
    # ragged tails: scans whose length is no multiple of 4 / 256, a scan of 3 rows, an empty scan 2
    a, b = pairs[0]
    pairs += [(a[:, :70001].contiguous(), b[:, :69997].contiguous()), (a[:, :5000].contiguous(), b[:, :3].contiguous()), (a[:, :5000].contiguous(), b[:, :0].contiguous())]
    return pairs


@pytest.mark.gpu
def test_keep_list_is_bit_neutral(gpu_ctx, frames, sample_pc):
    """The keep list of the point pass (include/icet_hip.h option "keep"; src/icet.cpp:290-302: H^T W H sees only points in the angular bin of an active voxel):
    a throughput batch solved with the list off, on, with budgets so tight that nearly every pass falls back and rebuilds, with loose budgets, and with the marks
    made from iteration 0 -- 51 pairs (synthetic, real scans with thousands of zero rows, the two pairs whose X keeps moving, ragged and empty scans) must come
    out bit for bit the same, with and without a non-zero X0; and the statistics must show that lists were in fact walked."""
    import icet_amd
    from icet_amd import api
    dev = torch.device("cuda", 0)
    pairs = _keep_batch(dev, frames, sample_pc)
    N = len(pairs)
    d1 = [(p[0].data_ptr(), p[0].shape[1], p[0].shape[1]) for p in pairs]; d2 = [(p[1].data_ptr(), p[1].shape[1], p[1].shape[1]) for p in pairs]
    x0 = torch.zeros((N, 6), dtype=torch.float32, device=dev)
    x0[:, 0] = 0.02 * (torch.arange(N, device=dev) % 5).float(); x0[:, 5] = 0.002 * (torch.arange(N, device=dev) % 3).float()
    torch.cuda.synchronize()
    ctx = icet_amd.Context(0)

    def run(opts, runlen=7, x0p=None, flags=0):
        for k, v in dict(keep=1, keep_from=1, keep_budget_t=0.08, keep_budget_r=0.008).items():
            ctx.set_option(k, v)
        for k, v in opts.items():
            ctx.set_option(k, v)
        out = torch.full((N, 48), float("nan"), dtype=torch.float32, device=dev)
        ctx.solve_batch_device(d1, d2, api.Params(runlen, 24, 75, 25, 0.1, 0.1, flags), out.data_ptr(), x0p)
        ctx.sync()
        return out.cpu().numpy()

    for x0p in (None, x0.data_ptr()):
        off = run(dict(keep=0), x0p=x0p)
        on = run({}, x0p=x0p)
        st = ctx.keep_stats(N)
        n_groups = np.array([(p[1].shape[1] + 3) // 4 for p in pairs])
        frac = st[:N - 3, 1] / n_groups[:N - 3]
        print("keep list: groups kept min / median / max %.2f / %.2f / %.2f, list passes per pair %s, lists built %s" % (frac.min(), np.median(frac), frac.max(), np.bincount(st[:, 2]), np.bincount(st[:, 3])))
        assert np.array_equal(off.view(np.uint32), on.view(np.uint32)), np.nonzero((off.view(np.uint32) != on.view(np.uint32)).any(1))[0]
        assert (st[:N - 3, 2] >= 3).mean() > 0.7 and 0.2 < np.median(frac) < 0.85          # most pairs walked a list in most of the 5 passes behind the first marks, and the list is shorter than the scan
        for opts in (dict(keep_budget_t=1e-4, keep_budget_r=1e-5), dict(keep_budget_t=0.5, keep_budget_r=0.05), dict(keep_from=0), dict(keep_from=2, keep_budget_t=0.02)):
            got = run(opts, x0p=x0p)
            assert np.array_equal(off.view(np.uint32), got.view(np.uint32)), opts
    # other loop lengths (3: the shortest a list is used for; 12), per-iteration timing launches, the moving-object extension
    for rl, fl in ((3, 0), (4, 0), (12, 0), (7, api.FLAG_TIMING), (7, api.FLAG_REJECT_MOVING)):
        assert np.array_equal(run(dict(keep=0), rl, None, fl).view(np.uint32), run({}, rl, None, fl).view(np.uint32)), (rl, fl)
    # a finer grid with the one-block solve of 512 threads, and a coarse one
    for (P, T) in ((48, 150), (6, 11)):
        outs = []
        for keep in (0, 1):
            ctx.set_option("keep", keep)
            out = torch.zeros((N, 48), dtype=torch.float32, device=dev)
            ctx.solve_batch_device(d1, d2, api.Params(7, P, T, 25, 0.1, 0.1, 0), out.data_ptr()); ctx.sync()
            outs.append(out.cpu().numpy())
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), (P, T)
    ctx.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_multi_shard_fuzz_on_one_card(gpu_ctx, seed):
    """icet_multi_* with 1-6 shards on ONE card (an id may repeat: one context and one host thread per entry), ten random ragged batches per seed through the host-pointer
    entry and the device-resident entries, synchronous and asynchronous (scripts/fuzz_multi.py, 76 batches by hand in round 5): every pair carries the bits of its single solve."""
    from icet_amd import api
    from tests.param_sweep import pools as make_pools, draw_scan_pair
    rng = np.random.default_rng(seed)
    pools = make_pools()
    dev = torch.device("cuda", 0)
    for bno in range(10):
        shards = int(rng.integers(1, 7)); k = int(rng.choice([1, 3, 8, 17, 40]))
        T = int(rng.choice([40, 75, 128])); P = int(rng.choice([11, 24, 48])); runlen = int(rng.integers(1, 8))
        kw = dict(n=int(rng.choice([10, 25])), thresh=0.1, buff=float(rng.choice([0.0, 0.1])))
        pairs = [draw_scan_pair(rng, pools) for _ in range(k)]
        s1 = [np.ascontiguousarray(p[0]) for p in pairs]; s2 = [np.ascontiguousarray(p[1]) for p in pairs]
        x0 = (rng.normal(size=(k, 6)) * np.array([0.1, 0.1, 0.03, 0.003, 0.003, 0.01])).astype(np.float32); x0[rng.random(k) < 0.4] = 0
        m = api.MultiContext([0] * shards)
        host = m.solve_batch(s1, s2, runlen, x0, P, T, **kw)
        bufs1 = [torch.from_numpy(np.ascontiguousarray(s.T) if len(s) else np.zeros((3, 4), np.float32)).to(dev) for s in s1]
        bufs2 = [torch.from_numpy(np.ascontiguousarray(s.T) if len(s) else np.zeros((3, 4), np.float32)).to(dev) for s in s2]
        d1 = [(b.data_ptr(), len(s), b.shape[1]) for b, s in zip(bufs1, s1)]; d2 = [(b.data_ptr(), len(s), b.shape[1]) for b, s in zip(bufs2, s2)]
        prm = api.Params(runlen, P, T, kw["n"], kw["thresh"], kw["buff"], 0)
        out = torch.zeros(k, 48, device=dev); dx0 = torch.from_numpy(x0).to(dev); torch.cuda.synchronize()
        m.solve_batch_device(d1, d2, prm, out.data_ptr(), dx0.data_ptr()); o_sync = out.cpu().numpy()
        out.zero_(); torch.cuda.synchronize()
        m.solve_batch_device(d1, d2, prm, out.data_ptr(), dx0.data_ptr(), asynchronous=True); m.sync(); o_async = out.cpu().numpy()
        m.close()
        for j in range(k):
            r = gpu_ctx.solve(s1[j], s2[j], runlen, x0[j], P, T, **kw)
            ref = np.concatenate([r["X"], r["pred_stds"], r["cov"].reshape(36)]).view(np.uint32)
            h = np.concatenate([host["X"][j], host["pred_stds"][j], host["cov"][j].reshape(36)]).view(np.uint32)
            assert np.array_equal(h, ref) and np.array_equal(o_sync[j].view(np.uint32), ref) and np.array_equal(o_async[j].view(np.uint32), ref), (seed, bno, shards, k, j)


def test_four_host_threads_with_their_own_contexts(gpu_ctx):
    """Four host threads, each with its own context, solving different random draws of the parameter space at the same time, three times over (scripts/fuzz_threads.py):
    every result carries the bits of the same solve done alone -- nothing static in the library, as in the reference (include/icet.h:43 hints at concurrent objects)."""
    import threading
    from icet_amd import api
    from tests.param_sweep import draw_case, pools as make_pools
    rng = np.random.default_rng(1); pools = make_pools()
    ctxs = [api.Context() for _ in range(4)]
    for rnd in range(3):
        jobs = [[draw_case(rng, pools) for _ in range(5)] for _ in range(4)]
        refs = [[gpu_ctx.solve(a, b, rl, x0, P, T, **kw) for (a, b, T, P, kw, rl, x0) in js] for js in jobs]
        out = [[None] * 5 for _ in range(4)]; errs = []
        def work(t):
            try:
                for _ in range(3):
                    for i, (a, b, T, P, kw, rl, x0) in enumerate(jobs[t]):
                        out[t][i] = ctxs[t].solve(a, b, rl, x0, P, T, **kw)
            except Exception as e:
                errs.append(repr(e))
        th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        [t.start() for t in th]; [t.join() for t in th]
        assert not errs, errs
        for t in range(4):
            for i in range(5):
                assert all(np.array_equal(out[t][i][k].view(np.uint32), refs[t][i][k].view(np.uint32)) for k in ("X", "pred_stds", "cov")), (rnd, t, i)
    for c in ctxs:
        c.close()


def _pinv3_matrices():
    rng = np.random.default_rng(7)
    mats = []
    def rot():
        q, _ = np.linalg.qr(rng.normal(size=(3, 3))); return q
    for cond in np.logspace(0, 10, 41):
        for _ in range(24):
            Q = rot(); lam = np.array([1.0, cond ** -rng.uniform(0.3, 1.0), 1.0 / cond]) * 10.0 ** rng.uniform(-8, 2)
            A = (Q * lam) @ Q.T
            A32 = A.astype(np.float32)
            mats.append(A32)                                                               # symmetric to the last bit
            mats.append((A32 * (1 + rng.uniform(-6e-8, 6e-8, (3, 3)))).astype(np.float32))   # the two triangles differ by roundings, as the reference's float products leave them
            for mask in ((1, 1, 0), (0, 1, 1), (1, 0, 0)):                                   # axes masked out by L: exact zero rows / columns
                m = np.array(mask, np.float32); mats.append(A32 * np.outer(m, m))
    for _ in range(200):                                                                     # exactly rank-deficient
        u, v = rng.normal(size=3), rng.normal(size=3)
        mats.append(np.outer(u, u).astype(np.float32)); mats.append((np.outer(u, u) + np.outer(v, v)).astype(np.float32))
    mats += [np.zeros((3, 3), np.float32), np.eye(3, dtype=np.float32), np.full((3, 3), np.nan, np.float32), np.diag([1e-38, 1e-40, 1e-42]).astype(np.float32),
             np.diag([1e30, 1.0, 1e-30]).astype(np.float32), np.array([[1, 0, 0], [0, np.nan, 0], [0, 0, 1]], np.float32)]
    return np.stack(mats)


def test_pinv3_reference_bits(gpu_ctx):
    """The per-voxel weight of ICET_FLAG_REFERENCE_W: W = CompleteOrthogonalDecomposition<MatrixXf>(R).pseudoInverse() (src/icet.cpp:320-321) evaluated on the device for
    5300 3 x 3 matrices -- condition numbers 1 .. 1e10 (symmetric and with the two triangles a rounding apart), one or two axes masked out by L, exactly rank-deficient, zero,
    NaN, denormal and huge scales -- through the device function the solve kernel runs (icet_debug_pinv3): BIT-IDENTICAL to the oracle's restatement of Eigen's algorithm
    (oracle/smalllinalg.h cod_pinv), rank decisions included."""
    from oracle import pyoracle as po
    mats = _pinv3_matrices()
    dev = gpu_ctx.debug_pinv3(mats)
    ranks = []
    for k in range(mats.shape[0]):
        ref, rk = po.pinv(mats[k]); ranks.append(rk)
        nr, nd = np.isnan(ref), np.isnan(dev[k])
        assert np.array_equal(nr, nd) and np.array_equal(ref.view(np.uint32)[~nr], dev[k].view(np.uint32)[~nd]), (k, mats[k], ref, dev[k])
    print("pinv3: %d matrices bit-identical to the oracle; rank histogram %s" % (mats.shape[0], np.bincount(ranks, minlength=4).tolist()))
    assert min(np.bincount(ranks, minlength=4)[1:]) > 100


@pytest.mark.parametrize("name", ["wall_s10", "ground_s10_m", "ground_s05"])
def test_double_w_option_on_degenerate_scenes(gpu_ctx, name):
    """The per-voxel weight W (src/icet.cpp:320-321): by default as the reference takes it -- Eigen's float COD -- and with ICET_FLAG_DOUBLE_W (include/icet_hip.h) in double, the
    default of rounds 2-5.  On the scenes where the two stand apart the DEFAULT follows the unmodified oracle: H^T W H within 1.5 x the oracle's own 32-trial spread in
    every iteration, the pruned axes equal, pred_stds equal to 1e-3 where the oracle's are stable (ground_s10_m: the double path prunes nothing and reports pred_stds of
    1e-4 where the reference reports -0.9995).  On an ordinary pair the two paths agree to float tolerance."""
    from oracle import pyoracle as po
    from icet_amd import api
    a, b = _degenerate_scans(name)
    ref = po.solve(a, b, trace=True); t = ref["trace"]
    sp = _oracle_spread(a, b, ref)
    out = {}
    for flag in (0, api.FLAG_DOUBLE_W):
        r = gpu_ctx.solve(a, b, 7, np.zeros(6), 24, 75, aux=True, flags=flag)
        dH = float((np.abs(r["aux"]["htwh"] - t["HTWH"]).reshape(7, -1).max(1) / np.abs(t["HTWH"]).reshape(7, -1).max(1)).max())
        out[flag] = (dH, r["aux"]["cond_info"][:, 6].astype(int).tolist(), float(np.abs(r["pred_stds"] - ref["pred_stds"]).max()))
    print("%s: dH default (float COD) %.2e, ICET_FLAG_DOUBLE_W %.2e (oracle spread max %.2e p90 %.2e); pruned default %s double %s oracle %s; d pred_stds %.2e / %.2e (oracle spread %.2e)"
          % (name, out[0][0], out[32][0], sp["H"], sp["H_p90"], out[0][1], out[32][1], t["pruned"].tolist(), out[0][2], out[32][2], sp["ps"]))
    assert out[0][0] <= max(2e-3, HATCH * sp["H"]), (out, sp)
    assert out[0][1] == t["pruned"].tolist() or not sp["pruned_same"], (out, t["pruned"])
    if sp["ps"] < 1e-3:
        assert out[0][2] <= 1e-3, out


def test_double_w_option_on_an_ordinary_pair(gpu_ctx, frames):
    from icet_amd import api
    r0 = gpu_ctx.solve(frames[0], frames[1], 7, np.zeros(6), 24, 75)
    r1 = gpu_ctx.solve(frames[0], frames[1], 7, np.zeros(6), 24, 75, flags=api.FLAG_DOUBLE_W)
    _check_solution(r1, r0)
    assert not np.array_equal(r0["X"], r1["X"])                            # (the flag does change the arithmetic)
