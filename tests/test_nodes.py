"""The callers around the constructor: per-frame body of odometry_node and
map_maker_node (/root/reference/src/odometry.cpp:46-98, src/simpleMapMaker.cpp:18-59,86-172).

CPU tests pin the oracle's restatement (oracle/icet_nodes_oracle.cpp) against the already-tested single-pair oracle and
against NumPy restatements of the bookkeeping; GPU tests compare include/icet_nodes.h (through icet_amd.api.Node) with it:
  * range filter: kept rows and their order BIT-EXACT (it is a comparison of a correctly rounded float norm);
  * X / pred_stds: the single-pair tolerances of test_gpu_parity.py (3e-4 m, 1e-4 rad per frame);
  * pose: 1e-3 after a few frames (it chains the per-frame X);
  * map queue: bit-exact when the guard zeroes X (R = I, t = 0: pure data movement), 2e-3 m otherwise.
"""
import numpy as np
import pytest
import torch   # before libicet_hip.so is loaded: both must share one HIP runtime (torch ships its own libamdhip64)



def _sequence(n_frames=4, rings=32, steps=1024):
    from icet_amd import lidar_sim as ls
    motion = (0.25, 0.02, 0.005, 0.001, -0.001, 0.006)
    return [s.T.contiguous().numpy() for s in ls.make_sequence(n_frames, motion=motion, rings=rings, steps=steps)]


def _range_filter(scan, min_range):
    x, y, z = scan[:, 0], scan[:, 1], scan[:, 2]
    d = np.sqrt((x * x + y * y) + z * z)          # float32 throughout, no fused multiply-add in NumPy
    return scan[d > np.float32(min_range)]


def _euler_R(a):
    from oracle import pyoracle as po
    return po.euler_R(np.asarray(a, np.float32)).reshape(3, 3).astype(np.float64)


# ------------------------------------------------------------------------------------------------ CPU: the oracle itself
def test_node_oracle_first_frame_and_pair_selection():
    """First scan stored unfiltered, nothing solved; afterwards frame k is ICET(prev, filter(cur), X0)."""
    from oracle import pyoracle as po
    seq = _sequence(3)
    nd = po.Node(min_range=2.0, seed_x0=1)
    r0 = nd.push(seq[0])
    assert not r0["solved"] and r0["n_kept"] == len(seq[0]) and np.array_equal(r0["pose"], np.eye(4, dtype=np.float32))
    assert np.array_equal(r0["quat"], np.array([0, 0, 0, 1], np.float32))
    r1 = nd.push(seq[1])
    f1 = _range_filter(seq[1], 2.0)
    assert r1["solved"] and r1["n_kept"] == len(f1) < len(seq[1])
    ref1 = po.solve(seq[0], f1, x0=np.zeros(6))                      # scan 1 = UNFILTERED first frame (odometry.cpp:46-52)
    assert np.array_equal(r1["X"], ref1["X"]) and np.array_equal(r1["pred_stds"], ref1["pred_stds"])
    r2 = nd.push(seq[2])
    f2 = _range_filter(seq[2], 2.0)
    ref2 = po.solve(f1, f2, x0=ref1["X"])                            # X0 seeded with the previous X (odometry.cpp:82)
    assert np.array_equal(r2["X"], ref2["X"])
    # pose = product of the per-frame [R t; 0 1]
    H = np.eye(4)
    for X in (ref1["X"], ref2["X"]):
        Hi = np.eye(4); Hi[:3, :3] = _euler_R(X[3:]); Hi[:3, 3] = X[:3]
        H = H @ Hi
    assert np.allclose(r2["pose"], H, atol=2e-6)
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(H[:3, :3]).as_quat()
    assert min(np.abs(r2["quat"] - q).max(), np.abs(r2["quat"] + q).max()) < 2e-6
    nd.close()


def test_node_oracle_map_maker_settings_guard_and_reset():
    """simpleMapMaker: X0 reset to zero every frame; guard zeroes X (pose unchanged) when a component exceeds the threshold."""
    from oracle import pyoracle as po
    seq = _sequence(3)
    nd = po.Node(runlen=3, min_range=0.2, seed_x0=0, trans_thresh=0.3, rot_thresh=0.3)
    nd.push(seq[0])
    r1 = nd.push(seq[1]); r2 = nd.push(seq[2])
    f1, f2 = _range_filter(seq[1], 0.2), _range_filter(seq[2], 0.2)
    assert np.array_equal(r2["X"], po.solve(f1, f2, x0=np.zeros(6), runlen=3)["X"]) and not r2["diverged"]
    tight = po.Node(runlen=3, min_range=0.2, seed_x0=0, trans_thresh=1e-4, rot_thresh=1e-4)
    tight.push(seq[0])
    g = tight.push(seq[1])
    assert g["diverged"] and not g["X"].any() and np.array_equal(g["pose"], np.eye(4, dtype=np.float32))
    assert g["pred_stds"].any()                                       # pred_stds are published as they are
    nd.close(); tight.close()
    # a threshold of 0 switches ITS group off (include/icet_nodes.h): with only the rotation threshold set -- generously -- the
    # translation components must not be compared against 0 (every frame used to be flagged)
    rot_only = po.Node(runlen=3, min_range=0.2, seed_x0=0, trans_thresh=0.0, rot_thresh=0.3)
    rot_only.push(seq[0])
    g = rot_only.push(seq[1])
    assert not g["diverged"] and np.array_equal(g["X"], r1["X"])
    trans_only = po.Node(runlen=3, min_range=0.2, seed_x0=0, trans_thresh=1e-4, rot_thresh=0.0)
    trans_only.push(seq[0])
    assert trans_only.push(seq[1])["diverged"]
    rot_only.close(); trans_only.close()


def test_node_oracle_map_queue_matches_numpy_ring():
    """EigenQueue: m rows per frame enter at `pos`, then ALL rows become (row - t) R^-1; getQueue returns oldest first.
    With a capacity of 2.5 frames the ring wraps on the third insertion."""
    from oracle import pyoracle as po
    seq = _sequence(5, rings=16, steps=512)
    m, cap = 1000, 2500
    kw = dict(runlen=2, min_range=0.2, seed_x0=0, trans_thresh=0.3, rot_thresh=0.3, map_capacity=cap, map_downsample=m)
    a, b = po.Node(**kw), po.Node(**kw)
    a.push(seq[0]); b.push(seq[0])
    assert a.map().shape == (0, 3)
    for k in range(1, 5):
        r = a.push(seq[k]); b.push(seq[k])
        f = _range_filter(seq[k], 0.2)
        mp = a.map()
        assert r["map_rows"] == min(k * m, cap) == len(mp)
        # the newest m rows are a subset of this frame's filtered scan, expressed in the new sensor frame
        Rinv = np.linalg.inv(_euler_R(r["X"][3:])); t = r["X"][:3].astype(np.float64)
        back = mp[-m:].astype(np.float64) @ np.linalg.inv(Rinv) + t
        # every older row moved rigidly: distances between consecutive rows are preserved
        if k >= 2:
            old_now = mp[-2 * m:-m].astype(np.float64)
            assert np.allclose(np.linalg.norm(np.diff(old_now, axis=0), axis=1), np.linalg.norm(np.diff(prev_newest, axis=0), axis=1), atol=2e-4)
        prev_newest = mp[-m:].astype(np.float64)
        # exact membership: undo the transform in float64 and match to the nearest scan row
        from scipy.spatial import cKDTree
        dist, _ = cKDTree(f.astype(np.float64)).query(back)
        assert dist.max() < 1e-4
    assert np.array_equal(a.map(), b.map())                           # default-seeded mt19937: two nodes agree
    a.close(); b.close()


# ------------------------------------------------------------------------------------------------ GPU: parity
@pytest.fixture(scope="module")
def seq64():
    return _sequence(4, rings=64, steps=2048)


@pytest.mark.gpu
def test_gpu_odometry_node_matches_oracle(gpu_ctx, seq64):
    from oracle import pyoracle as po
    from icet_amd import api
    g = api.Node(gpu_ctx, **api.ODOMETRY_NODE)
    o = po.Node(**api.ODOMETRY_NODE)
    for k, s in enumerate(seq64):
        rg, ro = g.push(s), o.push(s)
        assert rg["solved"] == ro["solved"] and rg["n_kept"] == ro["n_kept"] and rg["diverged"] == ro["diverged"]
        expect = s if k == 0 else _range_filter(s, 2.0)
        assert np.array_equal(g.prev_scan(), expect)                  # filter: same rows, same order, bit for bit
        if k:
            assert np.abs(rg["X"][:3] - ro["X"][:3]).max() <= 3e-4 * k and np.abs(rg["X"][3:] - ro["X"][3:]).max() <= 1e-4 * k, (k, rg["X"], ro["X"])
            assert np.allclose(rg["pred_stds"], ro["pred_stds"], rtol=2e-2, atol=1e-7)
            assert np.abs(rg["pose"] - ro["pose"]).max() <= 1e-3
            assert min(np.abs(rg["quat"] - ro["quat"]).max(), np.abs(rg["quat"] + ro["quat"]).max()) <= 1e-3
    g.close(); o.close()                                               # (phase timings: test_gpu_one_launch_frame_equals_phased_frame)
    # the keyframe of every scan is built one frame ahead on a second stream; with ICET_NODE_NO_PIPELINE (flags = 8) it
    # is built inside the frame's own solve, as the reference does: the results must be the same BITS
    kw = dict(api.ODOMETRY_NODE); kw["flags"] = kw.get("flags", 0) | 8
    a_, b_ = api.Node(gpu_ctx, **api.ODOMETRY_NODE), api.Node(gpu_ctx, **kw)
    for s in seq64:
        ra, rb = a_.push(s), b_.push(s)
        assert np.array_equal(ra["X"], rb["X"]) and np.array_equal(ra["pred_stds"], rb["pred_stds"]) and np.array_equal(ra["pose"], rb["pose"])
    a_.close(); b_.close()


@pytest.mark.gpu
def test_gpu_odometry_burst_equals_frame_by_frame(gpu_ctx, seq64):
    """icet_node_push_many_device: a burst of frames through one call must give the bits of frame-by-frame pushes -- X, pred_stds, pose, quaternion, kept rows, the
    stored previous scan -- for the burst started cold (first cloud inside the burst), for a burst after single pushes, for bursts in a row and for single pushes
    between bursts; the map maker goes through the same entry.  (Rounds 4-5 chained the frames on the device; round 6 pushes them one by one inside the library.)"""
    from icet_amd import api
    dev = torch.device("cuda", 0)
    seq = seq64 + _sequence(12, rings=64, steps=2048)[4:]             # twelve frames of the same drive
    bufs = [torch.from_numpy(np.ascontiguousarray(s.T)).to(dev) for s in seq]
    fr = [(b.data_ptr(), b.shape[1], b.shape[1]) for b in bufs]
    ref_node = api.Node(gpu_ctx, **api.ODOMETRY_NODE)
    ref = [ref_node.push_device(*f) for f in fr]
    ref_prev = ref_node.prev_scan()
    ref_node.close()
    def same(a, b):
        return all(np.array_equal(a[k], b[k]) for k in ("X", "pred_stds", "pose", "quat")) and a["solved"] == b["solved"] and a["n_kept"] == b["n_kept"] and a["diverged"] == b["diverged"]
    for split in ((len(fr),), (1, len(fr) - 1), (2, 1, len(fr) - 3), (1, 3, 1, 3, 1, len(fr) - 9)):
        nd = api.Node(gpu_ctx, **api.ODOMETRY_NODE)
        got, k = [], 0
        for j, m in enumerate(split):
            if m == 1 and (j % 2 == 1 or len(split) > 4):
                got.append(nd.push_device(*fr[k]))                      # a single push between bursts
            else:
                got += nd.push_many_device(fr[k:k + m])
            k += m
        assert len(got) == len(ref) and all(same(a, b) for a, b in zip(got, ref)), split
        assert np.array_equal(nd.prev_scan(), ref_prev)
        nd.close()
    mm, mref = api.Node(gpu_ctx, **api.MAP_MAKER_NODE), api.Node(gpu_ctx, **api.MAP_MAKER_NODE)
    a = mm.push_many_device(fr[:3]); b = [mref.push_device(*f) for f in fr[:3]]
    assert all(same(x, y) for x, y in zip(a, b)) and np.array_equal(mm.map(), mref.map())
    mm.close(); mref.close()


@pytest.mark.gpu
def test_gpu_one_launch_frame_equals_phased_frame(gpu_ctx, seq64):
    """Round 6: a pipelined odometry frame is ONE graph launch (range filter captured in front of the loop, the keyframe build behind a filter of its own on the other
    stream).  ICET_NODE_TIME_PHASES keeps the older arrangement (filter, then loop, events in between).  Same bits for everything a caller can read -- over frames whose
    row counts differ (the captured launches are sized by the buffers' capacity), a frame that grows the buffers, a burst in between and the map-less settings of the
    random-settings test."""
    from icet_amd import api
    dev = torch.device("cuda", 0)
    seq = seq64 + _sequence(10, rings=64, steps=2048)[4:]
    seq = [s if k % 3 else s[: len(s) - 517 * k] for k, s in enumerate(seq)]                # ragged row counts
    seq.insert(6, np.concatenate([seq[5], seq[4][:40000]]))                                 # one frame larger than every buffer so far
    bufs = [torch.from_numpy(np.ascontiguousarray(s.T)).to(dev) for s in seq]
    fr = [(b.data_ptr(), b.shape[1], b.shape[1]) for b in bufs]
    def same(a, b):
        return all(np.array_equal(a[k], b[k]) for k in ("X", "pred_stds", "pose", "quat")) and a["solved"] == b["solved"] and a["n_kept"] == b["n_kept"] and a["diverged"] == b["diverged"]
    # (the keyframe side of a one-launch frame enqueued by the calling thread instead of the helper: ICET_NODE_SERIAL_ENQUEUE)
    ser, ref = api.Node(gpu_ctx, **dict(api.ODOMETRY_NODE, flags=api.NODE_SERIAL_ENQUEUE)), api.Node(gpu_ctx, **api.ODOMETRY_NODE)
    for k, f in enumerate(fr[:6]):
        assert same(ser.push_device(*f), ref.push_device(*f)), k
    ser.close(); ref.close()
    for extra in (dict(), dict(seed_x0=0), dict(min_range=6.0, runlen=3)):
        kw = dict(api.ODOMETRY_NODE); kw.update(extra)
        one, ph = api.Node(gpu_ctx, **kw), api.Node(gpu_ctx, **dict(kw, flags=api.NODE_TIME_PHASES))
        for k, f in enumerate(fr):
            if k == 8:                                                                      # a burst of two in the middle of the sequence
                ra = one.push_many_device(fr[8:10]); rb = ph.push_many_device(fr[8:10])
                assert all(same(x, y) for x, y in zip(ra, rb))
                continue
            if k == 9: continue
            ra, rb = one.push_device(*f), ph.push_device(*f)
            assert same(ra, rb), (extra, k)
            assert np.array_equal(one.prev_scan(), ph.prev_scan())
        t2 = ph.last_timing()
        assert t2["filter_ms"] > 0.003 and t2["solve_ms"] > 0
        with pytest.raises(Exception):                                                      # the one-launch frame records no timing events
            one.last_timing()
        one.close(); ph.close()
    # the map maker's frame is one launch too (its host side watches the filter's row count arrive in pinned memory and shuffles beside the loop): results, ring
    # contents and the stored scan, a ring that wraps
    kw = dict(api.MAP_MAKER_NODE); kw.update(map_capacity=7000, runlen=5)
    one, ph = api.Node(gpu_ctx, **kw), api.Node(gpu_ctx, **dict(kw, flags=api.NODE_TIME_PHASES))
    for k, f in enumerate(fr[:8]):
        ra, rb = one.push_device(*f), ph.push_device(*f)
        assert same(ra, rb) and ra["map_rows"] == rb["map_rows"], k
        assert np.array_equal(one.map(), ph.map()) and np.array_equal(one.prev_scan(), ph.prev_scan())
    assert ph.last_timing()["map_ms"] > 0
    one.close(); ph.close()


@pytest.mark.gpu
def test_gpu_one_launch_frame_soak(gpu_ctx):
    """A thousand frames through the one-launch path (helper thread, pinned done word, folded filter, ragged row counts) against the phased node, bit for bit, and two
    nodes driven from two host threads at once against one node alone (scripts/node_soak.py runs thousands more; profiles/r06_node_soak.txt)."""
    import threading
    import icet_amd
    from icet_amd import api, lidar_sim as ls
    dev = torch.device("cuda", 0)
    frames = ls.make_sequence(24, motion=(0.25, 0.02, 0.005, 0.001, -0.001, 0.006), rings=32, steps=1024, device=dev)
    frames = [f if k % 5 else f[:, : f.shape[1] - 97 * (k % 7)].contiguous() for k, f in enumerate(frames)]
    a, b = api.Node(gpu_ctx, **api.ODOMETRY_NODE), api.Node(gpu_ctx, **dict(api.ODOMETRY_NODE, flags=api.NODE_TIME_PHASES))
    ref = []
    for k in range(1000):
        f = frames[(k * 7) % len(frames)]
        ra, rb = a.push_device(f.data_ptr(), f.shape[1], f.shape[1]), b.push_device(f.data_ptr(), f.shape[1], f.shape[1])
        assert np.array_equal(ra["X"], rb["X"]) and np.array_equal(ra["pred_stds"], rb["pred_stds"]) and ra["n_kept"] == rb["n_kept"], k
        ref.append(ra["X"].copy())
    a.close(); b.close()
    bad = [0, 0]
    def drive(i):
        c = icet_amd.Context(0); nd = api.Node(c, **api.ODOMETRY_NODE)
        for k in range(400):
            f = frames[(k * 7) % len(frames)]
            if not np.array_equal(nd.push_device(f.data_ptr(), f.shape[1], f.shape[1])["X"], ref[k]): bad[i] += 1
        nd.close(); c.close()
    ts = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert bad == [0, 0]


@pytest.mark.gpu
def test_gpu_map_maker_node_matches_oracle(gpu_ctx, seq64):
    from oracle import pyoracle as po
    from icet_amd import api
    kw = dict(api.MAP_MAKER_NODE); kw.update(map_capacity=5000, map_downsample=2000, runlen=7)     # wraps on the third frame
    g, o = api.Node(gpu_ctx, **kw), po.Node(**kw)
    for k, s in enumerate(seq64):
        rg, ro = g.push(s), o.push(s)
        assert rg["n_kept"] == ro["n_kept"] and rg["map_rows"] == ro["map_rows"] and rg["diverged"] == ro["diverged"]
        if k:
            assert np.abs(rg["X"] - ro["X"]).max() <= 3e-4
        mg, mo = g.map(), o.map()
        assert mg.shape == mo.shape and (len(mg) == 0 or np.abs(mg - mo).max() <= 2e-3)
    g.close(); o.close()
    # guard tripped on every frame: X = 0, R = I, the queue only moves data -> bit-exact ring behaviour incl. wrap-around
    kw.update(trans_thresh=1e-6, rot_thresh=1e-6, map_capacity=4500)
    g, o = api.Node(gpu_ctx, **kw), po.Node(**kw)
    for s in seq64:
        rg, ro = g.push(s), o.push(s)
        assert rg["diverged"] == ro["diverged"] and rg["map_rows"] == ro["map_rows"]
        assert np.array_equal(g.map(), o.map())
        assert np.array_equal(rg["pose"], ro["pose"])
    g.close(); o.close()
    # one threshold set, the other 0 (= off): only the set group is tested
    kw.update(trans_thresh=0.0, rot_thresh=0.3, map_capacity=5000)
    g, o = api.Node(gpu_ctx, **kw), po.Node(**kw)
    for s in seq64[:3]:
        rg, ro = g.push(s), o.push(s)
        assert rg["diverged"] == ro["diverged"] == 0
    g.close(); o.close()


@pytest.mark.gpu
def test_gpu_range_filter_full_size_and_device_push(gpu_ctx):
    """config-5 size (524288 rows), rows on both sides of the threshold and exactly on it, pushed from HBM."""
    from icet_amd import api
    rs = np.random.RandomState(7)
    n = 524288
    dirs = rs.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    r = rs.uniform(0.0, 4.0, n); r[::1000] = 2.0; r[5::1000] = 0.0
    scan = (dirs * r[:, None]).astype(np.float32)
    g = api.Node(gpu_ctx, runlen=1, min_range=2.0)
    dev = torch.device("cuda", 0)
    first = torch.from_numpy(np.ascontiguousarray(scan.T)).to(dev)
    g.push_device(first.data_ptr(), n, n)
    assert np.array_equal(g.prev_scan(), scan)                        # first frame: unfiltered
    ld = n + 64
    buf = torch.zeros((3, ld), dtype=torch.float32, device=dev); buf[:, :n] = first
    torch.cuda.synchronize()
    res = g.push_device(buf.data_ptr(), n, ld)
    expect = _range_filter(scan, 2.0)
    assert res["n_kept"] == len(expect) and np.array_equal(g.prev_scan(), expect)
    # ragged / degenerate inputs
    for m in (0, 1, 63, 2049):
        res = g.push(scan[:m])
        assert res["n_kept"] == len(_range_filter(scan[:m], 2.0))
    g.close()


@pytest.mark.gpu
def test_gpu_cpp_node_demo(tmp_path, gpu_ctx, seq64):
    """include/icet_nodes.hpp compiled with plain g++: frames read from .npy files by icet_load_scan, fed to OdometryNode /
    MapMakerNode::pointCloudCallback; same numbers as the Python mirror of the same C ABI, bit for bit."""
    import os, subprocess
    from icet_amd import api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = []
    for k, s in enumerate(seq64[:3]):
        p = str(tmp_path / ("frame_%03d.npy" % k)); np.save(p, s.astype(np.float64)); files.append(p)      # float64 like the reference's sample data
    exe = str(tmp_path / "node_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "node_demo.cpp"),
                           "-L", os.path.join(root, "icet_amd", "lib"), "-licet_hip", "-Wl,-rpath," + os.path.join(root, "icet_amd", "lib"), "-o", exe])
    for mode, kw in (("odometry", api.ODOMETRY_NODE), ("mapmaker", api.MAP_MAKER_NODE)):
        out = subprocess.run([exe, mode] + files, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        rows = [l.split() for l in out.stdout.strip().splitlines() if l.startswith("frame")]
        g = api.Node(gpu_ctx, **kw)
        for k, s in enumerate(seq64[:3]):
            r = g.push(s)
            assert int(rows[k][3]) == int(r["solved"]) and int(rows[k][5]) == r["n_kept"]
            assert np.array_equal(np.array(rows[k][7:13], np.float32), r["X"])
            assert np.array_equal(np.array(rows[k][14:17], np.float32), r["pose"][:3, 3])
            assert int(rows[k][-1]) == r["map_rows"]
        g.close()
        if mode == "mapmaker":
            assert out.stdout.strip().splitlines()[-1] == "map_rows 4000"


def test_node_oracle_scan_registration_mode():
    """scanMatcher.cpp: no range filter, X0 = 0 every frame, scan 2 re-expressed in scan 1's frame as (M R^-1) - t (rotate, THEN
    subtract) and the snail trail transformed the same way with a fresh origin row appended."""
    from oracle import pyoracle as po
    seq = _sequence(3)
    seq[1] = seq[1].copy(); seq[1][5] = [np.nan, 1.0, 2.0]; seq[1][6] = 0.0          # NaN and zero rows are NOT filtered here (:44)
    nd = po.Node(min_range=2.0, seed_x0=0, flags=7)
    nd.push(seq[0])
    assert nd.snail_trail().shape == (1, 3) and nd.aligned().shape == (0, 3)
    r1 = nd.push(seq[1])
    assert r1["n_kept"] == len(seq[1])
    ref = po.solve(seq[0], seq[1], x0=np.zeros(6))
    assert np.array_equal(r1["X"], ref["X"])
    Rinv = np.linalg.inv(_euler_R(r1["X"][3:])); t = r1["X"][:3].astype(np.float64)
    al = nd.aligned()
    ok = np.isfinite(seq[1]).all(1)
    assert al.shape == seq[1].shape and np.allclose(al[ok], seq[1][ok].astype(np.float64) @ Rinv - t, atol=2e-5)
    assert np.isnan(al[5]).any()
    st = nd.snail_trail()
    assert st.shape == (2, 3) and not st[1].any() and np.allclose(st[0], -t, atol=1e-6)      # the old origin seen from the new frame
    r2 = nd.push(seq[2])
    assert np.array_equal(r2["X"], po.solve(seq[1], seq[2], x0=np.zeros(6))["X"])             # X0 reset, prev = unfiltered scan 1
    assert nd.snail_trail().shape == (3, 3)
    nd.close()


@pytest.mark.gpu
def test_gpu_scan_registration_node_matches_oracle(gpu_ctx, seq64):
    from oracle import pyoracle as po
    from icet_amd import api
    seq = [s.copy() for s in seq64]
    seq[2][100] = [np.nan, 0.5, 0.5]; seq[2][101] = 0.0
    g, o = api.Node(gpu_ctx, **api.SCAN_REGISTRATION_NODE), po.Node(**api.SCAN_REGISTRATION_NODE)
    for k, s in enumerate(seq):
        rg, ro = g.push(s), o.push(s)
        assert rg["n_kept"] == ro["n_kept"] == len(s)
        assert np.array_equal(g.prev_scan().view(np.uint32), s.view(np.uint32))               # nothing filtered, NaN row included
        if k:
            assert np.abs(rg["X"] - ro["X"]).max() <= 3e-4
            ag, ao = g.aligned(), o.aligned()
            fin = np.isfinite(ao).all(1)
            assert ag.shape == ao.shape and np.abs(ag[fin] - ao[fin]).max() <= 5e-3 and np.isnan(ag[~fin]).any(1).all()
            sg, so = g.snail_trail(), o.snail_trail()
            assert sg.shape == so.shape == (k + 1, 3) and np.abs(sg - so).max() <= 2e-3
    g.close(); o.close()
    # with X forced to zero (guard) the two clouds are pure data movement: bit-exact
    kw = dict(api.SCAN_REGISTRATION_NODE); kw.update(trans_thresh=1e-6, rot_thresh=1e-6)
    g, o = api.Node(gpu_ctx, **kw), po.Node(**kw)
    for s in seq[:3]:
        g.push(s); o.push(s)
        assert np.array_equal(g.aligned().view(np.uint32), o.aligned().view(np.uint32))
        assert np.array_equal(g.snail_trail(), o.snail_trail())
    g.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gpu_nodes_random_settings_and_hostile_frames(gpu_ctx, seed):
    """(Three seeds in the suite since round 6 rewrote the frame path.)  15 seeded draws (tests/param_sweep.draw_node_case; 60 more by hand, scripts/fuzz_nodes.py -> profiles/r05_fuzz_nodes.txt, 330 in profiles/r06_fuzz_final_pass.txt): odometry / map-maker / registration
    settings with random runlen, min_range, minimum points, guard thresholds and queue sizes, over short drives in which frames are replaced by hostile ones -- an empty
    cloud, a cloud entirely inside min_range, a cloud of a few rows, zero and NaN rows, the same cloud twice.  Frame by frame the device node takes the oracle node's
    decisions (solved, diverged, kept rows, map rows; finite X where the oracle's is); and the same frames as ONE device burst give the bits of the frame-by-frame pushes."""
    from tests.param_sweep import draw_node_case, run_node_case
    rng = np.random.default_rng(seed)
    for c in range(15):
        kw, frames = draw_node_case(rng)
        bad = run_node_case(gpu_ctx, kw, frames)
        assert not bad, (c, kw, [len(f) for f in frames], bad[:4])
