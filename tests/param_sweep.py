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




# ---- random batches: every pair must carry the bits of its own single solve ------------------------------------------------------------------
def draw_scan_pair(rng, pools):
    a0, b0 = pools[rng.integers(len(pools))]
    kind = rng.random()
    if kind < 0.05: return np.zeros((0, 3), np.float32), b0[:2000]
    if kind < 0.10: return a0[:3000], np.zeros((0, 3), np.float32)
    if kind < 0.15: return a0[:int(rng.integers(1, 40))], b0[:int(rng.integers(1, 40))]
    m = int(rng.choice([2500, 9000, 30000, a0.shape[0]]))
    if m >= a0.shape[0]: return a0, b0
    if rng.random() < 0.5:
        s = int(rng.integers(0, a0.shape[0] - m)); return a0[s:s + m], b0[s:s + m + int(rng.integers(0, 300))]
    k = a0.shape[0] // m
    return a0[::k], b0[int(rng.integers(0, k))::k]



def run_batch(ctx, single, rng, pools):
    """One random batch -- ragged pairs (empty, tiny, strided, whole scans), a size on either side of the library's small-batch / throughput switch (32 pairs),
    random parameters -- through icet_solve_batch (host pointers) and icet_solve_batch_device (descriptors whose leading dimension exceeds the row count, NaN in
    the padding).  Returns (description, [(pair, n1, n2, host path equal, device path equal) for every pair that differs from its single solve])."""
    import torch
    from icet_amd import api
    k = int(rng.choice([1, 2, 7, 31, 32, 33, 48]))
    T = int(rng.choice([12, 40, 75, 128])); P = int(rng.choice([3, 11, 24, 48]))
    kw = dict(n=int(rng.choice([3, 25, 60])), thresh=float(rng.choice([0.05, 0.1, 0.5])), buff=float(rng.choice([0.0, 0.1, 1.0])))
    runlen = int(rng.integers(1, 9))
    pairs = [draw_scan_pair(rng, pools) for _ in range(k)]
    s1 = [np.ascontiguousarray(p[0]) for p in pairs]; s2 = [np.ascontiguousarray(p[1]) for p in pairs]
    x0 = (rng.normal(size=(k, 6)) * np.array([0.1, 0.1, 0.03, 0.003, 0.003, 0.01])).astype(np.float32)
    x0[rng.random(k) < 0.4] = 0
    host = ctx.solve_batch(s1, s2, runlen, x0, P, T, **kw)
    # the same batch resident in HBM, every scan in a buffer with a leading dimension larger than its row count (3 x ld floats, rows 0 .. n-1 used)
    bufs, d1, d2 = [], [], []
    for lst, dl in ((s1, d1), (s2, d2)):
        for s in lst:
            n = s.shape[0]; ld = (n + int(rng.integers(0, 70)) + 3) & ~3
            t = torch.full((3, max(ld, 4)), float("nan"), device="cuda")
            if n: t[:, :n] = torch.from_numpy(np.ascontiguousarray(s.T)).cuda()
            bufs.append(t); dl.append((t.data_ptr(), n, t.shape[1]))
    out = torch.zeros(k, 48, device="cuda"); dx0 = torch.from_numpy(x0).cuda()
    prm = api.Params(runlen, P, T, kw["n"], kw["thresh"], kw["buff"], 0)
    ctx.solve_batch_device(d1, d2, prm, out.data_ptr(), dx0.data_ptr()); torch.cuda.synchronize()
    o = out.cpu().numpy()
    diffs = []
    for j in range(k):
        r = single.solve(s1[j], s2[j], runlen, x0[j], P, T, **kw)
        okh = np.array_equal(host["X"][j].view(np.uint32), r["X"].view(np.uint32)) and np.array_equal(host["pred_stds"][j].view(np.uint32), r["pred_stds"].view(np.uint32)) \
            and np.array_equal(host["cov"][j].view(np.uint32), r["cov"].reshape(6, 6).view(np.uint32))
        okd = np.array_equal(o[j, :6].view(np.uint32), r["X"].view(np.uint32)) and np.array_equal(o[j, 6:12].view(np.uint32), r["pred_stds"].view(np.uint32)) \
            and np.array_equal(o[j, 12:48].view(np.uint32), r["cov"].reshape(36).view(np.uint32))
        if not (okh and okd): diffs.append((j, s1[j].shape[0], s2[j].shape[0], okh, okd))
    desc = "pairs=%2d T=%3d P=%2d n=%2d thresh=%.2f buff=%.1f runlen=%d rows1=%d..%d" % (k, T, P, kw["n"], kw["thresh"], kw["buff"], runlen, min(s.shape[0] for s in s1), max(s.shape[0] for s in s1))
    return desc, diffs


# ---- adversarial scans: ties, lattice points on voxel edges, non-finite rows, extreme magnitudes --------------------------------------------------
def spoil(rng, scan):
    """A copy of `scan` with one or more of: ranges quantised to 1 cm (thousands of equal sort keys: the reference's order among them is by row), points snapped
    to a 0.25 m lattice (azimuths of exactly pi / 4, pi / 2 ...: ON voxel edges of the usual grids; z = 0: phi = pi / 2 exactly), duplicated rows, NaN / +-inf
    entries, rows scaled by 1e-20 / 1e+18 (squares under- / overflow), exact zeros of either sign."""
    o = scan.astype(np.float32).copy()
    n = o.shape[0]
    what = []
    if rng.random() < 0.5:
        r = np.linalg.norm(o.astype(np.float64), axis=1); ok = r > 0
        q = np.round(r[ok] * 100) / 100
        o[ok] = (o[ok].astype(np.float64) * (q / r[ok])[:, None]).astype(np.float32); what.append("quantised")
    if rng.random() < 0.5:
        idx = rng.choice(n, n // 3, replace=False); o[idx] = np.round(o[idx] * 4) / 4; what.append("lattice")
    if rng.random() < 0.4:
        idx = rng.choice(n, n // 10, replace=False); o[idx] = o[rng.choice(n, n // 10)]; what.append("duplicates")
    if rng.random() < 0.4:
        idx = rng.choice(n, 30, replace=False)
        o[idx, rng.integers(0, 3, 30)] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), 30); what.append("nonfinite")
    if rng.random() < 0.4:
        idx = rng.choice(n, 40, replace=False); o[idx[:20]] *= np.float32(1e-20); o[idx[20:]] *= np.float32(1e18); what.append("magnitudes")
    if rng.random() < 0.4:
        idx = rng.choice(n, n // 20, replace=False); o[idx] = rng.choice(np.array([0.0, -0.0], np.float32), (n // 20, 3)); what.append("zeros")
    return o, "+".join(what) or "plain"


def draw_adversarial(rng, pools):
    a, b, T, P, kw, runlen, x0 = draw_case(rng, pools)
    if a.shape[0] > 45000: a, b = a[::3], b[::3]
    T = int(rng.choice([8, 16, 24, 64, 75, 128])); P = int(rng.choice([2, 4, 8, 24, 32]))      # multiples of 8 put the lattice azimuths ON edges; P even puts z = 0 on one
    a, wa = spoil(rng, a); b, wb = spoil(rng, b)
    x0 = np.zeros(6, np.float32)                      # the identity transform: the first iteration's counts must be the oracle's exactly
    return a, b, T, P, kw, min(runlen, 4), x0, wa + " | " + wb


# ---- the sequential callers: random node settings, hostile frames ----------------------------------------------------------------------------------
def draw_node_case(rng):
    """(node keyword arguments, list of frames): a short synthetic drive with some frames replaced by hostile ones -- empty, entirely inside min_range, a few rows,
    NaN / zero rows mixed in, the same frame twice."""
    from icet_amd import api, lidar_sim as ls
    base = dict([api.ODOMETRY_NODE, api.MAP_MAKER_NODE, api.SCAN_REGISTRATION_NODE][int(rng.integers(3))])
    base.update(runlen=int(rng.integers(1, 8)), min_range=float(rng.choice([0.0, 0.2, 2.0, 6.0])), n=int(rng.choice([10, 25, 50])))
    if base.get("map_capacity", 0):
        base.update(map_capacity=int(rng.choice([3000, 20000, 600000])), map_downsample=int(rng.choice([0, 500, 4000])))
        base["map_downsample"] = min(base["map_downsample"], base["map_capacity"])      # (icet_node_create refuses a per-frame sample larger than the queue)
    if rng.random() < 0.4: base.update(trans_thresh=float(rng.choice([1e-6, 0.05, 1.0])), rot_thresh=float(rng.choice([0.0, 1e-6, 0.5])))
    nf = int(rng.integers(3, 8))
    motion = (float(rng.normal(0.2, 0.1)), float(rng.normal(0, 0.03)), 0.005, 0.001, -0.001, float(rng.normal(0, 0.01)))
    frames = [s.T.contiguous().numpy() for s in ls.make_sequence(nf, motion=motion, rings=int(rng.choice([16, 32])), steps=int(rng.choice([512, 1024])))]
    for k in range(nf):
        u = rng.random()
        f = frames[k]
        if u < 0.08: frames[k] = np.zeros((0, 3), np.float32)
        elif u < 0.16: frames[k] = (f / np.maximum(np.linalg.norm(f, axis=1, keepdims=True), 1e-9) * 0.1).astype(np.float32)      # everything inside min_range (unless 0)
        elif u < 0.24: frames[k] = f[:int(rng.integers(1, 30))].copy()
        elif u < 0.36:
            g = f.copy(); idx = rng.choice(len(g), len(g) // 8, replace=False); g[idx] = 0.0
            g[rng.choice(len(g), 5, replace=False), int(rng.integers(3))] = np.nan; frames[k] = g
        elif u < 0.42 and k: frames[k] = frames[k - 1].copy()
    return base, frames


def run_node_case(ctx, kw, frames):
    """GPU node against the oracle's node, frame by frame; then the same frames as ONE device burst against the frame-by-frame results (bits).  Returns a list of
    complaints (empty = fine)."""
    import torch
    from icet_amd import api
    from oracle import pyoracle as po
    bad = []
    g, o = api.Node(ctx, **kw), po.Node(**kw)
    got = []
    for k, s in enumerate(frames):
        rg, ro = g.push(s), o.push(s)
        got.append(rg)
        if rg["diverged"] != ro["diverged"] and rg["solved"] == ro["solved"] and rg["n_kept"] == ro["n_kept"]:
            # a guard threshold inside the parity tolerance of the solution (a duplicated frame solves to ~1e-7 against rot_thresh 1e-6): either answer is right, and the
            # two nodes' states part ways from here on -- the rest of the drive says nothing
            raw = np.abs((ro if not ro["diverged"] else rg)["X"]); tt, rt = kw.get("trans_thresh", 0.0), kw.get("rot_thresh", 0.0)
            if (tt > 0 and (np.abs(raw[:3] - tt) <= 2e-4).any()) or (rt > 0 and (np.abs(raw[3:] - rt) <= 2e-5).any()): break
        for key in ("solved", "diverged", "n_kept", "map_rows"):
            if rg[key] != ro[key]: bad.append("frame %d: %s device %s oracle %s" % (k, key, rg[key], ro[key]))
        if not np.isfinite(rg["X"]).all() == np.isfinite(ro["X"]).all(): bad.append("frame %d: finiteness of X differs" % k)
    if kw.get("map_capacity", 0) and not bad:
        mg, mo = g.map(), o.map()
        if mg.shape != mo.shape: bad.append("map shape device %s oracle %s" % (mg.shape, mo.shape))
    g.close(); o.close()
    dev = torch.device("cuda", 0)
    bufs = [torch.from_numpy(np.ascontiguousarray(s.T) if len(s) else np.zeros((3, 4), np.float32)).to(dev) for s in frames]
    fr = [(b.data_ptr(), len(s), b.shape[1]) for b, s in zip(bufs, frames)]
    nb = api.Node(ctx, **kw)
    burst = nb.push_many_device(fr)
    for k, (a, b) in enumerate(zip(burst, got)):
        same = all(np.array_equal(a[key], b[key], equal_nan=True) for key in ("X", "pred_stds", "pose", "quat")) and all(a[key] == b[key] for key in ("solved", "diverged", "n_kept", "map_rows"))
        if not same: bad.append("burst frame %d differs from the frame-by-frame push" % k)
    nb.close()
    return bad


# ---- sort-key stress: ties by the ten thousand ----------------------------------------------------------------------------------------------------
def tie_cases():
    """[(name, scan1, scan2)]: scans made of a few distinct rows repeated many times, unit directions scaled to one range (two distinct float ranges in 60 000 rows),
    a scan whose first half is ONE row."""
    rng = np.random.default_rng(5)
    a0, b0 = pools()[0]
    cases = []
    for distinct, n in ((1, 50000), (5, 200000), (40, 120000), (1000, 300000)):
        pts = a0[rng.choice(len(a0), distinct, replace=False)]
        pts = pts[np.linalg.norm(pts, axis=1) > 0] if distinct > 1 else a0[np.linalg.norm(a0, axis=1) > 1][:1]
        cases.append(("%d distinct rows x %d" % (len(pts), n), pts[rng.integers(0, len(pts), n)].copy(), pts[rng.integers(0, len(pts), n)].copy()))
    shell = a0[np.linalg.norm(a0, axis=1) > 1][:60000].astype(np.float64)
    shell = (shell / np.linalg.norm(shell, axis=1, keepdims=True) * 8.0).astype(np.float32)          # ranges within an ulp or two of 8: thousands of exact ties
    cases.append(("unit directions x 8.0", shell, shell[::-1].copy()))
    half = a0.copy(); half[: len(half) // 2] = half[0]                                                  # half the scan is ONE row
    cases.append(("half the scan one row", half, b0))
    return cases


# ---- launch-shape and path-selection knobs -----------------------------------------------------------------------------------------------------------
# (force_exact, guard_scale and lut_polar_quantile are not here: they change WHICH points wait for the literal formulas -- the decisions stay, tests/test_gpu_parity.py
# ::test_fast_classification_equals_literal_evaluation -- but a parked point enters the sums as a run of one, so the float partial sums are grouped differently)
KNOBS = dict(lds_slots=[0, 32, 64, 200, 500, 1800, 100000], acc_pts=[1, 2, 8, 64], acc_blocks=[1, 100, 1536, 4000, 20000], kf_pts=[1, 2, 8],
             batch_parts=[0, 1, 2, 3, 8], rs_cap=[0, 64, 500, 3000], rs_max_cell=[0, 1, 4, 100], exec_bits_lds=[0, 1], lds_rank=[-1, 0, 1], exec_pairwise=[-1, 0, 1],
             graph=[-1, 0, 1], fuse_solve=[-1, 0, 1])



def run_knob_draws(draws, seed, log=None):
    """Random settings of the knobs above, each in a fresh context: a single solve (twice: the second call may be a graph replay) and a ragged 40-pair device batch
    (twice) must give the bits of the default settings.  Returns the knob settings that did not."""
    import torch
    from icet_amd import api
    failed = []
    rng = np.random.default_rng(seed)
    pl = pools()
    ref_ctx = api.Context()
    dev = torch.device("cuda", 0)
    # one ragged 40-pair batch resident in HBM
    pairs = []
    for k in range(40):
        a0, b0 = pl[k % len(pl)]
        m = [a0.shape[0], 30000, 9000][k % 3]
        pairs.append((np.ascontiguousarray(a0[:m]), np.ascontiguousarray(b0[:m])))
    bufs = [(torch.from_numpy(np.ascontiguousarray(a.T)).to(dev), torch.from_numpy(np.ascontiguousarray(b.T)).to(dev)) for a, b in pairs]
    d1 = [(t.data_ptr(), t.shape[1], t.shape[1]) for t, _ in bufs]; d2 = [(t.data_ptr(), t.shape[1], t.shape[1]) for _, t in bufs]
    prm = api.Params(5, 24, 75, 25, 0.1, 0.1, 0)
    x0 = torch.zeros(40, 6, device=dev); x0[::3, 0] = 0.05
    def batch(ctx):
        out = torch.zeros(40, 48, device=dev)
        ctx.solve_batch_device(d1, d2, prm, out.data_ptr(), x0.data_ptr()); torch.cuda.synchronize()
        return out.cpu().numpy()
    sa, sb = pl[1]
    ref_single = ref_ctx.solve(sa, sb, 5, np.array([0.02, 0, 0, 0, 0, 0.001], np.float32), 24, 75)
    ref_batch = batch(ref_ctx)
    bad = 0
    for dno in range(draws):
        ctx = api.Context()
        kn = {k: (rng.choice(v) if rng.random() < 0.5 else None) for k, v in KNOBS.items()}
        kn = {k: (float(v) if isinstance(v, (float, np.floating)) else int(v)) for k, v in kn.items() if v is not None}
        for k, v in kn.items(): ctx.set_option(k, v)
        r = ctx.solve(sa, sb, 5, np.array([0.02, 0, 0, 0, 0, 0.001], np.float32), 24, 75)
        r2 = ctx.solve(sa, sb, 5, np.array([0.02, 0, 0, 0, 0, 0.001], np.float32), 24, 75)       # (a second call: the graph replay path when "graph" allows it)
        ob = batch(ctx); ob2 = batch(ctx)
        ok_s = all(np.array_equal(r[k].view(np.uint32), ref_single[k].view(np.uint32)) and np.array_equal(r2[k].view(np.uint32), ref_single[k].view(np.uint32)) for k in ("X", "pred_stds", "cov"))
        ok_b = np.array_equal(ob.view(np.uint32), ref_batch.view(np.uint32)) and np.array_equal(ob2.view(np.uint32), ref_batch.view(np.uint32))
        bad += 0 if (ok_s and ok_b) else 1
        if log: log("draw %3d single=%s batch=%s  %s" % (dno, "ok" if ok_s else "DIFF", "ok" if ok_b else "DIFF", kn))
        if not (ok_s and ok_b): failed.append(kn)
        ctx.close()
    return failed
