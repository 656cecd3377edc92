"""Seeded synthetic lidar scan pairs for the BASELINE configs 2-5.

The reference ships only two real scan pairs (src/sample_data/frame_804/805.npy and
python/point_clouds/sample_pc_1/2.npy); its 64-channel Ouster CSV scans are missing blobs.  The
benchmark workloads therefore ray-cast a spinning lidar through a procedurally generated street
scene.  Written with torch ops only so that the same code runs on the CPU (tests, small cases) and
on the GPU (bench: 256 pairs are generated in HBM and never touch the host).

Scene (all in the frame of the first sensor pose, sensor 1.8 m above the ground):
  ground plane, 4 street walls (a rectangle of vertical planes with finite height), ~40 boxes and
  cylinders.  Because the reference's scrambled radial walk only ever finds clusters on surfaces that
  face the sensor, the scene is wall/box heavy on purpose.
Sensor: `rings` elevation channels uniform in [-22.3, +22.7] deg x `steps` azimuth steps; range noise
N(0, sigma); rays with no hit or range > max_range are dropped.  Scan 2 observes the same scene after
the ground-truth motion X_true = (x, y, z, roll, pitch, yaw) in the reference's convention
p_in_frame1 = R(angles)^T (p_in_frame2 + t) (src/icet.cpp:375-378), so a converged solve returns
X close to X_true.
"""
import math
import numpy as np
import torch

DEFAULT_MOTION = (0.50, 0.05, 0.01, 0.002, -0.001, 0.010)


def euler_R(phi, theta, psi):
    c, s = math.cos, math.sin
    return np.array([
        [c(theta) * c(psi), s(psi) * c(phi) + s(phi) * s(theta) * c(psi), s(phi) * s(psi) - s(theta) * c(phi) * c(psi)],
        [-s(psi) * c(theta), c(phi) * c(psi) - s(phi) * s(theta) * s(psi), s(phi) * c(psi) + s(theta) * s(psi) * c(phi)],
        [s(theta), -s(phi) * c(theta), c(phi) * c(theta)]], dtype=np.float64)


def make_scene(seed, n_objects=40, sensor_height=1.8):
    """Deterministic scene description (plain floats; built on the host from numpy's RandomState)."""
    rs = np.random.RandomState(seed)
    g = -sensor_height
    half_w = rs.uniform(8.0, 14.0)             # street half width  (walls at y = +-half_w)
    half_l = rs.uniform(35.0, 60.0)            # street half length (walls at x = +-half_l)
    wall_h = rs.uniform(6.0, 10.0)
    boxes, cyls = [], []
    for k in range(n_objects):
        for _ in range(50):
            cx = rs.uniform(-half_l + 2, half_l - 2); cy = rs.uniform(-half_w + 1, half_w - 1)
            if abs(cy) > 2.5 or abs(cx) > 6.0:
                # keep the driving corridor |y| < 2.5 mostly free near the sensor
                if not (abs(cy) < 2.0 and abs(cx) < 15.0):
                    break
        if k % 4 == 3:
            cyls.append((cx, cy, rs.uniform(0.15, 0.6), g, g + rs.uniform(3.0, 7.0)))
        else:
            sx, sy, sz = rs.uniform(0.8, 4.5), rs.uniform(0.8, 4.5), rs.uniform(1.0, 3.5)
            boxes.append((cx - sx / 2, cx + sx / 2, cy - sy / 2, cy + sy / 2, g, g + sz))
    return dict(ground=g, half_w=half_w, half_l=half_l, wall_top=g + wall_h, boxes=boxes, cyls=cyls)


def _raycast(scene, origin, dirs, max_range):
    """Nearest hit distance along unit rays `dirs` (M x 3 tensor) from `origin` (3 floats). inf = no hit."""
    ox, oy, oz = origin
    dx, dy, dz = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    inf = torch.full_like(dx, float("inf"))
    eps = 1e-9
    best = inf.clone()

    def upd(t, ok):
        nonlocal best
        t = torch.where(ok & (t > 0.05), t, inf)
        best = torch.minimum(best, t)

    # ground
    t = (scene["ground"] - oz) / torch.where(dz.abs() < eps, torch.full_like(dz, eps), dz)
    upd(t, dz < 0)
    if scene.get("ceiling") is not None:          # tunnel scenes: a horizontal plane above the sensor
        t = (scene["ceiling"] - oz) / torch.where(dz.abs() < eps, torch.full_like(dz, eps), dz)
        upd(t, dz > 0)
    # walls: y = +-half_w (|x| <= half_l), x = +-half_l (|y| <= half_w), ground <= z <= wall_top
    for yy in ((scene["half_w"],) if scene.get("one_wall") else (-scene["half_w"], scene["half_w"])):
        t = (yy - oy) / torch.where(dy.abs() < eps, torch.full_like(dy, eps), dy)
        x = ox + t * dx; z = oz + t * dz
        upd(t, (x.abs() <= scene["half_l"]) & (z >= scene["ground"]) & (z <= scene["wall_top"]))
    for xx in (-scene["half_l"], scene["half_l"]):
        t = (xx - ox) / torch.where(dx.abs() < eps, torch.full_like(dx, eps), dx)
        y = oy + t * dy; z = oz + t * dz
        upd(t, (y.abs() <= scene["half_w"]) & (z >= scene["ground"]) & (z <= scene["wall_top"]))
    # boxes (slab test)
    idx = 1.0 / torch.where(dx.abs() < eps, torch.full_like(dx, eps), dx)
    idy = 1.0 / torch.where(dy.abs() < eps, torch.full_like(dy, eps), dy)
    idz = 1.0 / torch.where(dz.abs() < eps, torch.full_like(dz, eps), dz)
    for (x0, x1, y0, y1, z0, z1) in scene["boxes"]:
        tx0 = (x0 - ox) * idx; tx1 = (x1 - ox) * idx
        ty0 = (y0 - oy) * idy; ty1 = (y1 - oy) * idy
        tz0 = (z0 - oz) * idz; tz1 = (z1 - oz) * idz
        tmin = torch.maximum(torch.maximum(torch.minimum(tx0, tx1), torch.minimum(ty0, ty1)), torch.minimum(tz0, tz1))
        tmax = torch.minimum(torch.minimum(torch.maximum(tx0, tx1), torch.maximum(ty0, ty1)), torch.maximum(tz0, tz1))
        upd(tmin, tmax >= tmin)
    # vertical cylinders
    a = dx * dx + dy * dy
    a_safe = torch.where(a < eps, torch.full_like(a, eps), a)
    for (cx, cy, rad, z0, z1) in scene["cyls"]:
        fx = ox - cx; fy = oy - cy
        b = 2.0 * (fx * dx + fy * dy)
        c = fx * fx + fy * fy - rad * rad
        disc = b * b - 4.0 * a_safe * c
        sq = torch.sqrt(torch.clamp(disc, min=0.0))
        t = (-b - sq) / (2.0 * a_safe)
        z = oz + t * dz
        upd(t, (disc >= 0) & (z >= z0) & (z <= z1))
    return torch.where(best <= max_range, best, inf)


def sensor_dirs(rings, steps, device, order="ring"):
    # +0.2 deg / half-step offsets keep every beam off the spherical-voxel edges (multiples of 7.5 or 3.75 deg):
    # a ring lying EXACTLY on an edge makes the bin of thousands of points depend on the last ulp of acosf,
    # which no real sensor does and which differs between glibc and the device math library.
    el = torch.linspace(math.radians(-22.5 + 0.2), math.radians(22.5 + 0.2), rings, dtype=torch.float64, device=device)
    az = (torch.arange(steps, dtype=torch.float64, device=device) + 0.5) * (2.0 * math.pi / steps)
    if order == "ring":          # ring index slow, azimuth fast
        E = el[:, None].expand(rings, steps).reshape(-1); A = az[None, :].expand(rings, steps).reshape(-1)
    elif order == "azimuth":     # azimuth slow, ring fast (column-wise firing order)
        E = el[None, :].expand(steps, rings).reshape(-1); A = az[:, None].expand(steps, rings).reshape(-1)
    else:
        raise ValueError("order must be 'ring' or 'azimuth'")
    ce = torch.cos(E)
    return torch.stack([ce * torch.cos(A), ce * torch.sin(A), torch.sin(E)], 1)


def make_scan(scene, pose, noise_seed, rings=64, steps=2048, sigma=0.02, max_range=120.0, device="cpu", order="ring"):
    """One scan in the sensor frame. pose = (t_s[3], R_s[3x3]) of the sensor in the scene frame.
    Returns a float32 (3, N) tensor: column-major N x 3 (x[N] | y[N] | z[N]), N = rays that hit."""
    t_s, R_s = pose
    d = sensor_dirs(rings, steps, device, order)
    Rt = torch.as_tensor(np.asarray(R_s, np.float64), device=device)
    dw = d @ Rt.T                                        # world directions
    rng = _raycast(scene, [float(v) for v in t_s], dw, max_range)
    gen = torch.Generator(device=device); gen.manual_seed(int(noise_seed))
    noise = torch.randn(rng.shape[0], generator=gen, device=device, dtype=torch.float64) * sigma
    hit = torch.isfinite(rng)
    r = (rng + noise)[hit]
    pts = d[hit] * r[:, None]
    return pts.to(torch.float32).T.contiguous()


def make_pair(scene_seed=1000, noise_seed=1001, motion=DEFAULT_MOTION, rings=64, steps=2048, device="cpu", order="ring", sigma=0.02):
    """(scan1, scan2, X_true); scans are float32 (3, N) tensors (column-major N x 3)."""
    scene = make_scene(scene_seed)
    X = np.asarray(motion, np.float64)
    R = euler_R(X[3], X[4], X[5])
    R_s = R.T
    t_s = R.T @ X[:3]
    s1 = make_scan(scene, (np.zeros(3), np.eye(3)), noise_seed * 2 + 1, rings, steps, sigma, device=device, order=order)
    s2 = make_scan(scene, (t_s, R_s), noise_seed * 2 + 2, rings, steps, sigma, device=device, order=order)
    return s1, s2, X.astype(np.float32)


def make_pair_with_moving_objects(scene_seed=1000, noise_seed=1001, motion=DEFAULT_MOTION, shift=(1.2, 0.4), every=3, rings=64, steps=2048, device="cpu"):
    """A pair in which every `every`-th box of the scene has moved by `shift` (m, in x / y) between the two scans: the input the
    moving-object rejection of the Python variant (python/ICET_spherical.py:175-250) exists for.  Returns (scan1, scan2, X_true)."""
    scene = make_scene(scene_seed)
    moved = dict(scene)
    moved["boxes"] = [(b[0] + shift[0], b[1] + shift[0], b[2] + shift[1], b[3] + shift[1], b[4], b[5]) if k % every == 0 else b for k, b in enumerate(scene["boxes"])]
    X = np.asarray(motion, np.float64)
    R = euler_R(X[3], X[4], X[5])
    s1 = make_scan(scene, (np.zeros(3), np.eye(3)), noise_seed * 2 + 1, rings, steps, device=device)
    s2 = make_scan(moved, (R.T @ X[:3], R.T), noise_seed * 2 + 2, rings, steps, device=device)
    return s1, s2, X.astype(np.float32)


def batch_motion(k):
    """Motion of pair k of the batched configs: U(+-0.6, +-0.05, +-0.02 m; +-0.005, +-0.005, +-0.02 rad), seed 5000+k."""
    rs = np.random.RandomState(5000 + k)
    lim = np.array([0.6, 0.05, 0.02, 0.005, 0.005, 0.02])
    return rs.uniform(-lim, lim)


def make_degenerate_scene(kind="tunnel", sensor_height=1.8):
    """Scenes whose geometry leaves solution axes unobservable -- the input ICET::checkCondition (src/icet.cpp:443-492) exists for.
    "tunnel": walls y = +-4.5 m, floor, ceiling 2.5 m above the sensor, no end walls within range (translation along x is free: with 0.5-1 mm
    of range noise cond(H^T W H) = 2e6, one axis pruned, and the sign of the pruned eigenvector differs between the two noise levels);
    "wall": the floor and ONE wall at y = 3 m; "ground": the floor alone (x, y and yaw are free)."""
    g = -sensor_height
    far = 1.0e9
    if kind == "tunnel":
        return dict(ground=g, half_w=4.5, half_l=far, wall_top=2.5, ceiling=2.5, boxes=[], cyls=[])
    if kind == "wall":
        return dict(ground=g, half_w=3.0, half_l=far, wall_top=6.0, ceiling=None, boxes=[], cyls=[], one_wall=True)
    if kind == "ground":
        return dict(ground=g, half_w=far, half_l=far, wall_top=g, ceiling=None, boxes=[], cyls=[])
    raise ValueError("kind must be tunnel, wall or ground")


# name -> (kind, sigma, motion): the degenerate pairs of tests/golden/golden_degenerate.npz and tests/test_gpu_parity.py.  What the CPU restatement
# does on them (7 iterations, 75 x 24): tunnel_s05 / tunnel_s10: one axis pruned, pred_stds[0] = +1.0 / -1.0 (the sign of an eigenvector);
# wall_s10: one axis, cond 7e6; wall_s30: cond 9.7e5, just below checkCondition's cutoff, nothing pruned; ground_s05 / ground_s02: two / three
# axes; ground_s10_m: nothing pruned in the first iteration, one axis from the second on.
_SMALL_MOTION = (0.1, 0.02, 0.0, 0.001, 0.0, 0.005)
DEGENERATE_SCENES = {
    "tunnel_s05": ("tunnel", 0.0005, (0, 0, 0, 0, 0, 0)), "tunnel_s10": ("tunnel", 0.001, (0, 0, 0, 0, 0, 0)), "tunnel_s10_m": ("tunnel", 0.001, _SMALL_MOTION),
    "wall_s10": ("wall", 0.001, (0, 0, 0, 0, 0, 0)), "wall_s30": ("wall", 0.003, (0, 0, 0, 0, 0, 0)),
    "ground_s02": ("ground", 0.0002, (0, 0, 0, 0, 0, 0)), "ground_s05": ("ground", 0.0005, (0, 0, 0, 0, 0, 0)), "ground_s10_m": ("ground", 0.001, _SMALL_MOTION),
}


def make_degenerate_named(name, device="cpu"):
    kind, sigma, motion = DEGENERATE_SCENES[name]
    return make_degenerate_pair(kind, sigma=sigma, motion=motion, device=device)


def make_degenerate_pair(kind="tunnel", noise_seed=7001, motion=(0, 0, 0, 0, 0, 0), sigma=0.002, rings=64, steps=2048, device="cpu", order="ring", max_range=120.0):
    """(scan1, scan2, X_true) of a degenerate scene (make_degenerate_scene), range noise N(0, sigma)."""
    scene = make_degenerate_scene(kind)
    X = np.asarray(motion, np.float64)
    R = euler_R(X[3], X[4], X[5])
    s1 = make_scan(scene, (np.zeros(3), np.eye(3)), noise_seed * 2 + 1, rings, steps, sigma, max_range, device=device, order=order)
    s2 = make_scan(scene, (R.T @ X[:3], R.T), noise_seed * 2 + 2, rings, steps, sigma, max_range, device=device, order=order)
    return s1, s2, X.astype(np.float32)


def make_batch_pair(k, rings=64, steps=2048, device="cpu", order="ring"):
    """Pair k of configs 3/4: scene seed 1000+2k, noise seed 1001+2k, motion seed 5000+k."""
    return make_pair(1000 + 2 * k, 1001 + 2 * k, batch_motion(k), rings, steps, device, order)


def make_sequence(n_frames, scene_seed=2000, noise_seed=2001, motion=DEFAULT_MOTION, rings=64, steps=2048, device="cpu", order="ring", sigma=0.02):
    """A drive through one scene: frame k+1 is observed after `motion` relative to frame k (same convention as make_pair,
    so registering frame k -> k+1 should return about `motion`).  Returns a list of float32 (3, N_k) tensors."""
    scene = make_scene(scene_seed)
    X = np.asarray(motion, np.float64)
    R = euler_R(X[3], X[4], X[5])
    R_step, t_step = R.T, R.T @ X[:3]
    t, Rw = np.zeros(3), np.eye(3)
    scans = []
    for k in range(n_frames):
        scans.append(make_scan(scene, (t.copy(), Rw.copy()), noise_seed * 2 + 1 + k, rings, steps, sigma, device=device, order=order))
        t = t + Rw @ t_step
        Rw = Rw @ R_step
    return scans


def real_batch_rotation(k):
    """Rotation of pair k of the REAL-data batch (bench.py `sample_batch`, tests): the reference's two sample pairs alternate (even k: frame_804/805, odd k:
    sample_pc_1/2) and BOTH scans of pair k are turned by one small rotation (roll / pitch +-0.01, yaw +-0.05 rad, seed 7000 + k), so that every pair's rows
    fall into other voxels and sort differently while the registration stays the pair's and the invalid returns stay exact-zero rows.  float32 3 x 3."""
    ang = np.random.RandomState(7000 + k).uniform(-1, 1, 3) * np.array([0.01, 0.01, 0.05])
    return euler_R(*ang).astype(np.float32)
