"""Synthetic nuScenes/Waymo-shaped inputs.  numpy only; seeds are explicit.

Row format is the reference's collated point buffer (det3d/datasets/loader/collate.py:15-22):
``[batch_idx, x, y, z, intensity, dt]`` fp32, batch index stored as a float.
"""
import numpy as np

# name -> (points per sample, voxel_size, pc_range)   (BASELINE.json configs / SURVEY.md section 8 table)
CONFIGS = {
    "C1": dict(n=50_000, voxel_size=(0.2, 0.2, 8.0), pc_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0)),
    "C2": dict(n=300_000, voxel_size=(0.075, 0.075, 8.0), pc_range=(-54.0, -54.0, -5.0, 54.0, 54.0, 3.0)),
    "C2ref": dict(n=300_000, voxel_size=(0.075, 0.075, 8.0), pc_range=(-50.4, -50.4, -5.0, 50.4, 50.4, 3.0)),
    "C4": dict(n=180_000, voxel_size=(0.1, 0.1, 6.0), pc_range=(-75.2, -75.2, -2.0, 75.2, 75.2, 4.0)),
    "C5": dict(n=540_000, voxel_size=(0.1, 0.1, 6.0), pc_range=(-75.2, -75.2, -2.0, 75.2, 75.2, 4.0)),
    "C5ref": dict(n=540_000, voxel_size=(0.075, 0.075, 6.0), pc_range=(-76.8, -76.8, -2.0, 76.8, 76.8, 4.0)),
}


def uniform_cloud(n, pc_range, seed, batch_idx=0):
    """D-uniform: r ~ U(0, 1.05 R), theta ~ U -- worst case (~1.2 points per pillar at C2)."""
    rng = np.random.default_rng(seed)
    R = 0.5 * (pc_range[3] - pc_range[0])
    cx, cy = 0.5 * (pc_range[3] + pc_range[0]), 0.5 * (pc_range[4] + pc_range[1])
    r = rng.uniform(0.0, 1.05 * R, n)
    th = rng.uniform(0.0, 2 * np.pi, n)
    pts = np.empty((n, 6), np.float32)
    pts[:, 0] = batch_idx
    pts[:, 1] = cx + r * np.cos(th)
    pts[:, 2] = cy + r * np.sin(th)
    pts[:, 3] = rng.uniform(pc_range[2], pc_range[5], n)
    pts[:, 4] = rng.uniform(0.0, 1.0, n)
    pts[:, 5] = rng.integers(0, 10, n) * 0.05
    return pts


def sweep_cloud(n, pc_range, seed, batch_idx=0, beams=32, sweeps=10):
    """D-sweep: ring-structured LiDAR returns (beams x azimuth steps x sweeps, ground plane plus
    smoothed facades and box-shaped clusters) -- realistic ~3 points per pillar at 0.075 m."""
    rng = np.random.default_rng(seed)
    R = 0.5 * (pc_range[3] - pc_range[0])
    az_steps = max(int(np.ceil(n / (beams * sweeps))), 1)
    az = np.linspace(0.0, 2 * np.pi, az_steps, endpoint=False)
    elev = np.deg2rad(np.linspace(-30.0, 10.0, beams))
    # facade range per azimuth: smoothed random walk in [8, 1.1 R]
    wall = rng.uniform(8.0, 1.1 * R, az_steps)
    k = min(31, az_steps)
    wall = sum(np.roll(wall, s - k // 2) for s in range(k)) / k  # circular box smoothing
    out = []
    sensor_h = 1.84
    for s in range(sweeps):
        ego = np.array([0.4 * s * np.cos(0.3), 0.4 * s * np.sin(0.3)])
        a = az[None, :] + rng.normal(0, 2e-4, (beams, az_steps))
        e = elev[:, None] + rng.normal(0, 2e-4, (beams, az_steps))
        with np.errstate(divide="ignore"):
            ground_r = np.where(e < -1e-3, sensor_h / np.tan(-e), np.inf)
        rr = np.minimum(ground_r, wall[None, :])
        z = np.where(ground_r <= wall[None, :], -sensor_h, rr * np.tan(e))
        x = rr * np.cos(a) + ego[0]
        y = rr * np.sin(a) + ego[1]
        p = np.stack([x, y, z + rng.normal(0, 0.02, x.shape), rng.uniform(0, 1, x.shape), np.full(x.shape, 0.05 * s)], -1)
        out.append(p.reshape(-1, 5))
    p = np.concatenate(out)
    # ~60 box-shaped clusters replace a slice of the returns
    ncl = 60
    m = max(len(p) // 40, 1)
    ctr = rng.uniform(-0.7 * R, 0.7 * R, (ncl, 2))
    which = rng.integers(0, ncl, m)
    cl = np.empty((m, 5))
    cl[:, 0:2] = ctr[which] + rng.uniform(-2.2, 2.2, (m, 2)) * np.array([1.0, 0.45])
    cl[:, 2] = rng.uniform(-1.6, 0.2, m)
    cl[:, 3] = rng.uniform(0, 1, m)
    cl[:, 4] = rng.integers(0, sweeps, m) * 0.05
    p[rng.choice(len(p), m, replace=False)] = cl
    p = p[np.isfinite(p).all(1)]
    if len(p) >= n:
        p = p[rng.permutation(len(p))[:n]]
    else:
        p = np.concatenate([p, p[rng.integers(0, len(p), n - len(p))] + rng.normal(0, 0.01, (n - len(p), 5))])
    cx, cy = 0.5 * (pc_range[3] + pc_range[0]), 0.5 * (pc_range[4] + pc_range[1])
    pts = np.empty((n, 6), np.float32)
    pts[:, 0] = batch_idx
    pts[:, 1] = p[:, 0] + cx
    pts[:, 2] = p[:, 1] + cy
    pts[:, 3:6] = p[:, 2:5]
    return pts


def push_outside(pts, pc_range, share, seed):
    """~1-2 % of the points lie outside the x/y range (they exercise the range mask, pillar_encoder.py:98-104).  A seeded
    `share` of the rows is moved radially to 1.01 .. 1.3 x the half-range (in place)."""
    n = len(pts)
    k = int(round(n * share))
    if k <= 0:
        return pts
    rng = np.random.default_rng(seed)
    idx = rng.choice(n, k, replace=False)
    cx, cy = 0.5 * (pc_range[3] + pc_range[0]), 0.5 * (pc_range[4] + pc_range[1])
    R = 0.5 * (pc_range[3] - pc_range[0])
    dx, dy = pts[idx, 1].astype(np.float64) - cx, pts[idx, 2].astype(np.float64) - cy
    m = np.maximum(np.maximum(np.abs(dx), np.abs(dy)), 1e-3)
    f = R * rng.uniform(1.01, 1.3, k) / m
    pts[idx, 1] = (cx + dx * f).astype(np.float32)
    pts[idx, 2] = (cy + dy * f).astype(np.float32)
    return pts


def make_batch(config="C2", batch=1, dist="uniform", frame0=0, n=None, outside=None):
    """Collated batch (sum N, 6) fp32; seeds follow SURVEY 8d: cloud seed = 1000 + frame index.  `outside`: share of the rows pushed
    outside the x/y range (default: 1.5 % of a sweep cloud; the uniform disc already reaches 1.05 R)."""
    cfg = CONFIGS[config]
    gen = uniform_cloud if dist == "uniform" else sweep_cloud
    share = (0.015 if dist != "uniform" else 0.0) if outside is None else outside
    return np.concatenate([push_outside(gen(n or cfg["n"], cfg["pc_range"], 1000 + frame0 + b, batch_idx=b), cfg["pc_range"], share, 77000 + frame0 + b)
                           for b in range(batch)])


def raw_sweeps(batch_pts, sweep_dt=0.05):
    """The collated batch taken apart into what a nuScenes sample is BEFORE det3d/datasets/nuscenes/nusc.py:76-121 merges it: per frame
    one raw (n_s, 5) array [x, y, z, intensity, ring] per sweep (the points that carry that sweep's time lag, in the key frame's
    coordinates minus a per-sweep translation, so that the merge has a transform to apply), plus the segment table for
    io.SweepMerger.  Returns (raw (M, 5) fp32, segments).  Merging gives the batch back up to the fp32 rounding of x - t + t."""
    raws, segs, o = [], [], 0
    B = int(batch_pts[:, 0].max()) + 1 if len(batch_pts) else 0
    for b in range(B):
        f = batch_pts[batch_pts[:, 0] == b]
        sidx = np.rint(f[:, 5] / sweep_dt).astype(np.int64)
        for s in np.unique(sidx):
            p = f[sidx == s]
            T = None
            r = np.zeros((len(p), 5), np.float32)
            r[:, :4] = p[:, 1:5]
            if s > 0:
                t = np.array([0.4 * s * np.cos(0.3), 0.4 * s * np.sin(0.3), 0.01 * s])
                T = np.eye(4)
                T[:3, 3] = t
                r[:, :3] = (p[:, 1:4].astype(np.float64) - t).astype(np.float32)
            raws.append(r)
            segs.append(dict(begin=o, end=o + len(r), batch=b, time=float(sweep_dt * s), radius=0.0, transform=T))
            o += len(r)
    return (np.concatenate(raws) if raws else np.zeros((0, 5), np.float32)), segs


def pfn_params(num_input_features=5, num_filters=(64, 64), seed=0):
    """Seeded PFN weights with non-trivial BN running statistics (weights seed 0 by convention).
    Returns a list of per-layer dicts W (units, cin), gamma, beta, mean, var (fp32)."""
    rng = np.random.default_rng(seed)
    cin = num_input_features + 5
    layers = []
    for i, f in enumerate(num_filters):
        last = i == len(num_filters) - 1
        units = f if last else f // 2
        bound = 1.0 / np.sqrt(cin)
        layers.append(dict(
            W=rng.uniform(-bound, bound, (units, cin)).astype(np.float32),
            gamma=rng.uniform(0.5, 1.5, units).astype(np.float32),
            beta=rng.uniform(-0.3, 0.3, units).astype(np.float32),
            mean=rng.uniform(-0.5, 0.5, units).astype(np.float32),
            var=rng.uniform(0.3, 2.0, units).astype(np.float32),
        ))
        cin = f
    return layers


def clustered_boxes(n, seed, spread=40.0, n_clusters=None):
    """Score-sorted NMS test set: boxes jittered around cluster centres (SURVEY 8d NMS inputs)."""
    rng = np.random.default_rng(seed)
    ncl = n_clusters or max(n // 14, 1)
    ctr = rng.uniform(-spread, spread, (ncl, 2))
    base_dim = rng.uniform([1.5, 0.6, 1.0], [5.0, 2.2, 2.5], (ncl, 3))
    base_h = rng.uniform(-np.pi, np.pi, ncl)
    w = rng.integers(0, ncl, n)
    b = np.empty((n, 7), np.float32)
    b[:, 0:2] = ctr[w] + rng.normal(0, 0.45, (n, 2))
    b[:, 2] = rng.uniform(-1.5, 0.5, n)
    b[:, 3:6] = base_dim[w] * rng.uniform(0.85, 1.15, (n, 3))
    b[:, 6] = base_h[w] + rng.normal(0, 0.08, n)
    scores = np.sort(rng.uniform(0.1, 1.0, n).astype(np.float32))[::-1].copy()
    # strictly decreasing scores (torch.sort tie order is unspecified -- box_torch_ops.py:13)
    scores = (scores - np.arange(n, dtype=np.float32) * 1e-6).astype(np.float32)
    return b, scores
