"""torch_cpu_reader.py -- TEST/BENCH INFRASTRUCTURE ONLY (same rules as oracle.py: tests/, smoke(), bench.py's cpu_baseline).

An own PyTorch-CPU statement of the reader's op sequence, in the reference's style (sort-unique + scatter; pillar_encoder.py:78-125,
:35-50, :174-182 and the dense canvas of sparse_resnet.py:63-68), used as the second CPU baseline of SURVEY.md 8(d): it is what a
PyTorch user gets on the host cores with torch.set_num_threads(k).  BatchNorm in eval mode."""
import numpy as np
import torch


def reader_forward(points, pc_range, voxel_size, layers, batch=1, want_canvas=True):
    pts = torch.from_numpy(np.ascontiguousarray(points))
    pc_min = torch.tensor(pc_range[:3], dtype=torch.float32)
    vs = torch.tensor(voxel_size, dtype=torch.float32)
    grid = np.round((np.asarray(pc_range[3:], np.float64) - np.asarray(pc_range[:3], np.float64)) / np.asarray(voxel_size, np.float64)).astype(np.int64)
    gx, gy = int(grid[0]), int(grid[1])
    c = (pts[:, 1:4] - pc_min) / vs                                         # pe:95-96
    keep = (c[:, 0] >= 0) & (c[:, 0] < gx) & (c[:, 1] >= 0) & (c[:, 1] < gy)  # pe:98-101
    pts, c = pts[keep], c[keep]
    ci = c.long()
    key = (pts[:, 0].long() * gx + ci[:, 0]) * gy + ci[:, 1]                 # == unique rows of [b, xi, yi]  (pe:109-110)
    unq, inv = torch.unique(key, return_inverse=True)
    P = unq.shape[0]
    cnt = torch.zeros(P, dtype=torch.float32).index_add_(0, inv, torch.ones_like(inv, dtype=torch.float32))
    mean = torch.zeros((P, 3), dtype=torch.float32).index_add_(0, inv, pts[:, 1:4]) / cnt[:, None]  # scatter_mean (pe:113)
    f_cluster = pts[:, 1:4] - mean[inv]
    f_center = pts[:, 1:3] - (ci[:, :2].float() * vs[:2] + vs[:2] / 2 + pc_min[:2])
    x = torch.cat([pts[:, 1:], f_cluster, f_center], dim=1)
    for i, L in enumerate(layers):                                           # PFNLayer (pe:35-50), BN eval
        W = torch.from_numpy(L["W"])
        a = torch.from_numpy(L["gamma"]) / torch.sqrt(torch.from_numpy(L["var"]) + 1e-3)
        y = torch.relu((x @ W.t() - torch.from_numpy(L["mean"])) * a + torch.from_numpy(L["beta"]))
        m = torch.zeros((P, y.shape[1]), dtype=torch.float32).scatter_reduce(0, inv[:, None].expand_as(y), y, "amax", include_self=True)
        x = m if i == len(layers) - 1 else torch.cat([y, m[inv]], dim=1)
    if not want_canvas:
        return x
    canvas = torch.zeros((batch, gy, gx, 64), dtype=torch.float32)           # dense NHWC canvas (sparse_resnet.py:63-68 .dense())
    b = unq // (gx * gy)
    xi = (unq // gy) % gx
    yi = unq % gy
    canvas[b, yi, xi] = x
    return canvas
