"""Host-side mirror of det3d/models/readers/voxel_encoder.py ("ve:" below): dynamic 3-D voxelisation + per-voxel mean.

    VoxelFeatureNet(voxel_size, pc_range).forward(points (N, 1+F) [b,x,y,z,..]) -> (features (V, F) = mean of the voxel's rows [x y z f..],
                                                                                  coords (V, 4) int32 [b, z, y, x], grid [gz, gy, gx])   # ve:75-87

SURVEY 8f-4 row (after the PillarNeXt hot path): grouping by torch.unique on one int64 voxel key per kept point -- ascending key ==
the reference's lexicographic torch.unique(dim=0) over [b, x, y, z] rows (ve:63) -- and index_add sums; no kernel of its own."""
import numpy as np
import torch
from torch import nn


def scatter_mean(src, index, num):
    """torch_scatter.scatter_mean(src, index, dim=0) (ve:20): per-index sum, then true divide by the count."""
    s = torch.zeros((num, src.shape[1]), dtype=src.dtype, device=src.device).index_add_(0, index, src)
    cnt = torch.zeros((num,), dtype=src.dtype, device=src.device).index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
    return s / cnt.clamp(min=1).unsqueeze(1)


class DynamicVoxelEncoder(nn.Module):
    """ve:12-22."""

    def forward(self, inputs, unq_inv, num_voxels=None):
        if num_voxels is None:
            num_voxels = int(unq_inv.max().item()) + 1 if unq_inv.numel() else 0
        return scatter_mean(inputs, unq_inv, num_voxels)


def grid_of(pc_range, voxel_size):
    """np.round((max - min) / voxel) in fp64, half-to-even (ve:40-41, pillar_encoder.py:87-89)."""
    g = (np.asarray(pc_range, np.float64)[3:] - np.asarray(pc_range, np.float64)[:3]) / np.asarray(voxel_size, np.float64)
    return np.round(g, 0, g).astype(np.int64)


class VoxelNet(nn.Module):
    """ve:25-72: fp32 (x - min) / voxel with an IEEE divide, float range test on all three axes, truncation, unique voxel rows."""

    def __init__(self, voxel_size, pc_range):
        super().__init__()
        self.voxel_size = np.array(voxel_size)
        self.pc_range = np.array(pc_range)

    def forward(self, points):
        g = grid_of(self.pc_range, self.voxel_size)                                            # x, y, z
        vs = torch.from_numpy(self.voxel_size).type_as(points).to(points.device)
        pr = torch.from_numpy(self.pc_range).type_as(points).to(points.device)
        pc = (points[:, 1:4] - pr[:3].view(-1, 3)) / vs.view(-1, 3)
        mask = ((pc[:, 0] >= 0) & (pc[:, 0] < g[0]) & (pc[:, 1] >= 0) & (pc[:, 1] < g[1]) & (pc[:, 2] >= 0) & (pc[:, 2] < g[2]))
        points, pc = points[mask], pc[mask].long()
        b = points[:, 0].long()
        key = ((b * int(g[0]) + pc[:, 0]) * int(g[1]) + pc[:, 1]) * int(g[2]) + pc[:, 2]          # ascending key == sorted [b, x, y, z] rows
        unq, unq_inv = torch.unique(key, return_inverse=True)
        z = unq % int(g[2])
        t = unq // int(g[2])
        y = t % int(g[1])
        t = t // int(g[1])
        x = t % int(g[0])
        bb = t // int(g[0])
        coords = torch.stack([bb, z, y, x], 1).int()                                              # ve:70: unq[:, [0, 3, 2, 1]]
        return points[:, 1:], coords, unq_inv, g[[2, 1, 0]]


class VoxelFeatureNet(nn.Module):
    """ve:75-87."""

    def __init__(self, voxel_size, pc_range):
        super().__init__()
        self.voxelization = VoxelNet(voxel_size, pc_range)
        self.voxel_encoder = DynamicVoxelEncoder()

    def forward(self, points):
        features, coords, unq_inv, grid_size = self.voxelization(points)
        return self.voxel_encoder(features, unq_inv, coords.shape[0]), coords, grid_size
