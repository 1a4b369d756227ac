"""Host-side mirror of det3d/models/readers/voxel_encoder.py ("ve:" below): dynamic 3-D voxelisation + per-voxel mean, on the HIP grouping
kernel of csrc/group.hip (include/pnx.h: pnx_group_points, mode PNX_GROUP_VOXEL).

    VoxelFeatureNet(voxel_size, pc_range).forward(points (N, 1+F) [b,x,y,z,..]) -> (features (V, F+3) = mean of the voxel's rows [x y z f..],
                                                                                  coords (V, 4) int32 [b, z, y, x], grid [gz, gy, gx])   # ve:75-87

Same class names, constructor arguments and return values as the reference.  The voxel of a point follows from the reference's fp32 arithmetic
((x - min) / voxel with an IEEE divide, float range test, truncation); the rank of a voxel in torch.unique(dim=0)'s order comes from a key-order
occupancy bitmap + popcount prefix instead of a sort; the mean is an exact fp64 sum divided in fp32.  CUDA tensors only: there is no CPU path."""
import numpy as np
import torch
from torch import nn

from . import ops
from ._lib import PNX_GROUP_VOXEL, PnxError


def grid_of(pc_range, voxel_size):
    """np.round((max - min) / voxel) in fp64, half-to-even (ve:40-41, pillar_encoder.py:87-89)."""
    g = (np.asarray(pc_range, np.float64)[3:] - np.asarray(pc_range, np.float64)[:3]) / np.asarray(voxel_size, np.float64)
    return np.round(g, 0, g).astype(np.int64)


def batch_of(points, batch_size=None):
    """Number of samples: the caller's, else what the reference's index arithmetic implies (largest batch index + 1; one host sync)."""
    if batch_size is not None:
        return int(batch_size)
    return int(points[:, 0].max().item()) + 1 if points.shape[0] else 1


def _device_points(points, who):
    """The readers run on the HIP kernels only.  CPU tensors raise on purpose (the reference's torch_scatter path accepted them): this package is the
    MI355X product path, which carries no CPU fallback -- a silent one would void every parity claim; the torch statement of the
    same readers is kept with the tests' CPU checker, outside this package.  One more difference from the reference, shared by every grouping kernel: a row
    whose batch index lies outside [0, batch_size) is DROPPED (csrc/group.hip k_group_keys); the reference would index past the batch and grow it."""
    if not points.is_cuda:
        raise PnxError(f"{who}: points must be a CUDA (ROCm) tensor; the readers have no CPU implementation")
    return points.contiguous().float()


class DynamicVoxelEncoder(nn.Module):
    """ve:12-22: scatter_mean of arbitrary per-point rows (differentiable; VoxelFeatureNet itself takes the means the grouping kernel produces)."""

    def forward(self, inputs, unq_inv, num_voxels=None):
        if num_voxels is None:
            num_voxels = int(unq_inv.max().item()) + 1 if unq_inv.numel() else 0
        s = torch.zeros((num_voxels, inputs.shape[1]), dtype=inputs.dtype, device=inputs.device).index_add_(0, unq_inv, inputs)
        cnt = torch.zeros((num_voxels,), dtype=inputs.dtype, device=inputs.device).index_add_(0, unq_inv, torch.ones_like(unq_inv, dtype=inputs.dtype))
        return s / cnt.clamp(min=1).unsqueeze(1)


class VoxelNet(nn.Module):
    """ve:25-72: returns (features (N', F+3) = the kept rows, coords (V, 4) int32 [b, z, y, x], unq_inv (N'), grid [gz, gy, gx])."""

    def __init__(self, voxel_size, pc_range):
        super().__init__()
        self.voxel_size = np.array(voxel_size)
        self.pc_range = np.array(pc_range)
        self._geom = ops.group_geom(self.pc_range, self.voxel_size, PNX_GROUP_VOXEL)

    def group(self, points, batch_size=None, want_mean=False):
        points = _device_points(points, "VoxelNet")
        return ops.group_points(points, batch_of(points, batch_size), self._geom, want_features=True, want_mean=want_mean)

    def forward(self, points, batch_size=None):
        r = self.group(points, batch_size)
        return r["features"], r["coords"], r["unq_inv"], grid_of(self.pc_range, self.voxel_size)[[2, 1, 0]]


class VoxelFeatureNet(nn.Module):
    """ve:75-87."""

    def __init__(self, voxel_size, pc_range):
        super().__init__()
        self.voxelization = VoxelNet(voxel_size, pc_range)
        self.voxel_encoder = DynamicVoxelEncoder()

    def forward(self, points, batch_size=None):
        r = self.voxelization.group(points, batch_size, want_mean=True)
        return r["mean"], r["coords"], grid_of(self.voxelization.pc_range, self.voxelization.voxel_size)[[2, 1, 0]]
