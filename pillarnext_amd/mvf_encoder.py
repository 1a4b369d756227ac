"""Host-side mirror of det3d/models/readers/mvf_encoder.py ("mvf:" below): the multi-view (pillar + cylinder) reader of the
mvf18_aspp detectors (configs/models/reader/mvf_encoder.yaml, configs/experiments/waymo_det_mvf18_aspp_iou_car.yaml).

    MVFFeatureNet(in_channels, voxel_size, pc_range, cylinder_size, cylinder_range, num_filters, layer_nums, ds_layer_strides,
                  ds_num_filters, kernel_size, out_channels).forward(points) -> dense (B, out_channels, gy / ds, gx / ds) map    # mvf:257-327

Same class names, constructor arguments and state-dict keys as the reference.  SURVEY 8f-4 row (after the PillarNeXt hot path):
the two point-to-cell groupings (which CLAMP the cell index instead of dropping points, mvf:57-62, 111-116) are torch.unique over
one int64 key per point, the PFN layers are reader.PFNLayer on the HIP scatter-max (pnx_scatter_max), the per-view sparse
ResNets are the masked-dense blocks of models.py (spconv is absent from the image: like the backbone, that part cannot be pinned),
the rest is the reference's arithmetic restated on torch ops."""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from .reader import PFNLayer
from .voxel_encoder import grid_of, scatter_mean


class PointNet(nn.Module):
    """Linear(no bias) + BatchNorm1d(eps 1e-3, momentum 0.01) + ReLU per point (mvf:19-37)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.linear = nn.Linear(in_channels, out_channels, bias=False)
        self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)

    def forward(self, points):
        return F.relu(self.norm(self.linear(points)))


def _cells(points3, b, pc_range, voxel_size, grid):
    """(x - min) / voxel in fp32, CLAMPED to the grid (mvf:57-62), truncated; cell rows [b, c0, c1] made unique (mvf:67-69)."""
    vs = torch.from_numpy(voxel_size).type_as(points3).to(points3.device)
    pr = torch.from_numpy(pc_range).type_as(points3).to(points3.device)
    pc = (points3 - pr[:3].view(-1, 3)) / vs.view(-1, 3)
    for k in range(3):
        pc[:, k] = torch.clamp(pc[:, k], 0, int(grid[k]) - 1)
    pc = pc.long()
    key = (b * int(grid[0]) + pc[:, 0]) * int(grid[1]) + pc[:, 1]
    unq, unq_inv = torch.unique(key, return_inverse=True)
    c1 = unq % int(grid[1])
    t = unq // int(grid[1])
    c0 = t % int(grid[0])
    bb = t // int(grid[0])
    return pc, torch.stack([bb, c1, c0], 1).int(), unq_inv, vs, pr            # coords = unq[:, [0, 2, 1]]


def _decorate(points_rest, p3, pc, unq_inv, num, vs, pr):
    """[rest | xyz - cell mean | xy - cell centre] (mvf:71-83), evaluated left to right in fp32 like the reference."""
    mean = scatter_mean(p3, unq_inv, num)
    f_cluster = p3 - mean[unq_inv]
    f_center = p3[:, :2] - (pc[:, :2].to(p3.dtype) * vs[:2].unsqueeze(0) + vs[:2].unsqueeze(0) / 2 + pr[:2].unsqueeze(0))
    return torch.cat([points_rest, f_cluster, f_center], dim=-1)


class PillarVoxelNet(nn.Module):
    """mvf:39-86: returns (features (N, F + 5), coords (P, 3) int32 [b, y, x], unq_inv, grid [gy, gx])."""

    def __init__(self, voxel_size, pc_range):
        super().__init__()
        self.voxel_size = np.array(voxel_size)
        self.pc_range = np.array(pc_range)

    def forward(self, points):
        grid = grid_of(self.pc_range, self.voxel_size)
        pc, coords, unq_inv, vs, pr = _cells(points[:, 1:4], points[:, 0].long(), self.pc_range, self.voxel_size, grid)
        feats = _decorate(points[:, 1:], points[:, 1:4], pc, unq_inv, coords.shape[0], vs, pr)
        return feats, coords, unq_inv, grid[[1, 0]]


class CylinderNet(nn.Module):
    """mvf:88-141: the same grouping in (phi [deg], z, rho) coordinates."""

    def __init__(self, voxel_size, pc_range):
        super().__init__()
        self.voxel_size = np.array(voxel_size)
        self.pc_range = np.array(pc_range)

    def forward(self, points):
        x, y, z = points[:, 1:2], points[:, 2:3], points[:, 3:4]
        phi = torch.atan2(y, x) / np.pi * 180
        rho = torch.sqrt(x ** 2 + y ** 2)
        cyl = torch.cat((points[:, 0:1], phi, z, rho, points[:, 4:]), dim=-1)
        grid = grid_of(self.pc_range, self.voxel_size)
        pc, coords, unq_inv, vs, pr = _cells(cyl[:, 1:4], cyl[:, 0].long(), self.pc_range, self.voxel_size, grid)
        feats = _decorate(cyl[:, 1:], cyl[:, 1:4], pc, unq_inv, coords.shape[0], vs, pr)
        return feats, coords, unq_inv, grid[[1, 0]]


def bilinear_interpolate(image, coords):
    """mvf:208-246: image (B, C, H, W), coords (N, 3) = [b, x, y] in cell units -> (N, C); corners clamped to the map."""
    x, y = coords[:, 1], coords[:, 2]
    x0 = torch.floor(x).long()
    y0 = torch.floor(y).long()
    x1, y1 = x0 + 1, y0 + 1
    B = coords[:, 0].long()
    x0 = torch.clamp(x0, 0, image.shape[3] - 1)
    x1 = torch.clamp(x1, 0, image.shape[3] - 1)
    y0 = torch.clamp(y0, 0, image.shape[2] - 1)
    y1 = torch.clamp(y1, 0, image.shape[2] - 1)
    Ia, Ib, Ic, Id = image[B, :, y0, x0], image[B, :, y1, x0], image[B, :, y0, x1], image[B, :, y1, x1]
    wa = ((x1.float() - x) * (y1.float() - y)).unsqueeze(-1)
    wb = ((x1.float() - x) * (y - y0.float())).unsqueeze(-1)
    wc = ((x - x0.float()) * (y1.float() - y)).unsqueeze(-1)
    wd = ((x - x0.float()) * (y - y0.float())).unsqueeze(-1)
    return Ia * wa + Ib * wb + Ic * wc + Id * wd


class SingleView(nn.Module):
    """PFN layers + a sparse ResNet over one view's cells, sampled back at the points (mvf:143-206).  The sparse blocks are the
    masked-dense stand-ins of models.py (same keys: blocks.{i}.{j}.conv.weight ...)."""

    def __init__(self, in_channels, num_filters, layer_nums, ds_layer_strides, ds_num_filters, kernel_size, mode, voxel_size, pc_range, norm_cfg=None,
                 act_cfg=None):
        super().__init__()
        from .models import SparseBasicBlock, SparseConvBlock, _Seq

        self.mode = mode
        self.voxel_size = np.array(voxel_size[:2])
        self.bias = np.array(pc_range[:2])
        nf = [in_channels] + list(num_filters)
        self.pfn_layers = nn.ModuleList([PFNLayer(nf[i], nf[i + 1], norm_cfg=norm_cfg, last_layer=i >= len(nf) - 2) for i in range(len(nf) - 1)])
        in_filters = [nf[-1], *ds_num_filters[:-1]]
        self.blocks = nn.ModuleList([_Seq([SparseConvBlock(in_filters[i], ds_num_filters[i], kernel_size[i], ds_layer_strides[i], use_subm=False)]
                                          + [SparseBasicBlock(ds_num_filters[i], kernel_size[i]) for _ in range(n)]) for i, n in enumerate(layer_nums)])
        self.ds_rate = np.prod(np.array(ds_layer_strides))

    def forward(self, features, unq, unq_inv, grid_size, batch_size=None):
        pos = features[:, 0:2] if self.mode == "pillar" else features[:, 10:12]
        vs = torch.from_numpy(self.voxel_size).type_as(pos).to(pos.device)
        bias = torch.from_numpy(self.bias).type_as(pos).to(pos.device)
        pos = (pos - bias) / vs
        P = unq.shape[0]
        for pfn in self.pfn_layers:
            features = pfn(features, unq_inv, P)
        fv = ops.scatter_max(features, unq_inv, P)[0]
        if batch_size is None:
            batch_size = len(torch.unique(unq[:, 0]))                          # the reference's rule (mvf:190)
        H, W = int(grid_size[0]), int(grid_size[1])
        canvas = torch.zeros((batch_size, H, W, fv.shape[1]), dtype=fv.dtype, device=fv.device)
        mask = torch.zeros((batch_size, 1, H, W), dtype=fv.dtype, device=fv.device)
        u = unq.long()
        canvas[u[:, 0], u[:, 1], u[:, 2]] = fv
        mask[u[:, 0], 0, u[:, 1], u[:, 2]] = 1
        x = canvas.permute(0, 3, 1, 2)
        for blk in self.blocks:
            x, mask = blk(x, mask)
        pos = torch.cat((unq[unq_inv][:, 0:1].to(pos.dtype), pos / float(self.ds_rate)), dim=-1)
        return bilinear_interpolate(x, pos)

    bilinear_interpolate = staticmethod(bilinear_interpolate)


class MVFFeatureNet(nn.Module):
    """mvf:249-327."""

    def __init__(self, in_channels, voxel_size, pc_range, cylinder_size, cylinder_range, num_filters, layer_nums, ds_layer_strides, ds_num_filters,
                 kernel_size, out_channels):
        super().__init__()
        self.in_channels = in_channels
        self.voxel_size, self.pc_range = voxel_size, pc_range
        self.cylinder_range, self.cylinder_size = cylinder_range, cylinder_size
        self.voxelization = PillarVoxelNet(voxel_size, pc_range)
        self.cylinderlization = CylinderNet(cylinder_size, cylinder_range)
        c = (in_channels + 5) * 2
        self.pillarview = SingleView(c, num_filters, layer_nums, ds_layer_strides, ds_num_filters, kernel_size, "pillar", self.voxel_size, self.pc_range)
        self.cylinderview = SingleView(c, num_filters, layer_nums, ds_layer_strides, ds_num_filters, kernel_size, "cylinder", self.cylinder_size,
                                       self.cylinder_range)
        self.ds_rate = np.prod(np.array(ds_layer_strides))
        self.pointnet1 = PointNet(c, ds_num_filters[-1])
        self.pointnet2 = PointNet(ds_num_filters[-1] * 3, out_channels)

    def forward(self, points, batch_size=None):
        r = torch.tensor(self.pc_range, dtype=points.dtype, device=points.device)
        mask = ((points[:, 1] >= r[0]) & (points[:, 1] < r[3]) & (points[:, 2] >= r[1]) & (points[:, 2] < r[4]) & (points[:, 3] >= r[2])
                & (points[:, 3] < r[5]))
        points = points[mask]
        pf, pcoords, pinv, psize = self.voxelization(points)
        cf, ccoords, cinv, csize = self.cylinderlization(points)
        feat = torch.cat((pf, cf), dim=-1)
        pv = self.pillarview(feat, pcoords, pinv, psize, batch_size)
        cv = self.cylinderview(feat, ccoords, cinv, csize, batch_size)
        feat = torch.cat((self.pointnet1(feat), pv, cv), dim=-1)
        pillar = ops.scatter_max(self.pointnet2(feat), pinv, pcoords.shape[0])[0]
        if batch_size is None:
            batch_size = len(torch.unique(pcoords[:, 0]))
        ds = int(self.ds_rate)
        u = pcoords.long()
        H, W = int(psize[0]) // ds, int(psize[1]) // ds
        # SparseConvTensor(features, coords // ds, ...).dense() (mvf:322-327): several pillars fall into one coarse cell; spconv's dense()
        # scatters them in index order, the last one wins -- the same rule here (index_put with accumulate=False keeps the last write on
        # the CPU; on the GPU the winner among equal cells is unspecified, as it is in spconv's scatter kernel)
        out = torch.zeros((batch_size, H, W, pillar.shape[1]), dtype=pillar.dtype, device=pillar.device)
        out[u[:, 0], u[:, 1] // ds, u[:, 2] // ds] = pillar
        return out.permute(0, 3, 1, 2)
