"""Host-side mirror of the reference reader (det3d/models/readers/pillar_encoder.py) on top of libpnx_hip.so.

Same class names, constructor arguments, parameter names (``pfn_layers.{i}.linear.weight``,
``pfn_layers.{i}.norm.*`` -- so reference checkpoints load) and return values:

    PillarFeatureNet(num_input_features, num_filters, voxel_size, pc_range, norm_cfg=None)   # :129-136
    forward(points (N, 1+F) fp32 [b,x,y,z,..]) -> (feat_max (P,64), coords (P,3) int32 [b,y,x], grid_size [ny,nx])  # :174-182

* eval mode: one fused call (voxelize + PFN with BatchNorm folded + max) -- ``pnx_reader_forward``.
* train mode: the fused training passes of csrc/pfn_train.hip behind ONE autograd.Function (pfn_train.py): batch statistics,
  running-stat updates, argmax-routed backward, SyncBatchNorm semantics via enable_sync() -- no (N',64) tensor in memory.
  PNX_TRAIN_FUSED=0 (or num_filters other than [64,64]) takes the round-1 path: HIP voxelizer + torch Linear/BatchNorm1d + HIP
  scatter-max with autograd.
* ``forward_dense`` is the MI355X path the detector uses: it writes the dense channels-last BEV canvas
  directly (what SparseConvTensor(...).dense() would give, sparse_resnet.py:63-68) with no host sync.
"""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from ._lib import PNX_NCHW, PNX_NHWC, PnxError, make_geom


class PFNLayer(nn.Module):
    """Linear(no bias) + BatchNorm1d(eps=1e-3, momentum=0.01) + ReLU + per-pillar max (+ concat)  (pe:15-50)."""

    def __init__(self, in_channels, out_channels, norm_cfg=None, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        if not self.last_vfe:
            out_channels = out_channels // 2
        self.units = out_channels
        self.linear = nn.Linear(in_channels, out_channels, bias=False)
        self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)

    def forward(self, inputs, unq_inv, num_pillars=None):
        x = F.relu(self.norm(self.linear(inputs)))
        if num_pillars is None:
            num_pillars = int(unq_inv.max().item()) + 1
        feat_max = ops.scatter_max(x, unq_inv, num_pillars)[0]
        x_max = feat_max[unq_inv]
        if self.last_vfe:
            return x_max
        return torch.cat([x, x_max], dim=1)


class PillarNet(nn.Module):
    """Dynamic pillarisation (pe:53-125): returns (features (N',F+5), coords (P,3) int32 [b,y,x], unq_inv (N',), grid [ny,nx])."""

    def __init__(self, num_input_features, voxel_size, pc_range):
        super().__init__()
        self.voxel_size = np.array(voxel_size)
        self.pc_range = np.array(pc_range)
        self._geom = make_geom(self.pc_range, self.voxel_size)
        self._ws = ops.Workspace()

    @property
    def grid_size(self):
        return np.array([self._geom.gy, self._geom.gx], dtype=np.int64)

    def forward(self, points, batch_size=None):
        points = points.contiguous().float()
        n, stride = points.shape
        if batch_size is None:
            # the reference infers B from the data (sparse_resnet.py:62); one tiny sync, only on this API
            batch_size = int(points[:, 0].max().item()) + 1 if n > 0 else 1
        dev = points.device
        feats = torch.empty((n, stride + 4), dtype=torch.float32, device=dev)
        coords = torch.empty((max(n, 1), 3), dtype=torch.int32, device=dev)
        inv = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
        counts = torch.zeros((2,), dtype=torch.int32, device=dev)
        ops.voxelize(points, batch_size, self._geom, self._ws, features=feats, coords=coords, unq_inv=inv, counts=counts)
        P, m = (int(v) for v in counts.tolist())
        return feats[:m], coords[:P], inv[:m], self.grid_size


class PillarFeatureNet(nn.Module):
    def __init__(self, num_input_features, num_filters, voxel_size, pc_range, norm_cfg=None):
        super().__init__()
        assert len(num_filters) > 0
        self.num_point_features = int(num_input_features)
        num_input_features += 5
        num_filters = [num_input_features] + list(num_filters)
        layers = []
        for i in range(len(num_filters) - 1):
            layers.append(PFNLayer(num_filters[i], num_filters[i + 1], norm_cfg=norm_cfg, last_layer=(i == len(num_filters) - 2)))
        self.pfn_layers = nn.ModuleList(layers)
        self.feature_output_dim = num_filters[-1]
        self.voxel_size = np.array(voxel_size)
        self.pc_range = np.array(pc_range)
        self.voxelization = PillarNet(num_input_features, voxel_size, pc_range)
        self._geom = self.voxelization._geom
        self._ws = ops.Workspace()
        self._folded = None
        self._folded_key = None
        self.sync, self.sync_group = False, None  # SyncBatchNorm mode of the fused training path (enable_sync)

    # ------------------------------------------------------------------ helpers
    @property
    def grid_size(self):
        return self.voxelization.grid_size

    def _fused_supported(self):
        return (len(self.pfn_layers) == 2 and self.pfn_layers[0].units == 32 and self.pfn_layers[1].units == 64
                and 3 <= self.num_point_features <= 6
                and all(isinstance(l.norm, (nn.BatchNorm1d, nn.SyncBatchNorm)) for l in self.pfn_layers))

    @staticmethod
    def _unfused_training():
        import os

        return os.environ.get("PNX_TRAIN_FUSED", "1") == "0"  # the round-1 path: torch Linear/BatchNorm1d + HIP scatter-max

    def folded_params(self):
        """BN-folded parameter buffer, re-folded whenever a parameter/buffer changed (tensor version counters)."""
        l0, l1 = self.pfn_layers
        ts = [l0.linear.weight, l0.norm.weight, l0.norm.bias, l0.norm.running_mean, l0.norm.running_var,
              l1.linear.weight, l1.norm.weight, l1.norm.bias, l1.norm.running_mean, l1.norm.running_var]
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if self._folded is None or key != self._folded_key:
            d = [t.detach().contiguous().float() for t in ts]
            self._folded = ops.fold_bn(self.num_point_features, d[0], d[1:5], d[5], d[6:10], l0.norm.eps, self._folded)
            self._folded_key = key
        return self._folded

    def enable_sync(self, process_group=None):
        """SyncBatchNorm semantics for the fused training path: batch statistics (and the BatchNorm backward sums) are all-reduced
        over `process_group` by pfn_train.py; the norm modules stay BatchNorm1d (their parameters/buffers are what gets trained)."""
        self.sync, self.sync_group = True, process_group
        return self

    # ------------------------------------------------------------------ reference API
    def forward(self, points, batch_size=None):
        if self.training and self._fused_supported() and points.is_cuda and not self._unfused_training():
            from .pfn_train import fused_pfn_train

            points = points.contiguous().float()
            if batch_size is None:
                batch_size = int(points[:, 0].max().item()) + 1 if points.shape[0] > 0 else 1
            feat_max, coords = fused_pfn_train(self, points, batch_size)
            return feat_max, coords, self.grid_size
        if self.training or not self._fused_supported():
            return self._forward_unfused(points, batch_size)
        points = points.contiguous().float()
        n = points.shape[0]
        if batch_size is None:
            batch_size = int(points[:, 0].max().item()) + 1 if n > 0 else 1
        dev = points.device
        cap = max(min(n, batch_size * int(self._geom.gx) * int(self._geom.gy)), 1)
        feat_max = torch.empty((cap, 64), dtype=torch.float32, device=dev)
        coords = torch.empty((cap, 3), dtype=torch.int32, device=dev)
        counts = torch.zeros((2,), dtype=torch.int32, device=dev)
        with torch.no_grad():
            ops.reader_forward(points, batch_size, self._geom, self.folded_params(), self._ws, feat_max=feat_max, coords=coords,
                               counts=counts)
        P = int(counts[0].item())  # the reference's output shape is data dependent: one sync
        return feat_max[:P], coords[:P], self.grid_size

    def _forward_unfused(self, points, batch_size=None):
        if self.training and self.sync:
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.sync_group) > 1 and \
                    not all(isinstance(l.norm, nn.SyncBatchNorm) for l in self.pfn_layers):
                raise PnxError("PillarFeatureNet is in synchronised-BatchNorm mode but its unfused training path would use per-rank "
                               "BatchNorm1d statistics: convert the model with models.convert_sync_batchnorm (it turns the reader's norms "
                               "into SyncBatchNorm) or keep PNX_TRAIN_FUSED=1 with the points on the GPU")
        features, coords, unq_inv, grid_size = self.voxelization(points, batch_size)
        P = coords.shape[0]
        for pfn in self.pfn_layers:
            features = pfn(features, unq_inv, P)
        feat_max = ops.scatter_max(features, unq_inv, P)[0]  # pe:180 (idempotent re-max, kept for gradient parity)
        return feat_max, coords, grid_size

    # ------------------------------------------------------------------ MI355X dense path
    def forward_dense(self, points, batch_size, dtype=torch.bfloat16, channels_last=True, out=None, counts=None, occupancy=None):
        """points -> dense BEV canvas (B, 64, ny, nx).  Eval mode: single fused call, no host sync, every canvas
        byte written exactly once.  Train mode: unfused path + scatter (gradients flow to the PFN parameters)."""
        ny, nx = int(self._geom.gy), int(self._geom.gx)
        dev = points.device
        if self.training or not self._fused_supported():
            feat_max, coords, _ = self.forward(points, batch_size) if self.training else self._forward_unfused(points, batch_size)
            canvas = torch.zeros((batch_size, ny, nx, 64), dtype=feat_max.dtype, device=dev)
            c = coords.long()
            canvas[c[:, 0], c[:, 1], c[:, 2]] = feat_max
            if occupancy is not None:
                occupancy.zero_()
                occupancy[c[:, 0], c[:, 1], c[:, 2]] = 1
            return canvas.permute(0, 3, 1, 2).to(dtype)
        points = points.contiguous().float()
        if out is None:
            mf = torch.channels_last if channels_last else torch.contiguous_format
            out = torch.empty((batch_size, 64, ny, nx), dtype=dtype, device=dev, memory_format=mf)
        if out.is_contiguous(memory_format=torch.channels_last):
            layout = PNX_NHWC
        elif out.is_contiguous():
            layout = PNX_NCHW
        else:
            raise PnxError("canvas must be channels_last or contiguous")
        with torch.no_grad():
            ops.reader_forward(points, batch_size, self._geom, self.folded_params(), self._ws, canvas=out, canvas_layout=layout,
                               occupancy=occupancy, counts=counts)
        return out
