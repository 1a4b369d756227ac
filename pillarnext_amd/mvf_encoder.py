"""Host-side mirror of det3d/models/readers/mvf_encoder.py ("mvf:" below): the multi-view (pillar + cylinder) reader of the mvf18_aspp detectors
(configs/models/reader/mvf_encoder.yaml, configs/experiments/waymo_det_mvf18_aspp_iou_car.yaml), on the kernels of csrc/group.hip.

    MVFFeatureNet(in_channels, voxel_size, pc_range, cylinder_size, cylinder_range, num_filters, layer_nums, ds_layer_strides,
                  ds_num_filters, kernel_size, out_channels).forward(points) -> dense (B, out_channels, gy / ds, gx / ds) map    # mvf:257-327

Same class names, constructor arguments and state-dict keys as the reference.  What runs where:
  * the two point -> cell groupings, which CLAMP the cell index instead of dropping points (mvf:57-62, 111-116), the cylinder transform, the
    per-cell means and the 10-column decoration: ONE C call each (pnx_group_points, modes PILLAR_CLAMP / CYLINDER_CLAMP); MVFFeatureNet's range
    mask (mvf:290-297) is the call's prefilter and both views write their columns into one (N', 20) buffer -- the torch.cat of mvf:304 never runs;
  * the PFN layers of a view in eval mode: pnx_pfn_layer_eval (Linear + folded BatchNorm + ReLU + per-cell max; the concat [x, max[inv]] of a
    non-last layer is read in place by the next one); in training they are reader.PFNLayer on the HIP scatter-max (autograd);
  * sampling the view's map back at the points: pnx_bilinear_gather (eval) / its torch statement over flat indices (training: needs autograd);
  * the per-view sparse ResNets: in eval mode the masked HIP convolution kernels of the PillarNeXt backbone (csrc/conv3x3.hip; BatchNorm folded,
    bf16, the 48 / 96 / 192 channels zero-padded to the kernels' 64 / 128 / 256: _HipViewNet), in training the masked-dense blocks of models.py
    (spconv is absent from the image; like the backbone that part is unpinned); the two PointNets: torch Linear (rocBLAS) + BatchNorm + ReLU.
CUDA tensors only: there is no CPU path."""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from ._lib import PNX_GROUP_CYLINDER_CLAMP, PNX_GROUP_PILLAR_CLAMP
from .reader import PFNLayer
from .voxel_encoder import _device_points, batch_of, grid_of


class PointNet(nn.Module):
    """Linear(no bias) + BatchNorm1d(eps 1e-3, momentum 0.01) + ReLU per point (mvf:19-37)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.linear = nn.Linear(in_channels, out_channels, bias=False)
        self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)

    def forward(self, points):
        return F.relu(self.norm(self.linear(points)))


class _ClampView(nn.Module):
    """Shared body of PillarVoxelNet / CylinderNet: returns (features (N, F + 5), coords (P, 3) int32 [b, c1, c0], unq_inv, grid [g1, g0])."""

    mode = PNX_GROUP_PILLAR_CLAMP

    def __init__(self, voxel_size, pc_range):
        super().__init__()
        self.voxel_size = np.array(voxel_size)
        self.pc_range = np.array(pc_range)
        self._geoms = {}

    def geom(self, keep_range=None):
        key = None if keep_range is None else tuple(float(v) for v in keep_range)
        if key not in self._geoms:
            self._geoms[key] = ops.group_geom(self.pc_range, self.voxel_size, self.mode, keep_range)
        return self._geoms[key]

    def group(self, points, batch_size=None, keep_range=None, features_out=None):
        points = _device_points(points, type(self).__name__)
        return ops.group_points(points, batch_of(points, batch_size), self.geom(keep_range), want_features=True, features_out=features_out)

    def forward(self, points, batch_size=None):
        r = self.group(points, batch_size)
        return r["features"], r["coords"], r["unq_inv"], grid_of(self.pc_range, self.voxel_size)[[1, 0]]


class PillarVoxelNet(_ClampView):
    """mvf:39-86: cells over (x, y)."""


class CylinderNet(_ClampView):
    """mvf:88-141: the same grouping over (phi [deg], z); features [phi z rho f.. | cluster | centre]."""

    mode = PNX_GROUP_CYLINDER_CLAMP


def bilinear_interpolate(image, coords):
    """mvf:208-246 as autograd-capable torch ops over flat indices (the training path; eval runs pnx_bilinear_gather): image (B, C, H, W),
    coords (N, 3) = [b, x, y] in cell units -> (N, C).  Corner (kx, ky) = (floor(x) + kx, floor(y) + ky) clamped to the map; its weight is the product
    of the distances to the OTHER clamped corner along each axis, as the reference computes it after clamping."""
    B, C, H, W = image.shape
    flat = image.permute(0, 2, 3, 1).reshape(B * H * W, C)
    b = coords[:, 0].long()
    px, py = coords[:, 1], coords[:, 2]
    fx, fy = torch.floor(px).long(), torch.floor(py).long()
    xs = torch.stack([fx.clamp(0, W - 1), (fx + 1).clamp(0, W - 1)], 1)
    ys = torch.stack([fy.clamp(0, H - 1), (fy + 1).clamp(0, H - 1)], 1)
    wx = torch.stack([xs[:, 1].to(px.dtype) - px, px - xs[:, 0].to(px.dtype)], 1)
    wy = torch.stack([ys[:, 1].to(py.dtype) - py, py - ys[:, 0].to(py.dtype)], 1)
    out = None
    for kx, ky in ((0, 0), (0, 1), (1, 0), (1, 1)):
        term = flat[(b * H + ys[:, ky]) * W + xs[:, kx]] * (wx[:, kx] * wy[:, ky]).unsqueeze(-1).to(flat.dtype)
        out = term if out is None else out + term
    return out


def _fold_pfn(pfn):
    """(W' transposed (cin, cout), shift (cout)): BatchNorm1d(eval) folded into the Linear, fp32 (pnx_pfn_fold_bn's algebra)."""
    a = pfn.norm.weight.detach().float() / torch.sqrt(pfn.norm.running_var.float() + pfn.norm.eps)
    wt = (pfn.linear.weight.detach().float() * a[:, None]).t().contiguous()
    return wt, (pfn.norm.bias.detach().float() - pfn.norm.running_mean.float() * a).contiguous()


def _pad_to(c):
    """Channel count of the masked HIP convolution kernel that serves c channels (csrc/conv3x3.hip: 64, 128, 256)."""
    for k in (64, 128, 256):
        if c <= k:
            return k
    return 0


class _HipViewNet:
    """The sparse ResNet of one view (mvf:165-185: per stage a SparseConv2d block + n SubM basic blocks, 48 / 96 / 192 / 192 channels in the YAML) in eval
    mode on the masked HIP convolution kernels of the PillarNeXt backbone: BatchNorm folded into the weights, channels zero-padded to the kernels'
    64 / 128 / 256 (padded output channels have zero weights and zero bias, so they stay exactly 0 and cost MFMA lanes, not correctness), bf16
    channels_last, active-site masks as uint8 maps (SparseConv2d: mask_out = the 3x3 / stride pooled mask, SubM: unchanged).  Built from a
    SingleView's `blocks`; rebuilt when its parameters change.  Inference only."""

    def __init__(self, blocks, dtype=torch.bfloat16):
        from .models import SparseBasicBlock, SparseConvBlock, _fold_bn, _HipConv3x3

        self.dtype, self.stages = dtype, []
        self.version = _params_version(blocks)
        for seq in blocks:
            mods = []
            for m in seq:
                if isinstance(m, SparseConvBlock):
                    mods.append(("conv", self._conv(m.conv, m.norm, _fold_bn, _HipConv3x3), m.stride, m.subm))
                elif isinstance(m, SparseBasicBlock):
                    mods.append(("block", self._conv(m.block1.conv, m.block1.norm, _fold_bn, _HipConv3x3), self._conv(m.conv2, m.norm2, _fold_bn, _HipConv3x3)))
                else:
                    raise ops.PnxError(f"_HipViewNet: unexpected module {type(m).__name__}")
            self.stages.append(mods)
        self.cin = self.stages[0][0][1].cin
        self.cout_true = blocks[-1][-1].norm2.num_features if hasattr(blocks[-1][-1], "norm2") else blocks[-1][-1].norm.num_features

    def _conv(self, conv, norm, fold, HipConv):
        if tuple(conv.kernel_size) != (3, 3) or tuple(conv.padding) != (1, 1) or conv.bias is not None:
            raise ops.PnxError("_HipViewNet: 3x3 / pad 1 / bias-free convolutions only")
        w, b = fold(conv.weight, norm)
        co, ci = w.shape[:2]
        cop, cip = _pad_to(co), _pad_to(ci)
        stride = int(conv.stride[0])
        shapes = ops.CONV3X3_SHAPES_S1 if stride == 1 else ops.CONV3X3_SHAPES_S2
        if not cop or not cip or (cip, cop) not in shapes:
            raise ops.PnxError(f"_HipViewNet: no kernel for {ci} -> {co} channels at stride {stride}")
        wp = torch.zeros((cop, cip, 3, 3), dtype=torch.float32, device=w.device)
        wp[:co, :ci] = w
        bp = torch.zeros((cop,), dtype=torch.float32, device=w.device)
        bp[:co] = b
        return HipConv(wp, bp, stride, dtype=self.dtype).to(w.device)

    @staticmethod
    def supported(blocks):
        try:
            from .models import SparseBasicBlock, SparseConvBlock

            for seq in blocks:
                for m in seq:
                    if isinstance(m, SparseConvBlock):
                        cs = [(m.conv, m.stride)]
                    elif isinstance(m, SparseBasicBlock):
                        cs = [(m.block1.conv, 1), (m.conv2, 1)]
                    else:
                        return False
                    for c, st in cs:
                        shapes = ops.CONV3X3_SHAPES_S1 if st == 1 else ops.CONV3X3_SHAPES_S2
                        if tuple(c.kernel_size) != (3, 3) or c.bias is not None or (_pad_to(c.in_channels), _pad_to(c.out_channels)) not in shapes:
                            return False
            return True
        except Exception:
            return False

    def __call__(self, cell_features, coords, batch_size, H, W):
        """cell_features (P, C) fp32 at coords (P, 3) int32 [b, h, w] -> (map (B, Cpad, H', W') channels_last in self.dtype, true channel count)."""
        dev = cell_features.device
        canvas = torch.zeros((batch_size, H, W, self.cin), dtype=self.dtype, device=dev)
        occ = torch.zeros((batch_size, H, W), dtype=torch.uint8, device=dev)
        u = coords.long()
        canvas[u[:, 0], u[:, 1], u[:, 2], : cell_features.shape[1]] = cell_features.to(self.dtype)
        occ[u[:, 0], u[:, 1], u[:, 2]] = 1
        x, mask = canvas.permute(0, 3, 1, 2), occ                     # channels_last view of the NHWC canvas
        for mods in self.stages:
            for m in mods:
                if m[0] == "conv":
                    _, conv, stride, subm = m
                    if not subm:
                        mask = ops.mask_pool3(mask, stride)             # active set of a SparseConv2d's output
                    x = conv(x, mask)
                else:
                    _, c1, c2 = m
                    x = c2(c1(x, mask), mask, residual=x)
        return x


def _params_version(module):
    return tuple((p.data_ptr(), p._version) for p in list(module.parameters()) + list(module.buffers()))


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
        self.conv_dtype = None      # None: the blocks run as fp32 torch modules; torch.bfloat16 / float16: eval runs them on the HIP kernels (use_hip_convs)

    def use_hip_convs(self, dtype=torch.bfloat16):
        """Inference in 16-bit activations for the view's sparse ResNet (what FusedPillarNeXt is to the PillarNeXt backbone): eval-mode forwards under
        no_grad run `blocks` on the masked HIP convolution kernels (_HipViewNet).  dtype=None switches back to the fp32 modules."""
        if dtype is not None and dtype not in ops._HALF:
            raise ops.PnxError("use_hip_convs: torch.bfloat16, torch.float16 or None")
        self.conv_dtype = dtype
        return self

    def _pos_columns(self, features):
        return features[:, 0:2] if self.mode == "pillar" else features[:, 10:12]

    def cell_features(self, features, unq_inv, num_cells):
        """(P, C) = scatter_max over the cells of the PFN stack's output (mvf:187-188)."""
        if self.training or torch.is_grad_enabled() and any(p.requires_grad for p in self.pfn_layers.parameters()):
            for pfn in self.pfn_layers:
                features = pfn(features, unq_inv, num_cells)
            return ops.scatter_max(features, unq_inv, num_cells)[0]
        x, g = features, None
        for k, pfn in enumerate(self.pfn_layers):     # eval: one fused launch per layer; [x, max[inv]] is read in place by the next layer
            wt, shift = _fold_pfn(pfn)
            last = k == len(self.pfn_layers) - 1
            if pfn.last_vfe != last:
                raise ops.PnxError("SingleView: only the final PFN layer may be a last_layer")
            x, g = ops.pfn_layer_eval(x, g, unq_inv, wt, shift, num_cells, store=not last, want_max=True)
        return g

    def forward(self, features, unq, unq_inv, grid_size, batch_size=None):
        pos = self._pos_columns(features)
        P = unq.shape[0]
        fv = self.cell_features(features, unq_inv, P)
        if batch_size is None:
            batch_size = len(torch.unique(unq[:, 0]))                          # the reference's rule (mvf:190)
        H, W = int(grid_size[0]), int(grid_size[1])
        if self.conv_dtype is not None and not (self.training or torch.is_grad_enabled()):
            # eval, 16-bit inference requested (use_hip_convs): the view's sparse ResNet on the masked HIP convolution kernels
            net = self.__dict__.get("_hip_net")
            if net is None or net.version != _params_version(self.blocks) or net.dtype != self.conv_dtype:
                net = _HipViewNet(self.blocks, self.conv_dtype) if _HipViewNet.supported(self.blocks) else False
                self.__dict__["_hip_net"] = net
            if net is False:
                raise ops.PnxError("SingleView.use_hip_convs: this view's layer shapes have no masked HIP convolution kernel")
            if net:
                x = net(fv, unq, batch_size, H, W)
                out = ops.bilinear_gather(x, pos, self.bias, self.voxel_size, unq, unq_inv, int(self.ds_rate))
                return out[:, : net.cout_true].contiguous()
        canvas = torch.zeros((batch_size, H, W, fv.shape[1]), dtype=fv.dtype, device=fv.device)
        mask = torch.zeros((batch_size, 1, H, W), dtype=fv.dtype, device=fv.device)
        u = unq.long()
        canvas[u[:, 0], u[:, 1], u[:, 2]] = fv
        mask[u[:, 0], 0, u[:, 1], u[:, 2]] = 1
        x = canvas.permute(0, 3, 1, 2)                                         # channels_last view of the NHWC canvas
        for blk in self.blocks:
            x, mask = blk(x, mask)
        if not (self.training or torch.is_grad_enabled() and x.requires_grad) and x.dtype in ops._DT:
            return ops.bilinear_gather(x.contiguous(memory_format=torch.channels_last), pos, self.bias, self.voxel_size, unq, unq_inv, int(self.ds_rate))
        vs = torch.from_numpy(self.voxel_size).type_as(pos).to(pos.device)
        bias = torch.from_numpy(self.bias).type_as(pos).to(pos.device)
        cell = (pos - bias) / vs
        cell = torch.cat((unq[unq_inv][:, 0:1].to(cell.dtype), cell / float(self.ds_rate)), dim=-1)
        return bilinear_interpolate(x, cell)

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

    def use_hip_convs(self, dtype=torch.bfloat16):
        """Both views' sparse ResNets on the masked HIP convolution kernels in eval mode (SingleView.use_hip_convs)."""
        self.pillarview.use_hip_convs(dtype), self.cylinderview.use_hip_convs(dtype)
        return self

    def group_views(self, points, batch_size=None):
        """Both groupings of the range-masked points (mvf:290-304): (feat (N', 2 (F + 5)), pillar result, cylinder result)."""
        points = _device_points(points, "MVFFeatureNet")
        B = batch_of(points, batch_size)
        c = points.shape[1] + 4
        feat = torch.empty((max(points.shape[0], 1), 2 * c), dtype=torch.float32, device=points.device)
        pr = self.voxelization.group(points, B, keep_range=self.pc_range, features_out=(feat, 0))
        cr = self.cylinderlization.group(points, B, keep_range=self.pc_range, features_out=(feat, c))
        return feat[:pr["Nk"]], pr, cr, B

    def forward(self, points, batch_size=None):
        feat, pr, cr, B = self.group_views(points, batch_size)
        pcoords, pinv, ccoords, cinv = pr["coords"], pr["unq_inv"], cr["coords"], cr["unq_inv"]
        psize = grid_of(self.voxelization.pc_range, self.voxelization.voxel_size)[[1, 0]]
        csize = grid_of(self.cylinderlization.pc_range, self.cylinderlization.voxel_size)[[1, 0]]
        if batch_size is None:
            batch_size = len(torch.unique(pcoords[:, 0]))                      # the reference's rule (mvf:320)
        pv = self.pillarview(feat, pcoords, pinv, psize, batch_size)
        cv = self.cylinderview(feat, ccoords, cinv, csize, batch_size)
        x = torch.cat((self.pointnet1(feat), pv, cv), dim=-1)
        pillar = ops.scatter_max(self.pointnet2(x), pinv, pcoords.shape[0])[0]
        ds = int(self.ds_rate)
        u = pcoords.long()
        H, W = int(psize[0]) // ds, int(psize[1]) // ds
        # SparseConvTensor(features, coords // ds, ...).dense() (mvf:322-327): several pillars fall into one coarse cell; spconv's dense() scatters them
        # in index order -- which one stays is unspecified there and here (index_put without accumulate)
        out = torch.zeros((batch_size, H, W, pillar.shape[1]), dtype=pillar.dtype, device=pillar.device)
        out[u[:, 0], u[:, 1] // ds, u[:, 2] // ds] = pillar
        return out.permute(0, 3, 1, 2)
