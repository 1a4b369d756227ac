"""PillarNeXt-B around the HIP hot path: dense (masked) 2-D ResNet-18 backbone, ASPP neck, CenterHead and the
single-stage detector -- plain PyTorch-ROCm modules (nn.Conv2d -> MIOpen/MFMA), as BASELINE.json's north_star
prescribes.  Module/parameter names follow the reference so its checkpoints and YAML `_target_`s map 1:1:

    SparseResNet        det3d/models/backbones/sparse_resnet.py:10-68  (+ utils/sparse_conv.py:16-63)
    ASPPNeck            det3d/models/necks/aspp.py:8-40                (+ utils/conv.py)
    SepHead/CenterHead  det3d/models/heads/centerhead.py:12-136, predict :231-330, post_processing :332-384
    SingleStageDetector det3d/models/detectors/single_stage.py:6-59

The reference backbone is spconv; here it is dense convolution on the BEV canvas with the masked-dense rule
that reproduces spconv's active-site semantics exactly in eval mode:
    SubMConv2d            : out = conv(x) * mask_in
    SparseConv2d(3,s,p=1) : mask_out = maxpool(mask_in, 3, s, 1);  out = conv(x) * mask_out
    BatchNorm1d over sites: per-channel affine at active sites, zero elsewhere (all convs are bias-free).
"""
import copy
import os
import weakref

import numpy as np

import torch
import torch.nn.functional as F
from torch import nn

from . import ops


# ------------------------------------------------------------------------------------------------ backbone
def _spconv_weight_to_conv2d(w, conv):
    """spconv >= 2.2 stores (Cout, kH, kW, Cin); older (kH, kW, Cin, Cout).  nn.Conv2d wants (Cout, Cin, kH, kW)."""
    want = tuple(conv.weight.shape)
    co, ci, kh, kw = want
    layout = getattr(conv, "assume_layout", None)           # checkpoint.load_checkpoint sets it from the file's meta for the duration of the load
    if layout == "dense":
        if tuple(w.shape) != want:
            raise RuntimeError(f"dense-layout checkpoint holds {tuple(w.shape)} for Conv2d {want}")
        return w
    if layout == "spconv":
        if tuple(w.shape) != (co, kh, kw, ci):
            raise RuntimeError(f"spconv-layout checkpoint holds {tuple(w.shape)} for Conv2d {want}")
        return w.permute(0, 3, 1, 2).contiguous()
    if tuple(w.shape) == want:
        if kh > 1 and ci == kh == kw:
            # (Cout, Cin, kH, kW) and (Cout, kH, kW, Cin) have the same shape: a file that does not say which one it holds cannot be read safely
            raise RuntimeError(f"sparse-conv weight {want}: the shape does not tell nn.Conv2d's layout from spconv's; load the file with "
                               "checkpoint.load_checkpoint (files written by checkpoint.save_checkpoint record their layout) or set "
                               "conv.assume_layout = 'dense' / 'spconv'")
        return w
    if tuple(w.shape) == (co, kh, kw, ci):
        return w.permute(0, 3, 1, 2).contiguous()
    if tuple(w.shape) == (kh, kw, ci, co):
        return w.permute(3, 2, 0, 1).contiguous()
    raise RuntimeError(f"cannot map sparse-conv weight {tuple(w.shape)} onto Conv2d {want}")


class _SpConv2d(nn.Conv2d):
    """nn.Conv2d that accepts spconv-layout checkpoints."""

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        k = prefix + "weight"
        if k in state_dict:
            state_dict[k] = _spconv_weight_to_conv2d(state_dict[k], self)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class MaskedBatchNorm(nn.BatchNorm2d):
    """BatchNorm1d over ACTIVE sites of a dense map (keys/shapes identical to the reference's BatchNorm1d).
    eval: affine with running stats.  train: statistics over mask==1 positions only -- of the GLOBAL batch once
    enable_sync() was called (the masked counterpart of SyncBatchNorm, tools/train.py:56): per channel [sum, count] and then
    [sum of squared deviations] are all-reduced with the differentiable collective, so the backward all-reduces the matching
    gradient sums by itself.  torch's own SyncBatchNorm cannot stand in: it takes (input) only and would average over every
    dense cell instead of the active sites."""

    sync_group = None
    sync = False

    def enable_sync(self, process_group=None):
        self.sync, self.sync_group = True, process_group
        return self

    def _all_reduce(self, t):
        import torch.distributed as dist

        if not (self.sync and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.sync_group) > 1):
            return t
        from .dist_utils import all_reduce_sum_autograd

        return all_reduce_sum_autograd(t, self.sync_group)

    def forward(self, x, mask=None):
        if not self.training or mask is None:
            return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0, self.eps)
        m = mask.to(torch.float32)
        xf = x.float()
        s1 = self._all_reduce(torch.cat([(xf * m).sum(dim=(0, 2, 3)), m.sum().view(1)]))
        cnt = s1[-1].detach().clamp(min=1.0)
        mean = s1[:-1] / cnt
        var = self._all_reduce((((xf - mean.view(1, -1, 1, 1)) ** 2) * m).sum(dim=(0, 2, 3))) / cnt
        with torch.no_grad():
            mom = self.momentum
            self.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            self.running_var.mul_(1 - mom).add_(var * cnt / (cnt - 1).clamp(min=1.0), alpha=mom)
            self.num_batches_tracked += 1
        y = (xf - mean.view(1, -1, 1, 1)) * torch.rsqrt(var + self.eps).view(1, -1, 1, 1)
        return (y * self.weight.view(1, -1, 1, 1) + self.bias.view(1, -1, 1, 1)).to(x.dtype)


class _MaskedBNActFn(torch.autograd.Function):
    """y = relu(BN_active_sites(x) [+ residual]) * mask in ONE autograd node (train mode).  The module-by-module form keeps ~6 full-size
    fp32 tensors per layer alive for the backward (x.float(), the masked products, the normalised map, the ReLU output ...); this node
    keeps x (in its own dtype), the residual (the block's input, alive anyway) and two per-channel vectors, and recomputes the rest:
    the C2 x 4-frame training step went from 106 to 50.5 GiB (profiles/r03_train_step_c2_b4_fp32.log).  SyncBatchNorm mode all-reduces [sum x, count], [sum (x-mu)^2]
    forward and [sum g, sum g*xhat] backward over the ACTIVE sites of the global batch (dist_utils.all_reduce_sum)."""

    @staticmethod
    def _hip_ok(x, residual, weight=None, bias=None, norm=None):
        """The fused HIP kernels (csrc/masked_bn.hip) take channels_last bf16 / fp32 CUDA maps with 8..256 channels (fp32 parameters and buffers)."""
        C = x.shape[1]
        f32 = all(t is None or (t.dtype == torch.float32 and t.is_contiguous()) for t in (weight, bias))
        if norm is not None:
            f32 = f32 and norm.momentum is not None and norm.running_mean is not None and norm.running_mean.dtype == torch.float32 \
                and norm.running_var.dtype == torch.float32
        return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32) and C in (8, 16, 32, 64, 128, 256) and f32
                and x.is_contiguous(memory_format=torch.channels_last) and os.environ.get("PNX_MASKED_BN_HIP", "1") != "0"
                and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape
                                          and residual.is_contiguous(memory_format=torch.channels_last))))

    @staticmethod
    def _forward_hip(ctx, x, m, weight, bias, residual, norm, relu):
        """stats -> reduce -> [all-reduce] -> finalize -> apply: five launches (round 6: the per-channel arithmetic between the passes was ~20 host tensor
        statements per layer, 800 tiny launches per training step)."""
        from ._lib import check, lib, ptr, stream_ptr

        L = lib()
        B, C, H, W = x.shape
        n = B * H * W
        dt = ops._DT[x.dtype]
        nblk = int(L.pnx_masked_bn_blocks())
        dev = x.device
        part = torch.empty((nblk, 2 * C + 1), dtype=torch.float32, device=dev)
        mflat = None if m is None else m.reshape(-1)
        # statistics around the running mean (identical on every rank: DDP broadcasts the buffers): sum d, sum d^2 with d = x - centre
        rm, rv = norm.running_mean, norm.running_var
        check(L.pnx_masked_bn_stats(ptr(x), dt, ptr(mflat), n, C, ptr(rm), ptr(part), stream_ptr()), "pnx_masked_bn_stats")
        s = torch.empty((2 * C + 1,), dtype=torch.float64, device=dev)       # [sum d | sum d^2 | count]
        check(L.pnx_masked_bn_reduce(ptr(part), nblk, 2 * C + 1, ptr(s), stream_ptr()), "pnx_masked_bn_reduce")
        group = norm.sync_group if getattr(norm, "sync", False) else False
        if group is not False:
            from .dist_utils import all_reduce_sum

            all_reduce_sum(s, group)
        vec = torch.empty((4 * C + 1,), dtype=torch.float32, device=dev)
        mean32, invstd32, scale, shift, cnt32 = vec[:C], vec[C:2 * C], vec[2 * C:3 * C], vec[3 * C:4 * C], vec[4 * C:]
        check(L.pnx_masked_bn_finalize(ptr(s), C, ptr(rm), ptr(weight), ptr(bias), float(norm.eps), float(norm.momentum), ptr(rm), ptr(rv),
                                       ptr(norm.num_batches_tracked), ptr(mean32), ptr(invstd32), ptr(scale), ptr(shift), ptr(cnt32), stream_ptr()),
              "pnx_masked_bn_finalize")
        y = torch.empty_like(x)
        check(L.pnx_masked_bn_apply(ptr(x), ptr(residual), dt, ptr(mflat), n, C, ptr(scale), ptr(shift), 1 if relu else 0, ptr(y), stream_ptr()),
              "pnx_masked_bn_apply")
        ctx.save_for_backward(x, m, weight, bias, residual, mean32, invstd32, cnt32, scale, shift)
        ctx.group, ctx.relu, ctx.hip = group, relu, True
        return y

    @staticmethod
    def _backward_hip(ctx, gy):
        from ._lib import check, lib, ptr, stream_ptr

        L = lib()
        x, m, weight, bias, residual, mean, invstd, cnt, scale, shift = ctx.saved_tensors
        B, C, H, W = x.shape
        n = B * H * W
        dt = ops._DT[x.dtype]
        gy = gy.to(x.dtype)
        if not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.contiguous(memory_format=torch.channels_last)
        mflat = None if m is None else m.reshape(-1)
        nblk = int(L.pnx_masked_bn_blocks())
        part = torch.empty((nblk, 2 * C), dtype=torch.float32, device=x.device)
        relu = 1 if ctx.relu else 0
        check(L.pnx_masked_bn_bwd_stats(ptr(gy), ptr(x), ptr(residual), dt, ptr(mflat), n, C, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), relu, ptr(part),
                                        stream_ptr()), "pnx_masked_bn_bwd_stats")
        sg = torch.empty((2 * C,), dtype=torch.float64, device=x.device)
        check(L.pnx_masked_bn_reduce(ptr(part), nblk, 2 * C, ptr(sg), stream_ptr()), "pnx_masked_bn_reduce")
        sg_all = None                                                       # parameter gradients: local sums (DDP averages them)
        if ctx.group is not False:
            from .dist_utils import all_reduce_sum

            sg_all = sg.clone()
            all_reduce_sum(sg_all, ctx.group)
        vec = torch.empty((4 * C,), dtype=torch.float32, device=x.device)
        dgamma, dbeta, mg, mgx = vec[:C], vec[C:2 * C], vec[2 * C:3 * C], vec[3 * C:]
        check(L.pnx_masked_bn_bwd_finalize(ptr(sg), ptr(sg_all), C, ptr(cnt), ptr(dgamma), ptr(dbeta), ptr(mg), ptr(mgx), stream_ptr()), "pnx_masked_bn_bwd_finalize")
        dx = torch.empty_like(x)
        gres = torch.empty_like(residual) if residual is not None and ctx.needs_input_grad[4] else None
        check(L.pnx_masked_bn_bwd_apply(ptr(gy), ptr(x), ptr(residual), dt, ptr(mflat), n, C, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), relu, ptr(mg),
                                        ptr(mgx), ptr(dx), ptr(gres), stream_ptr()), "pnx_masked_bn_bwd_apply")
        return dx, None, dgamma.to(weight.dtype), dbeta.to(bias.dtype), gres, None, None

    @staticmethod
    def forward(ctx, x, mask, weight, bias, residual, norm, relu):
        if mask is None:   # every site active (dense_bn_act): the HIP kernels only -- no mask word in front of a site's loads
            return _MaskedBNActFn._forward_hip(ctx, x, None, weight, bias, residual, norm, relu)
        m = mask if mask.dtype == torch.float32 else mask.float()
        if _MaskedBNActFn._hip_ok(x, residual, weight, bias, norm):
            return _MaskedBNActFn._forward_hip(ctx, x, m.contiguous(), weight, bias, residual, norm, relu)
        ctx.hip = False
        xf = x.float()
        s1 = torch.cat([(xf * m).sum(dim=(0, 2, 3)), m.sum().view(1)])
        group = norm.sync_group if getattr(norm, "sync", False) else False
        if group is not False:
            from .dist_utils import all_reduce_sum

            all_reduce_sum(s1, group)
        cnt = s1[-1].clamp(min=1.0)
        mean = s1[:-1] / cnt
        xf = xf - mean.view(1, -1, 1, 1)
        ssd = (xf * xf * m).sum(dim=(0, 2, 3))
        if group is not False:
            all_reduce_sum(ssd, group)
        var = ssd / cnt
        invstd = torch.rsqrt(var + norm.eps)
        with torch.no_grad():
            mom = norm.momentum
            norm.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            norm.running_var.mul_(1 - mom).add_(var * cnt / (cnt - 1).clamp(min=1.0), alpha=mom)
            norm.num_batches_tracked += 1
        xf.mul_((invstd * weight.float()).view(1, -1, 1, 1)).add_(bias.float().view(1, -1, 1, 1))
        if residual is not None:
            xf.add_(residual)
        if relu:
            xf.clamp_(min=0)
        xf.mul_(m)
        ctx.save_for_backward(x, m, weight, bias, residual, mean, invstd, cnt)
        ctx.group, ctx.relu = group, relu
        return xf.to(x.dtype)

    @staticmethod
    def backward(ctx, gy):
        if ctx.hip:
            return _MaskedBNActFn._backward_hip(ctx, gy)
        x, m, weight, bias, residual, mean, invstd, cnt = ctx.saved_tensors
        xhat = (x.float() - mean.view(1, -1, 1, 1)).mul_(invstd.view(1, -1, 1, 1))
        g = gy.float() * m
        if ctx.relu:
            pre = xhat * weight.float().view(1, -1, 1, 1) + bias.float().view(1, -1, 1, 1)
            if residual is not None:
                pre = pre + residual
            g = g * (pre > 0)
            del pre
        gres = g.to(residual.dtype) if residual is not None and ctx.needs_input_grad[4] else None
        sg = torch.cat([g.sum(dim=(0, 2, 3)), (g * xhat).sum(dim=(0, 2, 3))])
        C = weight.numel()
        dbeta, dgamma = sg[:C].clone(), sg[C:].clone()          # parameter gradients: local sums (DDP averages them)
        if ctx.group is not False:
            from .dist_utils import all_reduce_sum

            all_reduce_sum(sg, ctx.group)
        mg, mgx = (sg[:C] / cnt).view(1, -1, 1, 1), (sg[C:] / cnt).view(1, -1, 1, 1)
        dx = (g - mg - xhat * mgx).mul_((weight.float() * invstd).view(1, -1, 1, 1)).mul_(m)
        return dx.to(x.dtype), None, dgamma.to(weight.dtype), dbeta.to(bias.dtype), gres, None, None


def _mask_u8(mask):
    """(B,1,H,W) float occupancy -> (B,H,W) uint8 for the HIP convolution kernels; cached on the tensor object (one conversion per stage)."""
    m = getattr(mask, "_pnx_u8", None)
    if m is None:
        m = (mask[:, 0] != 0).to(torch.uint8).contiguous()
        try:
            mask._pnx_u8 = m
        except Exception:
            pass
    return m


_ZERO_BIAS = {}


def _zero_bias(c, device):
    key = (c, str(device))
    if key not in _ZERO_BIAS:
        _ZERO_BIAS[key] = torch.zeros(c, dtype=torch.float32, device=device)
    return _ZERO_BIAS[key]


def _hip_dgrad_s2(weight):
    """(cin, cout) of the stride-2 data-gradient kernels; PNX_TRAIN_HIP_DGRAD_S2=0: MIOpen / CK."""
    return (weight.shape[1], weight.shape[0]) in {(64, 128), (128, 256), (256, 256)} and os.environ.get("PNX_TRAIN_HIP_DGRAD_S2", "1") != "0"


class _MaskedConv3x3Fn(torch.autograd.Function):
    """y = mask_out * conv3x3(x, W) on the product's masked-convolution kernels (csrc/conv3x3.hip) in TRAINING, bf16 autocast
    (sparse_conv.py:16-63: SubMConv2d / SparseConv2d compute only at the active sites, forward and backward).
      forward   pnx_conv3x3_bf16 (no bias, no ReLU), row segments without an active site are skipped
      dgrad     stride 1: the SAME kernel on the flipped, transposed weights with the INPUT's active set as its mask -- the upstream
                gradient is zero outside mask_out (the BatchNorm node's backward writes zeros there), and what reaches an inactive input
                site would be thrown away by that site's own mask; stride 2: pnx_conv3x3_dgrad_s2_bf16 (four parity planes; csrc/conv_dgrad_s2.h)
      wgrad     stride 1: pnx_conv3x3_wgrad_bf16 (csrc/conv_wgrad.hip) over the 16-pixel row pieces that hold an active output, fp32 accumulation,
                deterministic, stride 1 and 2 (PNX_TRAIN_HIPWGRAD=0: MIOpen's dense wrw on (x, g), exact because g is zero outside the active outputs)
    The fp32 training graph (the reference's precision) has its own node, _MaskedConv3x3F32Fn."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, mask_out, mask_in, stride, bias=None):
        # No cast_inputs (round 6): the fp32 master weight goes straight into the packing kernel (which rounds to bf16 exactly like the cast), the bias
        # stays fp32 for the kernel, and the weight gradient comes back in fp32 -- autocast's casts of both and the bf16 round trip of dW were ~240
        # tiny launches per step.
        x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        co = weight.shape[0]
        b = _zero_bias(co, x.device) if bias is None else bias.detach().float().contiguous()
        y = ops.conv3x3_masked(x, ops.conv3x3_pack_weights(weight), b, co, stride=stride, mask=mask_out, relu=False)
        ctx.save_for_backward(x, weight, mask_in, mask_out)
        ctx.stride = stride
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, weight, mask_in, mask_out = ctx.saved_tensors
        g = g.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = dw = None
        db = g.sum(dim=(0, 2, 3), dtype=torch.float32) if len(ctx.needs_input_grad) > 5 and ctx.needs_input_grad[5] else None
        if mask_out is None:   # a dense layer (dense_conv below): the weight-gradient kernel wants the output's active set
            mask_out = _ones_mask(g.shape[0], g.shape[2], g.shape[3], g.device)
        if ctx.stride == 1:
            if need_x:
                ci = weight.shape[1]
                # dgrad of a stride-1 'same' convolution = convolution with W^T flipped (packed in one launch)
                dx = ops.conv3x3_masked(g, ops.conv3x3_pack_weights(weight, transposed=True), _zero_bias(ci, g.device), ci, stride=1, mask=mask_in, relu=False)
            if need_w:
                if os.environ.get("PNX_TRAIN_HIPWGRAD", "1") != "0":
                    dw = ops.conv3x3_wgrad(x, g, mask_out).to(weight.dtype)
                else:
                    dw = torch.nn.grad.conv2d_weight(x, weight.shape, g, stride=1, padding=1).to(weight.dtype)
        else:
            s = ctx.stride
            hip_w = need_w and os.environ.get("PNX_TRAIN_HIPWGRAD", "1") != "0"
            if hip_w:
                dw = ops.conv3x3_wgrad(x, g, mask_out, stride=s).to(weight.dtype)
            if need_x and s == 2 and mask_in is not None and _hip_dgrad_s2(weight):   # round 6: the four parity planes on csrc/conv_dgrad_s2.h
                dx = ops.conv3x3_dgrad_s2(g, ops.conv3x3_pack_weights(weight, transposed=True), weight.shape[1], x.shape[2:], mask_in)
                need_x = False
            if need_x or (need_w and not hip_w):
                dx, dw2, _ = torch.ops.aten.convolution_backward(g, x, weight.to(g.dtype), None, (s, s), (1, 1), (1, 1), False, (0, 0), 1,
                                                                 (need_x, need_w and not hip_w, False))
                dw = dw if hip_w else dw2.to(weight.dtype)
        return dx, dw, None, None, None, db


def _split_pack(weight, transposed=False):
    """fp32 (Cout,Cin,3,3) -> the packed bf16 halves (W_hi, W_lo) pnx_conv3x3_x3 takes."""
    hi, lo = ops.split_f32(weight.detach().contiguous())
    return ops.conv3x3_pack_weights(hi, transposed=transposed), ops.conv3x3_pack_weights(lo, transposed=transposed)


_ONES_MASK = {}


def _ones_mask(b, h, w, device):
    key = (b, h, w, device)
    if key not in _ONES_MASK:
        _ONES_MASK[key] = torch.ones((b, h, w), dtype=torch.uint8, device=device)
    return _ONES_MASK[key]


class _MaskedConv3x3F32Fn(torch.autograd.Function):
    """_MaskedConv3x3Fn for the fp32 training graph (the reference's training precision: tools/train.py runs without autocast): every fp32 operand
    is split into two bf16 halves (16 mantissa bits together) and the three significant products run on the bf16 matrix cores with fp32 accumulation.
      forward   pnx_conv3x3_x3 on (x_hi, x_lo) x (W_hi, W_lo) [+ bias]: one launch, fp32 out
      dgrad     stride 1: the same kernel on the halves of g and of W^T flipped, masked by the INPUT's active set; stride 2: pnx_conv3x3_dgrad_s2_x3
      wgrad     pnx_conv3x3_wgrad_x3: x_hi g_hi + x_lo g_hi + x_hi g_lo in one pass, accumulated in fp32
    mask_out = mask_in = None: a dense layer (the neck's and the head's 3x3 convolutions, x3_conv below).  halves: (x_hi, x_lo) when the caller
    already split x (the six branches of a SepHead share their input).
    Relative error of every product ~ 4e-6 (the dropped low x low term and the halves' rounding, 2^-17 each), against MIOpen's fp32 kernels' ~ 2e-7:
    tests/test_gpu_masked_conv_train.py holds the node against an fp64 convolution.  PNX_TRAIN_F32_HIP=0 keeps the fp32 graph on MIOpen."""

    @staticmethod
    def forward(ctx, x, weight, bias, mask_out, mask_in, stride, halves):
        if halves is None:
            halves = ops.split_f32(x.contiguous(memory_format=torch.channels_last), mask_in)   # x is zero outside its active set (masked_bn_act, the scatter)
        xh, xl = halves
        wh, wl = _split_pack(weight)
        y = ops.conv3x3_x3(xh, xl, wh, wl, weight.shape[0], stride, mask_out, bias=None if bias is None else bias.detach().contiguous())
        ctx.save_for_backward(xh, xl, weight, mask_in, mask_out)
        ctx.stride = stride
        return y

    @staticmethod
    def backward(ctx, g):
        xh, xl, weight, mask_in, mask_out = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        g = g.contiguous(memory_format=torch.channels_last)
        s = ctx.stride
        dx = dw = db = None
        if need_w or need_x:
            gh, gl = ops.split_f32(g, mask_out)   # the upstream gradient is zero outside mask_out (the BatchNorm node's backward writes zeros there)
        if need_x:
            if s == 1:
                wth, wtl = _split_pack(weight, transposed=True)
                dx = ops.conv3x3_x3(gh, gl, wth, wtl, weight.shape[1], 1, mask_in)
            elif mask_in is not None and _hip_dgrad_s2(weight):
                wth, wtl = _split_pack(weight, transposed=True)
                dx = ops.conv3x3_dgrad_s2(gh, wth, weight.shape[1], xh.shape[2:], mask_in, g_lo=gl, wfrag_t_lo=wtl)
            else:
                dx = torch.nn.grad.conv2d_input(xh.shape, weight, g, stride=s, padding=1)
        if need_w:
            m = mask_out if mask_out is not None else _ones_mask(g.shape[0], g.shape[2], g.shape[3], g.device)
            dw = ops.conv3x3_wgrad_x3(xh, xl, gh, gl, m, stride=s)
        if need_b:
            db = g.sum(dim=(0, 2, 3))
        return dx, dw, db, None, None, None, None


class _SmallKConv3x3Fn(torch.autograd.Function):
    """nn.Conv2d(64, k <= 4, 3, padding 1, bias) of a SepHead branch in training (centerhead.py:31-41) on csrc/head_train.hip: forward and the weight / bias
    gradient at the map's HBM rate (fp32 FMAs, fp32 accumulation, fp32 master weights also under autocast); the data gradient on MIOpen."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        x = x.contiguous(memory_format=torch.channels_last)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.conv3x3_smallk(x, weight.detach().contiguous(), None if bias is None else bias.detach().contiguous())

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad
        dx = dw = db = None
        g = g.to(x.dtype)
        if need_x:
            dx = torch.nn.grad.conv2d_input(x.shape, weight.to(x.dtype), g, stride=1, padding=1)
        if need_w or (need_b and ctx.has_bias):
            dw, db = ops.conv3x3_smallk_wgrad(x, g, want_bias=ctx.has_bias)
        return dx, dw, db


def smallk_ok(conv, x):
    """Does the output convolution of a SepHead branch run on _SmallKConv3x3Fn?  (training, a CUDA fp32 graph or bf16 autocast; PNX_TRAIN_HEAD_HIP=0: MIOpen)"""
    if not (type(conv) is nn.Conv2d and conv.training and torch.is_grad_enabled() and x.is_cuda and x.dim() == 4 and conv.in_channels == 64
            and 1 <= conv.out_channels <= 4 and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and conv.weight.dtype == torch.float32
            and os.environ.get("PNX_TRAIN_HEAD_HIP", "1") != "0"):
        return False
    if torch.is_autocast_enabled():
        return torch.get_autocast_dtype("cuda") == torch.bfloat16 and x.dtype in (torch.bfloat16, torch.float32)
    return x.dtype == torch.float32


def smallk_conv(conv, x):
    if smallk_ok(conv, x):
        if torch.is_autocast_enabled() and x.dtype == torch.float32:
            x = x.to(torch.bfloat16)     # what autocast does to a convolution's input
        return _SmallKConv3x3Fn.apply(x, conv.weight, conv.bias)
    return conv(x)


_X3_DENSE = {(64, 64), (128, 128), (256, 256)}   # stride-1 shapes of pnx_conv3x3_x3


def x3_ok(x, weight, stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1, training=True):
    """Does a DENSE 3x3 convolution of the fp32 training graph (neck / head: det3d/models/utils/conv.py, centerhead.py:24-41) run on the three-product node?"""
    return (training and torch.is_grad_enabled() and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and not torch.is_autocast_enabled() and tuple(weight.shape[2:]) == (3, 3) and (weight.shape[1], weight.shape[0]) in _X3_DENSE
            and x.shape[1] == weight.shape[1] and tuple(stride) == (1, 1) and tuple(padding) == (1, 1) and tuple(dilation) == (1, 1) and groups == 1
            and os.environ.get("PNX_TRAIN_F32_HIP", "1") != "0")


def bf16_dense_ok(x, weight, stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1, training=True):
    """x3_ok's twin under bf16 autocast: the same dense layers on the bf16 masked kernels with every site active (_MaskedConv3x3Fn)."""
    return (training and torch.is_grad_enabled() and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and tuple(weight.shape[2:]) == (3, 3) and (weight.shape[1], weight.shape[0]) in _X3_DENSE
            and x.shape[1] == weight.shape[1] and tuple(stride) == (1, 1) and tuple(padding) == (1, 1) and tuple(dilation) == (1, 1) and groups == 1
            and os.environ.get("PNX_TRAIN_HIPCONV", "1") != "0" and os.environ.get("PNX_TRAIN_DENSE_HIP", "1") != "0")


def x3_conv(conv, x, halves=None):
    """conv(x) of a dense nn.Conv2d in training: the fp32 graph on _MaskedConv3x3F32Fn where x3_ok says so, bf16 autocast on _MaskedConv3x3Fn where
    bf16_dense_ok does, the module itself (MIOpen) otherwise."""
    if type(conv) is nn.Conv2d and conv.padding_mode == "zeros":
        if x3_ok(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups, conv.training):
            return _MaskedConv3x3F32Fn.apply(x, conv.weight, conv.bias, None, None, 1, halves)
        if bf16_dense_ok(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups, conv.training):
            return _MaskedConv3x3Fn.apply(x, conv.weight, None, None, 1, conv.bias)
    return conv(x)


# (Cin, Cout, stride) served by the LDS-staged kernels of csrc/conv3x3.hip (every 3x3 layer of the PillarNeXt-B backbone)
_HIP_TRAIN_CONVS = {(64, 64, 1), (128, 128, 1), (256, 256, 1), (64, 128, 2), (128, 256, 2), (256, 256, 2)}


def masked_conv(conv, x, mask_out, mask_in):
    """conv(x) of a backbone block; in training the 3x3 layers run on the product's masked kernels: _MaskedConv3x3Fn under bf16 autocast
    (PNX_TRAIN_HIPCONV=0 keeps every layer on MIOpen), _MaskedConv3x3F32Fn in the fp32 graph (PNX_TRAIN_F32_HIP=0: MIOpen)."""
    if (conv.training and torch.is_grad_enabled() and x.is_cuda and conv.kernel_size == (3, 3) and conv.bias is None
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and os.environ.get("PNX_TRAIN_HIPCONV", "1") != "0"
            and (conv.in_channels, conv.out_channels, conv.stride[0]) in _HIP_TRAIN_CONVS):
        return _MaskedConv3x3Fn.apply(x, conv.weight, _mask_u8(mask_out), _mask_u8(mask_in), conv.stride[0])
    if (conv.training and torch.is_grad_enabled() and x.is_cuda and conv.kernel_size == (3, 3) and conv.bias is None
            and not torch.is_autocast_enabled() and x.dtype == torch.float32 and conv.weight.dtype == torch.float32
            and os.environ.get("PNX_TRAIN_F32_HIP", "1") != "0"
            and (conv.in_channels, conv.out_channels, conv.stride[0]) in _HIP_TRAIN_CONVS):
        return _MaskedConv3x3F32Fn.apply(x, conv.weight, None, _mask_u8(mask_out), _mask_u8(mask_in), conv.stride[0], None)
    return conv(x)


def masked_bn_act(x, mask, norm, residual=None, relu=True):
    """relu(norm(x, mask) [+ residual]) * mask -- one fused autograd node in train mode, the plain modules otherwise."""
    if norm.training and mask is not None and torch.is_grad_enabled():
        return _MaskedBNActFn.apply(x, mask, norm.weight, norm.bias, residual, norm, relu)
    out = norm(x, mask)
    if residual is not None:
        out = out + residual
    return (F.relu(out) if relu else out) * mask


def dense_bn_act(norm, x, relu=True):
    """[relu](norm(x)) of a DENSE nn.BatchNorm2d in training (the head's and the neck's layers: det3d/models/utils/conv.py:21-34, centerhead.py:24-30): the
    masked node with every site active -- statistics, apply + ReLU, and the backward in four passes over the map on csrc/masked_bn.hip instead of MIOpen's
    three + three kernels and a separate ReLU forward and backward.  Anything else (eval, SyncBatchNorm, CPU, other shapes, PNX_TRAIN_DENSE_BN_HIP=0): the modules."""
    if (type(norm) is nn.BatchNorm2d and norm.training and torch.is_grad_enabled() and norm.affine and norm.track_running_stats and x.is_cuda and x.dim() == 4
            and os.environ.get("PNX_TRAIN_DENSE_BN_HIP", "1") != "0"):
        xc = x if x.is_contiguous(memory_format=torch.channels_last) else None
        if xc is not None and _MaskedBNActFn._hip_ok(xc, None, norm.weight, norm.bias, norm):
            return _MaskedBNActFn.apply(xc, None, norm.weight, norm.bias, None, norm, relu)
    y = norm(x)
    return F.relu(y) if relu else y


def convert_sync_batchnorm(module, process_group=None, cpu_ok=False):
    """tools/train.py:56 for this model, callable BEFORE or after .cuda() like torch's own converter (the reference converts first,
    train.py:56 then :59): MaskedBatchNorm layers switch to global active-site statistics in place; the reader's fused training
    passes exchange their statistics themselves (pfn_train.py) AND its BatchNorm1d modules become SyncBatchNorm, so the unfused
    fallback (PNX_TRAIN_FUSED=0, CPU points) synchronises too; every other BatchNorm becomes torch.nn.SyncBatchNorm.
    torch's SyncBatchNorm has no CPU forward: a module whose parameters sit on the CPU is converted all the same unless
    cpu_ok=True (gloo test runs), in which case it is left per-rank WITH a warning -- never silently."""
    import warnings

    from .reader import PillarFeatureNet

    def plain(parent, name, child):
        on_cpu = not next(child.parameters()).is_cuda
        if on_cpu and cpu_ok:
            warnings.warn(f"convert_sync_batchnorm: {type(child).__name__} '{name}' stays a per-rank BatchNorm (CPU run, cpu_ok=True)")
            return
        setattr(parent, name, nn.SyncBatchNorm.convert_sync_batchnorm(child, process_group))

    for name, child in list(module.named_children()):
        if isinstance(child, PillarFeatureNet):
            if child._fused_supported():
                child.enable_sync(process_group)
            for layer in child.pfn_layers:
                plain(layer, "norm", layer.norm)
        elif isinstance(child, MaskedBatchNorm):
            child.enable_sync(process_group)
        elif isinstance(child, nn.modules.batchnorm._BatchNorm):
            plain(module, name, child)
        else:
            convert_sync_batchnorm(child, process_group, cpu_ok)
    return module


class SparseConvBlock(nn.Module):
    """conv + BN + ReLU on active sites (sparse_conv.py:16-39).  stride 1 & use_subm -> submanifold."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, use_subm=True, bias=False):
        super().__init__()
        self.subm = stride == 1 and use_subm
        self.stride = stride
        self.kernel_size = kernel_size
        self.conv = _SpConv2d(in_channels, out_channels, kernel_size, stride=stride, padding=kernel_size // 2, bias=bias)
        self.norm = MaskedBatchNorm(out_channels, eps=1e-3, momentum=0.01)

    def forward(self, x, mask):
        mask_in = mask
        if not self.subm:
            mask = F.max_pool2d(mask, self.kernel_size, self.stride, self.kernel_size // 2)
        out = masked_bn_act(masked_conv(self.conv, x, mask, mask_in), mask, self.norm)
        return out, mask


class SparseBasicBlock(nn.Module):
    """Residual block of two submanifold convs (sparse_conv.py:42-63)."""

    def __init__(self, channels, kernel_size):
        super().__init__()
        self.block1 = SparseConvBlock(channels, channels, kernel_size, 1)
        self.conv2 = _SpConv2d(channels, channels, kernel_size, stride=1, padding=kernel_size // 2, bias=False)
        self.norm2 = MaskedBatchNorm(channels, eps=1e-3, momentum=0.01)

    def forward(self, x, mask):
        out, _ = self.block1(x, mask)
        out = masked_bn_act(masked_conv(self.conv2, out, mask, mask), mask, self.norm2, residual=x)
        return out, mask


class _Seq(nn.ModuleList):
    def forward(self, x, mask):
        for m in self:
            x, mask = m(x, mask)
        return x, mask


class SparseResNet(nn.Module):
    """Same constructor/keys as the reference's spconv SparseResNet; consumes the dense canvas."""

    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, num_input_features, kernel_size=(3, 3, 3, 3), out_channels=256):
        super().__init__()
        assert len(ds_layer_strides) == len(layer_nums) == len(ds_num_filters)
        in_filters = [num_input_features, *ds_num_filters[:-1]]
        blocks = []
        for i, n in enumerate(layer_nums):
            layers = [SparseConvBlock(in_filters[i], ds_num_filters[i], kernel_size[i], ds_layer_strides[i], use_subm=False)]
            layers += [SparseBasicBlock(ds_num_filters[i], kernel_size[i]) for _ in range(n)]
            blocks.append(_Seq(layers))
        self.blocks = nn.ModuleList(blocks)
        self.mapping = nn.ModuleList([_SpConv2d(ds_num_filters[-1], out_channels, 1, 1, bias=False),
                                      MaskedBatchNorm(out_channels, eps=1e-3, momentum=0.01)])
        self.num_input_features = num_input_features

    def forward_dense(self, canvas, mask):
        """canvas (B,C,ny,nx), mask (B,1,ny,nx) in canvas dtype -> (B,256,ny/8,nx/8) dense, zero at inactive sites."""
        x = canvas
        for blk in self.blocks:
            x, mask = blk(x, mask)
        return masked_bn_act(self.mapping[0](x), mask, self.mapping[1])

    def forward(self, pillar_features, coors, input_shape, batch_size=None):
        """Reference signature (sparse_resnet.py:61): builds the canvas from the sparse list first."""
        ny, nx = int(input_shape[0]), int(input_shape[1])
        if batch_size is None:
            batch_size = len(torch.unique(coors[:, 0]))  # the reference's own rule (:62)
        dev, dt = pillar_features.device, pillar_features.dtype
        canvas = torch.zeros((batch_size, ny, nx, pillar_features.shape[1]), dtype=dt, device=dev)
        mask = torch.zeros((batch_size, ny, nx, 1), dtype=dt, device=dev)
        c = coors.long()
        canvas[c[:, 0], c[:, 1], c[:, 2]] = pillar_features
        mask[c[:, 0], c[:, 1], c[:, 2]] = 1
        return self.forward_dense(canvas.permute(0, 3, 1, 2), mask.permute(0, 3, 1, 2))


# ------------------------------------------------------------------------------------------------ neck
class Conv(nn.Module):
    def __init__(self, inplanes, planes, kernel_size, stride, conv_layer=nn.Conv2d, bias=False, **kwargs):
        super().__init__()
        padding = kwargs.get("padding", kernel_size // 2)
        self.conv = conv_layer(inplanes, planes, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias)

    def forward(self, x):
        return x3_conv(self.conv, x)


class ConvBlock(nn.Module):
    def __init__(self, inplanes, planes, kernel_size, stride=1, conv_layer=nn.Conv2d, norm_layer=nn.BatchNorm2d, act_layer=nn.ReLU, **kwargs):
        super().__init__()
        padding = kwargs.get("padding", kernel_size // 2)
        self.conv = Conv(inplanes, planes, kernel_size=kernel_size, stride=stride, padding=padding, bias=False, conv_layer=conv_layer)
        self.norm = norm_layer(planes)
        self.act = act_layer()

    def forward(self, x):
        if type(self.act) is nn.ReLU:
            return dense_bn_act(self.norm, self.conv(x))
        return self.act(self.norm(self.conv(x)))


class BasicBlock(nn.Module):
    def __init__(self, inplanes, kernel_size=3):
        super().__init__()
        self.block1 = ConvBlock(inplanes, inplanes, kernel_size=kernel_size)
        self.block2 = ConvBlock(inplanes, inplanes, kernel_size=kernel_size)
        self.act = nn.ReLU()

    def forward(self, x):
        return self.act(self.block2(self.block1(x)) + x)


class ASPPNeck(nn.Module):
    """One shared 3x3 weight applied at dilation 1/6/12/18 + 1x1 + identity, concatenated (aspp.py:19-32)."""

    def __init__(self, in_channels):
        super().__init__()
        self.pre_conv = BasicBlock(in_channels)
        self.conv1x1 = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, bias=False, padding=0)
        self.weight = nn.Parameter(torch.randn(in_channels, in_channels, 3, 3))
        self.post_conv = ConvBlock(in_channels * 6, in_channels, kernel_size=1, stride=1)

    def _forward(self, x):
        x = self.pre_conv(x)
        w = self.weight.to(x.dtype)
        if x3_ok(x, w, training=self.training):
            d1 = _MaskedConv3x3F32Fn.apply(x, w, None, None, None, 1, None)
        elif bf16_dense_ok(x, w, training=self.training):
            d1 = _MaskedConv3x3Fn.apply(x, w, None, None, 1, None)
        else:
            d1 = F.conv2d(x, w, stride=1, bias=None, padding=1)
        outs = [x, self.conv1x1(x), d1] + [F.conv2d(x, w, stride=1, bias=None, padding=d, dilation=d) for d in (6, 12, 18)]
        return self.post_conv(torch.cat(outs, dim=1))

    def forward(self, x):
        if x.requires_grad:
            return torch.utils.checkpoint.checkpoint(self._forward, x, use_reentrant=False)
        return self._forward(x)


# ------------------------------------------------------------------------------------------------ head
class SepHead(nn.Module):
    def __init__(self, in_channels, heads, stride=1, head_conv=64, final_kernel=1, bn=True, init_bias=-2.19, **kwargs):
        super().__init__()
        if stride > 1:
            self.deblock = ConvBlock(in_channels, head_conv, kernel_size=int(stride), stride=int(stride), padding=0, conv_layer=nn.ConvTranspose2d)
            in_channels = head_conv
        else:
            self.deblock = nn.Identity()
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            fc = nn.Sequential()
            for _ in range(num_conv - 1):
                fc.append(nn.Conv2d(in_channels, head_conv, kernel_size=final_kernel, stride=1, padding=final_kernel // 2, bias=True))
                if bn:
                    fc.append(nn.BatchNorm2d(head_conv))
                fc.append(nn.ReLU())
            fc.append(nn.Conv2d(head_conv, classes, kernel_size=final_kernel, stride=1, padding=final_kernel // 2, bias=True))
            if "hm" in head:
                fc[-1].bias.data.fill_(init_bias)
            setattr(self, head, fc)

    def forward(self, x):
        x = self.deblock(x)
        if not (self.training and x.is_cuda and torch.is_grad_enabled()):
            return {head: getattr(self, head)(x) for head in self.heads}
        # training: the branches' first 3x3 convolutions on the product's kernels (x3_conv: the fp32 graph on the three-product node, their shared input
        # split into its bf16 halves once; bf16 autocast on the bf16 kernels), the output convolutions (64 -> k <= 4) on csrc/head_train.hip; anything
        # else is the module itself
        first = [getattr(self, head)[0] for head in self.heads]
        halves = None
        if any(type(c) is nn.Conv2d and len(getattr(self, h)) > 1 and x3_ok(x, c.weight, c.stride, c.padding, c.dilation, c.groups, c.training)
               for h, c in zip(self.heads, first)):
            halves = ops.split_f32(x.contiguous(memory_format=torch.channels_last))
        out = {}
        for head in self.heads:
            layers = list(getattr(self, head))
            h, i = x, 0
            while i < len(layers):
                layer = layers[i]
                if i == len(layers) - 1:
                    h = smallk_conv(layer, h)
                elif i == 0:
                    h = x3_conv(layer, h, halves)
                elif type(layer) is nn.BatchNorm2d and type(layers[i + 1]) is nn.ReLU:
                    h = dense_bn_act(layer, h)
                    i += 1
                else:
                    h = layer(h)
                i += 1
            out[head] = h
        return out


def _cfg_get(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


class CenterHead(nn.Module):
    def __init__(self, in_channels, tasks, weight, code_weights, common_heads, strides, init_bias=-2.19, share_conv_channel=64,
                 num_hm_conv=2, with_reg_iou=False, voxel_size=None, pc_range=None, out_size_factor=None,
                 rectifier=((0.0,), (0.0,), (0.0,))):
        super().__init__()
        self.num_classes = [len(t) for t in tasks]
        self.class_names = tasks
        self.code_weights = code_weights
        self.weight = weight
        self.in_channels = in_channels
        self.with_reg_iou = with_reg_iou
        self.with_iou = "iou" in common_heads
        self.voxel_size, self.pc_range, self.out_size_factor = voxel_size, pc_range, out_size_factor
        self.strides = strides
        self.rectifier = rectifier
        self.shared_conv = nn.Sequential(nn.Conv2d(in_channels, share_conv_channel, kernel_size=3, padding=1, bias=True),
                                         nn.BatchNorm2d(share_conv_channel), nn.ReLU(inplace=True))
        self.tasks = nn.ModuleList()
        for num_cls, stride in zip(self.num_classes, strides):
            heads = copy.deepcopy(dict(common_heads))
            heads.update(dict(hm=(num_cls, num_hm_conv)))
            self.tasks.append(SepHead(share_conv_channel, heads, stride=stride, bn=True, init_bias=init_bias, final_kernel=3))

    def forward(self, x, *kwargs):
        sc = self.shared_conv
        if self.training and len(sc) == 3 and type(sc[1]) is nn.BatchNorm2d and type(sc[2]) is nn.ReLU:
            x = dense_bn_act(sc[1], sc[0](x))
        else:
            x = sc(x)
        return [task(x) for task in self.tasks]

    # ---- training loss (centerhead.py:142-229)
    def loss(self, example, preds_dicts, **kwargs):
        from collections import OrderedDict

        from .losses import FastFocalLoss, IouLoss, IouRegLoss, RegLoss, gather_at  # noqa: F401

        crit, crit_reg = FastFocalLoss(), RegLoss()
        rets, total = [], None
        import os

        fused = preds_dicts[0]["hm"].is_cuda and os.environ.get("PNX_FUSED_LOSS", "1") != "0"
        for t, pd in enumerate(preds_dicts):
            if fused:
                # the four losses of the task in a handful of launches over the (B,500) lists (csrc/center_loss.hip), no host sync
                from .losses import fused_center_loss

                geom4 = (self.out_size_factor[t] * self.voxel_size[0], self.out_size_factor[t] * self.voxel_size[1], self.pc_range[0], self.pc_range[1])
                hm_loss, box_loss, iou_loss, iou_reg = fused_center_loss(pd, example["hm"][t], example["ind"][t], example["mask"][t], example["cat"][t],
                                                                        example["anno_box"][t], example["gt_boxes"][t], geom4, self.with_reg_iou)
                loc_loss = (box_loss * box_loss.new_tensor(self.code_weights)).sum()
                loss = hm_loss + self.weight * loc_loss
                # the log dict holds host tensors, as the reference's does (centerhead.py:165,212-222): ONE device->host copy per task
                host = torch.cat([hm_loss.detach().view(1), loc_loss.detach().view(1), box_loss.detach(), iou_loss.detach().view(1),
                                  iou_reg.detach().view(1), example["mask"][t].float().sum().view(1)]).cpu()
                ret = OrderedDict(task=self.class_names[t], loss=loss, hm_loss=host[0], loc_loss=host[1], loc_loss_elem=host[2:12], num_positive=host[14])
                if self.with_iou:
                    loss = loss + iou_loss
                    ret["iou_loss"] = host[12]
                if self.with_reg_iou:
                    loss = loss + self.weight * iou_reg
                    ret["iou_reg_loss"] = host[13]
                rets.append(ret)
                total = loss if total is None else total + loss
                continue
            hm = torch.clamp(torch.sigmoid(pd["hm"].float()), min=1e-4, max=1 - 1e-4)
            hm_loss = crit(hm, example["hm"][t], example["ind"][t], example["mask"][t], example["cat"][t])
            anno = torch.cat((pd["reg"], pd["height"], pd["dim"], pd["vel"], pd["rot"]), dim=1).float()
            box_loss = crit_reg(anno, example["mask"][t], example["ind"][t], example["anno_box"][t])
            loc_loss = (box_loss * box_loss.new_tensor(self.code_weights)).sum()
            loss = hm_loss + self.weight * loc_loss
            ret = OrderedDict(task=self.class_names[t], loss=loss, hm_loss=hm_loss.detach().cpu(), loc_loss=loc_loss.detach().cpu(),
                              loc_loss_elem=box_loss.detach().cpu(), num_positive=example["mask"][t].float().sum().cpu())
            if self.with_iou or self.with_reg_iou:
                B, _, H, W = pd["dim"].shape
                dim = torch.exp(torch.clamp(pd["dim"].float(), min=-5, max=5))
                rot = torch.atan2(pd["rot"][:, 0:1].float(), pd["rot"][:, 1:2].float())
                ys, xs = torch.meshgrid(torch.arange(H, device=dim.device), torch.arange(W, device=dim.device), indexing="ij")
                xs = (xs.view(1, 1, H, W).float() + pd["reg"][:, 0:1].float()) * self.out_size_factor[t] * self.voxel_size[0] + self.pc_range[0]
                ys = (ys.view(1, 1, H, W).float() + pd["reg"][:, 1:2].float()) * self.out_size_factor[t] * self.voxel_size[1] + self.pc_range[1]
                boxes = torch.cat([xs, ys, pd["height"].float(), dim, rot], dim=1)          # (B,7,H,W)
                if self.with_iou:
                    iou_loss = IouLoss()(pd["iou"].float(), example["mask"][t], example["ind"][t], boxes.detach(), example["gt_boxes"][t])
                    loss = loss + iou_loss
                    ret["iou_loss"] = iou_loss.detach().cpu()
                if self.with_reg_iou:
                    iou_reg = IouRegLoss()(boxes, example["mask"][t], example["ind"][t], example["gt_boxes"][t])
                    loss = loss + self.weight * iou_reg
                    ret["iou_reg_loss"] = iou_reg.detach().cpu()
                # ret['loss'] keeps the pre-IoU value, as the reference's log dict does (centerhead.py:165,212-222)
            rets.append(ret)
            total = loss if total is None else total + loss
        return total, rets

    # ---- inference: decode + per-class rotated NMS (centerhead.py:231-384)
    @torch.no_grad()
    def predict(self, example, preds_dicts, test_cfg):
        nms_cfg = _cfg_get(test_cfg, "nms")
        pre_max = int(_cfg_get(nms_cfg, "nms_pre_max_size"))
        post_max = int(_cfg_get(nms_cfg, "nms_post_max_size"))
        thr_cfg = _cfg_get(nms_cfg, "nms_iou_threshold")
        score_thr = float(_cfg_get(test_cfg, "score_threshold"))
        osf = _cfg_get(test_cfg, "out_size_factor")
        vs = _cfg_get(test_cfg, "voxel_size")
        pcr = _cfg_get(test_cfg, "pc_range")
        lim = _cfg_get(test_cfg, "post_center_limit_range")
        dev = preds_dicts[0]["hm"].device
        lim_t = torch.tensor(list(lim), dtype=torch.float32, device=dev) if len(lim) > 0 else None
        B = preds_dicts[0]["hm"].shape[0]
        tokens = example["token"] if ("token" in example and len(example["token"]) > 0) else [None] * B

        per_task = []
        flag = 0
        for t, pd in enumerate(preds_dicts):
            hm = torch.sigmoid(pd["hm"].float()).permute(0, 2, 3, 1)           # (B,H,W,ncls)
            _, H, W, ncls = hm.shape
            dim = torch.exp(pd["dim"].float()).permute(0, 2, 3, 1).reshape(B, H * W, 3)
            rot = pd["rot"].float()
            rot = torch.atan2(rot[:, 0], rot[:, 1]).reshape(B, H * W, 1)
            reg = pd["reg"].float().permute(0, 2, 3, 1).reshape(B, H * W, 2)
            hei = pd["height"].float().permute(0, 2, 3, 1).reshape(B, H * W, 1)
            vel = pd["vel"].float().permute(0, 2, 3, 1).reshape(B, H * W, 2)
            if "iou" in pd:
                iou = ((pd["iou"].float().squeeze(1) + 1) * 0.5).reshape(B, H * W)
            else:
                iou = torch.ones((B, H * W), dtype=torch.float32, device=dev)
            ys, xs = torch.meshgrid(torch.arange(0, H, device=dev), torch.arange(0, W, device=dev), indexing="ij")
            xs = xs.reshape(1, H * W, 1).float() + reg[:, :, 0:1]
            ys = ys.reshape(1, H * W, 1).float() + reg[:, :, 1:2]
            xs = xs * osf[t] * vs[0] + pcr[0]
            ys = ys * osf[t] * vs[1] + pcr[1]
            boxes = torch.cat([xs, ys, hei, dim, vel, rot], dim=2)                 # (B,HW,9)
            scores, labels = torch.max(hm.reshape(B, H * W, ncls), dim=-1)
            mask = scores > score_thr
            if lim_t is not None:
                mask &= (boxes[..., :3] >= lim_t[:3]).all(-1) & (boxes[..., :3] <= lim_t[3:]).all(-1)
            rect = torch.tensor(list(self.rectifier[t]), dtype=torch.float32, device=dev)[labels]
            scores = torch.pow(scores, 1 - rect) * torch.pow(torch.clamp(iou, min=0.0, max=1.0), rect)
            thr = torch.tensor([float(v) for v in thr_cfg[t]], dtype=torch.float32, device=dev)
            per_task.append(_task_nms(boxes, scores, labels, mask, ncls, thr, pre_max, post_max, flag))
            flag += ncls

        ret_list = []
        for i in range(B):
            ret = {"box3d_lidar": torch.cat([pt[i][0] for pt in per_task]), "scores": torch.cat([pt[i][1] for pt in per_task]),
                   "label_preds": torch.cat([pt[i][2] for pt in per_task]), "token": tokens[i]}
            ret_list.append(ret)
        return ret_list


def _task_nms(boxes, scores, labels, mask, ncls, thr, pre_max, post_max, label_offset):
    """All (sample, class) segments of one task through ONE sort and ONE batched NMS launch.
    Equivalent to the per-sample / per-class loop of CenterHead.post_processing (centerhead.py:337-374)."""
    B, HW, _ = boxes.shape
    dev = boxes.device
    S = B * ncls
    seg = torch.arange(B, device=dev).view(B, 1) * ncls + labels                          # (B,HW) segment id
    seg = torch.where(mask, seg, torch.full_like(seg, S))                                 # dropped cells -> bin S
    # sort by (segment, score desc): scores are in [0,1], so key = seg*4 - score is monotone
    key = seg.reshape(-1).double() * 4.0 - scores.reshape(-1).double()
    order = torch.argsort(key, stable=True)
    counts = torch.bincount(seg.reshape(-1), minlength=S + 1)[:S]
    off = torch.zeros(S + 1, dtype=torch.int32, device=dev)
    off[1:] = torch.cumsum(counts, 0).to(torch.int32)
    total = int(off[-1].item())                                                           # the one host sync of the task
    order = order[:total]
    flat_boxes = boxes.reshape(B * HW, 9)[order]
    nms_boxes = flat_boxes[:, [0, 1, 2, 3, 4, 5, 8]].contiguous()
    seg_len = torch.clamp(counts, max=pre_max).to(torch.int32)
    keep, cnt = ops.nms_batched(nms_boxes, off, thr.repeat(B), min(pre_max, max(total, 1)), post_max=post_max, seg_len=seg_len)
    cnt_l = cnt[:S].tolist()
    off_l = off.tolist()
    flat_scores = scores.reshape(-1)[order]
    out = []
    for b in range(B):
        bb, ss, ll = [], [], []
        for c in range(ncls):
            s = b * ncls + c
            k = keep[off_l[s]: off_l[s] + cnt_l[s]].long() + off_l[s]
            bb.append(flat_boxes[k])
            ss.append(flat_scores[k])
            ll.append(torch.full((cnt_l[s],), c + label_offset, dtype=torch.int64, device=dev))
        out.append((torch.cat(bb), torch.cat(ss), torch.cat(ll)))
    return out


# ------------------------------------------------------------------------------------------------ detector
class SingleStageDetector(nn.Module):
    """reader -> backbone -> neck -> head (single_stage.py).  On MI355X the reader writes the dense channels-last
    canvas + occupancy directly (fused path) whenever reader and backbone are ours; otherwise the reference's
    tuple hand-off `backbone(*reader(points))` is used."""

    def __init__(self, reader, backbone=None, neck=None, head=None, post_processing=None, compute_dtype=torch.bfloat16, **kwargs):
        super().__init__()
        self.reader, self.backbone, self.neck, self.head = reader, backbone, neck, head
        self.post_processing = post_processing
        self.compute_dtype = compute_dtype

    def extract_feat(self, points, batch_size=None):
        if hasattr(self.reader, "forward_dense") and hasattr(self.backbone, "forward_dense"):
            if batch_size is None:
                batch_size = int(points[:, 0].max().item()) + 1
            # eval: the canvas takes the dtype of the dense part (fp32 modules as built / loaded from a reference checkpoint, bf16
            # after .to(torch.bfloat16)); compute_dtype only matters to FusedPillarNeXt, which casts its own folded weights
            dt = next(self.backbone.parameters()).dtype if not self.training else torch.float32
            ny, nx = (int(v) for v in self.reader.grid_size)
            occ = torch.empty((batch_size, ny, nx), dtype=torch.uint8, device=points.device)
            canvas = self.reader.forward_dense(points, batch_size, dtype=dt, occupancy=occ)
            x = self.backbone.forward_dense(canvas, occ.unsqueeze(1).to(dt))
        else:
            x = self.reader(points)
            if self.backbone is not None:
                x = self.backbone(*x)
        if self.neck is not None:
            x = self.neck(x)
        return x

    def _forward(self, example):
        x = self.extract_feat(example["points"], example.get("batch_size"))
        return self.head(x)

    def forward(self, example):
        return self.training_step(example) if self.training else self.validation_step(example)

    def training_step(self, example):
        preds = self._forward(example)
        return self.head.loss(example, preds)

    @torch.no_grad()
    def validation_step(self, example):
        preds = self._forward(example)
        outputs = self.head.predict(example, preds, self.post_processing)
        detections = {}
        for output in outputs:
            token = output["token"]
            for k, v in output.items():
                if k != "token":
                    output[k] = v.to(torch.device("cpu"))
            detections[token] = output
        return detections


NUSC_TASKS = [["car"], ["truck", "construction_vehicle"], ["bus", "trailer"], ["barrier"], ["motorcycle", "bicycle"], ["pedestrian", "traffic_cone"]]


def build_pillarnext_b(pc_range, voxel_size, tasks=None, num_point_features=5, ds_layer_strides=(1, 2, 2, 2), with_iou_head=False,
                       rectifier=None, post_processing=None):
    """PillarNeXt-B as configured by configs/experiments/nusc_det_pp18_aspp_iou_sp.yaml (geometry is a parameter)."""
    from .reader import PillarFeatureNet

    tasks = tasks or NUSC_TASKS
    common = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)}
    if with_iou_head:
        common["iou"] = (1, 2)
    reader = PillarFeatureNet(num_point_features, [64, 64], list(voxel_size), list(pc_range))
    backbone = SparseResNet([2, 2, 2, 2], list(ds_layer_strides), [64, 128, 256, 256], 64)
    neck = ASPPNeck(256)
    rect = rectifier or [[0.5] * len(t) for t in tasks]
    head = CenterHead(256, tasks, 0.25, [1.0] * 6 + [0.2, 0.2, 1.0, 1.0], common, [2] * len(tasks), with_reg_iou=True,
                      voxel_size=list(voxel_size), pc_range=list(pc_range), out_size_factor=[4] * len(tasks), rectifier=rect)
    if post_processing is None:
        post_processing = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.1,
                               nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=[[0.2] * len(t) for t in tasks]),
                               out_size_factor=[4] * len(tasks), voxel_size=list(voxel_size), pc_range=list(pc_range))
    return SingleStageDetector(reader, backbone, neck, head, post_processing)


# ------------------------------------------------------------------------------------------------ fused inference graph
def _fold_bn(weight, bn, conv_bias=None, transposed=False):
    """Fold an eval-mode BatchNorm into the preceding conv: returns (weight', bias') in fp32."""
    w = weight.detach().float()
    a = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    shape = (1, -1, 1, 1) if transposed else (-1, 1, 1, 1)
    b0 = conv_bias.detach().float() if conv_bias is not None else torch.zeros_like(a)
    return w * a.view(shape), (b0 - bn.running_mean.detach().float()) * a + bn.bias.detach().float()


def fold_aspp(nk):
    """ASPPNeck (eval) after pre_conv as ONE sum of four dilated convolutions of x: returns ([w_d (O,C,3,3) for d in 1,6,12,18], shift (O,))
    with  relu(sum_d conv_d(x, w_d) + shift) == post_conv(cat(x, conv1x1(x), conv_d(x, W)...))  (aspp.py:19-32).  The identity and
    1x1 branches are 1x1 convolutions of x, i.e. centre taps: they are added to the centre tap of the dilation-1 kernel."""
    wp, bp = _fold_bn(nk.post_conv.conv.conv.weight, nk.post_conv.norm)
    C = nk.weight.shape[0]
    P = wp.detach().float().reshape(wp.shape[0], 6, C)                       # (out, branch, mid): column blocks of the post weight
    w1 = nk.conv1x1.weight.detach().float().reshape(C, C)
    ws = nk.weight.detach().float()
    wds = [torch.einsum("om,mikl->oikl", P[:, 2 + k], ws) for k in range(4)]
    wds[0][:, :, 1, 1] += P[:, 0] + P[:, 1] @ w1
    return wds, bp.float()


class _FusedConv(nn.Module):
    """conv (BN folded, no bias inside MIOpen) + ONE HIP epilogue pass: [relu](y + b [+ res]) * mask."""

    def __init__(self, weight, bias, stride=1, padding=0, dilation=1, relu=True, transposed=False, dtype=torch.bfloat16):
        super().__init__()
        self.register_buffer("weight", weight.to(dtype).contiguous(memory_format=torch.channels_last) if weight.dim() == 4 else weight.to(dtype))
        self.register_buffer("bias", bias.float().contiguous())
        self.stride, self.padding, self.dilation, self.relu, self.transposed = stride, padding, dilation, relu, transposed

    def forward(self, x, mask=None, residual=None, post_residual=None):
        """post_residual: added AFTER this layer's ReLU (relu mode 2 of the epilogue) instead of before it."""
        if self.transposed:
            y = F.conv_transpose2d(x, self.weight, None, self.stride, self.padding)
        else:
            y = F.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        if post_residual is not None:
            return ops.bias_act_mask_(y, self.bias, mask, post_residual, 2 if self.relu else 0)
        return ops.bias_act_mask_(y, self.bias, mask, residual, self.relu)


class _HipConv3x3(nn.Module):
    """3x3 masked conv + folded BN + [residual] + ReLU + mask in ONE HIP kernel (csrc/conv3x3.hip) -- no MIOpen, no epilogue pass."""

    def __init__(self, weight, bias, stride=1, dtype=torch.bfloat16):
        super().__init__()
        self.cout, self.cin, self.stride = weight.shape[0], weight.shape[1], int(stride)
        self.register_buffer("wfrag", ops.conv3x3_pack_weights(weight, dtype=dtype))
        self.register_buffer("bias", bias.float().contiguous())
    def forward(self, x, mask=None, residual=None, out=None, tiles=None):
        return ops.conv3x3_masked(x, self.wfrag, self.bias, self.cout, self.stride, mask, residual, True, out=out,
                                  tiles=tiles if self.stride == 1 else None)


class _HipDeconv2x2(nn.Module):
    """ConvTranspose2d(k=2, s=2) + folded BN + ReLU in ONE HIP kernel (csrc/conv3x3.hip::k_deconv2x2_64): the SepHead deblock."""

    def __init__(self, weight, bias, relu=True, dtype=torch.bfloat16):
        super().__init__()
        self.cout, self.relu = weight.shape[1], relu
        self.register_buffer("wfrag", ops.deconv2x2_pack_weights(weight, dtype=dtype))
        self.register_buffer("bias", bias.float().contiguous())

    def forward(self, x, mask=None, residual=None):
        return ops.deconv2x2(x, self.wfrag, self.bias, self.cout, self.relu)


class _HipSepHeadOut(nn.Module):
    """Last 3x3 conv of all SepHead branches of a task as ONE HIP kernel over the block-diagonal weight (csrc/conv3x3.hip::k_sephead_out)."""

    def __init__(self, weight, bias, dtype=torch.bfloat16):
        super().__init__()
        self.cout = weight.shape[0]
        self.register_buffer("wfrag", ops.sephead_pack_weights(weight, dtype=dtype))
        self.register_buffer("bias", bias.float().contiguous())

    def forward(self, x, mask=None, residual=None):
        return ops.sephead_out(x, self.wfrag, self.bias)


def _backbone_conv(weight, bias, stride, padding, dtype, hip_conv):
    co, ci, kh, kw = weight.shape
    shapes = ops.CONV3X3_SHAPES_S1 if stride == 1 else ops.CONV3X3_SHAPES_S2
    if hip_conv and dtype in ops._HALF and (kh, kw) == (3, 3) and padding == 1 and (ci, co) in shapes and stride in (1, 2):
        return _HipConv3x3(weight, bias, stride, dtype=dtype)
    return _FusedConv(weight, bias, stride, padding, dtype=dtype)


class LazyTask:
    """What the lazy head hands the decoder for one task: the dense [iou] hm map and the deblocked features the regression branches
    are evaluated on at the selected candidates."""

    def __init__(self, dense, up):
        self.dense, self.up = dense, up


class _DeferredDecode:
    """The decoder launch of one frame batch, held back until the next batch's reader has been enqueued (FusedPillarNeXt.forward_async with
    decode_on_side_stream) or until somebody asks for the result.  Behaves like the decode.PendingDetections it turns into."""

    def __init__(self, model, packed, tokens):
        self.model, self.packed, self.tokens = model, packed, tokens
        self.head_done = torch.cuda.Event()
        self.head_done.record()                       # the head of this batch, on the caller's stream
        self.launched, self.pend = False, None

    def launch(self, behind_reader=False):
        if self.launched:
            return
        self.launched = True
        m = self.model
        main = torch.cuda.current_stream()
        side = m.__dict__.get("_decode_stream")
        if side is None or side.device != main.device:
            side = m.__dict__["_decode_stream"] = torch.cuda.Stream(device=main.device)
        side.wait_event(self.head_done)
        if behind_reader:                             # called right after the next batch's reader was enqueued on `main`
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
        with torch.cuda.stream(side):
            self.pend = m.launch_decode(self.packed, self.tokens)
        for p in self.packed:   # allocated on the main stream, consumed on the side stream: the caching allocator must not hand them out before that work ran
            p.dense.record_stream(side), p.up.record_stream(side)
        self.packed = None
        ref = m.__dict__.get("_deferred")
        if ref is not None and ref() is self:
            m.__dict__["_deferred"] = None

    def result(self):
        self.launch()
        return self.pend.result()

    def __getattr__(self, name):                      # flag_h, event, done, ... of the PendingDetections
        if name in ("pend", "launched", "packed", "model", "tokens", "head_done"):
            raise AttributeError(name)
        self.launch()
        return getattr(self.pend, name)


class FusedPillarNeXt(nn.Module):
    """Inference-only re-expression of SingleStageDetector (eval BN folded, epilogues fused, the 6-7 SepHead branches of a
    task merged into two convolutions).  Mathematically the same network; weights come from the trained modules.
    One instance drives ONE stream: the stage workspaces, the launch plans' canvas and the decoder's scratch are persistent and reused
    from call to call in stream order (a second stream needs a second instance).  forward_async additionally owns one internal side stream
    for the decoder (decode_on_side_stream); everything it returns is ordered for the caller by PendingDetections.result()."""

    def __init__(self, det, dtype=torch.bfloat16, hip_conv=None):
        super().__init__()
        import os

        if hip_conv is None:
            hip_conv = os.environ.get("PNX_HIP_CONV", "1") != "0"
        self._ws = {}
        # launch plans (plan.py / pnx_enqueue): the backbone and the head as one C call each; PNX_PLAN=0 issues every launch from Python
        self.use_plan = os.environ.get("PNX_PLAN", "1") != "0"
        self.sparse_ws = os.environ.get("PNX_SPARSE_WS", "1") != "0"
        self.tile_lists = os.environ.get("PNX_TILE_LISTS", "1") != "0"
        # the decoder on its own stream, launched behind the NEXT batch's reader (forward_async / _DeferredDecode): +4.8 % frames/s in bench.py's
        # serving loop with the reader's in-loop time unchanged (round 5); PNX_DECODE_STREAM=0: everything on one stream
        self.decode_on_side_stream = os.environ.get("PNX_DECODE_STREAM", "1") != "0"
        self.reader = det.reader
        self.post_processing = det.post_processing
        self.head_ref = det.head  # predict() / rectifier / class bookkeeping
        self.dtype = dtype
        self._decoder = None
        bb = det.backbone
        self.stages = nn.ModuleList()
        self.stage_meta = []
        for blk in bb.blocks:
            mods = nn.ModuleList()
            first = blk[0]
            w, b = _fold_bn(first.conv.weight, first.norm)
            mods.append(_backbone_conv(w, b, first.stride, first.kernel_size // 2, dtype, hip_conv))
            for rb in list(blk)[1:]:
                w1, b1 = _fold_bn(rb.block1.conv.weight, rb.block1.norm)
                w2, b2 = _fold_bn(rb.conv2.weight, rb.norm2)
                mods.append(_backbone_conv(w1, b1, 1, rb.block1.kernel_size // 2, dtype, hip_conv))
                mods.append(_backbone_conv(w2, b2, 1, rb.conv2.kernel_size[0] // 2, dtype, hip_conv))
            self.stages.append(mods)
            self.stage_meta.append((first.stride, first.subm))
        w, b = _fold_bn(bb.mapping[0].weight, bb.mapping[1])
        self.mapping = _FusedConv(w, b, 1, 0, dtype=dtype)
        nk = det.neck
        self.pre1 = _FusedConv(*_fold_bn(nk.pre_conv.block1.conv.conv.weight, nk.pre_conv.block1.norm), 1, 1, dtype=dtype)
        self.pre2 = _FusedConv(*_fold_bn(nk.pre_conv.block2.conv.conv.weight, nk.pre_conv.block2.norm), 1, 1, dtype=dtype)
        # ASPP (aspp.py:19-32): the identity, the 1x1 branch and the four dilated branches have no BN/activation of their own and
        # post_conv is a 1x1 over their concat, so it distributes over the branches:
        #     post(cat(x, conv1x1(x), conv_d(x, Ws)...)) = conv1x1(x, P0 + P1.W1) + sum_d conv_d(x, P_d.Ws)
        # (P_k = the k-th 256-column block of the BN-folded post weight), and the 1x1 term is a centre tap, added to the dilation-1
        # kernel.  The 1536-channel concat, the 1536 -> 256 GEMM and the 1x1 branch disappear; the four partial results are summed
        # in fp32 by one pass (pnx_sum_bias_act) that also applies the folded-BN shift + ReLU.
        wds, bp = fold_aspp(nk)
        for k, wd in enumerate(wds):
            self.register_buffer(f"aspp_w{k}", wd.to(dtype).contiguous(memory_format=torch.channels_last))
        self.register_buffer("aspp_bias", bp.float().contiguous())
        hd = det.head
        sc = hd.shared_conv[0]
        self.shared = _backbone_conv(*_fold_bn(sc.weight, hd.shared_conv[1], sc.bias), sc.stride[0], sc.padding[0], dtype, hip_conv)
        self.task_deblock = nn.ModuleList()
        self.task_conv1 = nn.ModuleList()
        self.task_conv2 = nn.ModuleList()
        self.task_split = []
        self.task_chans = []
        # Lazy head (PNX_HEAD_LAZY, default on): per task only the class (and iou) branches run over the whole map; the five regression
        # branches (reg, height, dim, rot, vel = 5/6 of the head's convolution work and its 9.6 GB/step intermediate) are evaluated
        # at the <= pre_max candidates per (sample, class) that the decoder selects -- what CenterHead.predict does after the fact
        # (centerhead.py:341-363) done before the fact.
        self.lazy_head = bool(hip_conv) and dtype in ops._HALF and os.environ.get("PNX_HEAD_LAZY", "1") != "0"
        self.lazy_conv1, self.lazy_conv2 = nn.ModuleList(), nn.ModuleList()
        self._lazy_ok = []
        for task in hd.tasks:
            db = task.deblock
            w, b = _fold_bn(db.conv.conv.weight, db.norm, transposed=True)
            dc = db.conv.conv
            if hip_conv and dtype in ops._HALF and tuple(w.shape) == (64, 64, 2, 2) and tuple(dc.stride) == (2, 2) and tuple(dc.padding) == (0, 0):
                self.task_deblock.append(_HipDeconv2x2(w, b, dtype=dtype))
            else:
                self.task_deblock.append(_FusedConv(w, b, dc.stride, 0, transposed=True, dtype=dtype))
            names = list(task.heads.keys())
            w1s, b1s, w2s, b2s, outs = [], [], [], [], []
            for nme in names:
                fc = getattr(task, nme)
                assert len(fc) == 4, "merged head expects conv-bn-relu-conv branches"
                w1, b1 = _fold_bn(fc[0].weight, fc[1], fc[0].bias)
                w1s.append(w1)
                b1s.append(b1)
                w2s.append(fc[3].weight.detach().float())
                b2s.append(fc[3].bias.detach().float())
                outs.append(fc[3].weight.shape[0])
            hc = w1s[0].shape[0]
            W1 = torch.cat(w1s, 0)                                   # (nh*hc, 64, 3, 3)
            tot = sum(outs)
            tot_p = (tot + 7) // 8 * 8                               # epilogue kernel wants channels % 8 == 0
            hip_out = hip_conv and dtype in ops._HALF and hc == 64 and tot <= 16 and len(names) in (5, 6, 7)
            if hip_out:
                tot_p = 16                                           # k_sephead_out: 16 output channels (one MFMA M tile)
            W2 = torch.zeros((tot_p, hc * len(names), 3, 3), dtype=torch.float32, device=W1.device)
            B2 = torch.zeros((tot_p,), dtype=torch.float32, device=W1.device)
            o = 0
            for j, (w2, b2) in enumerate(zip(w2s, b2s)):             # block-diagonal: branch j only sees its own 64 channels
                W2[o:o + w2.shape[0], j * hc:(j + 1) * hc] = w2
                B2[o:o + w2.shape[0]] = b2
                o += w2.shape[0]
            if hip_conv and dtype in ops._HALF and (W1.shape[1], W1.shape[0]) in ops.CONV3X3_SHAPES_S1:
                self.task_conv1.append(_HipConv3x3(W1, torch.cat(b1s), 1, dtype=dtype))   # input tile staged once, reused for all 64-channel passes
            else:
                self.task_conv1.append(_FusedConv(W1, torch.cat(b1s), 1, 1, dtype=dtype))
            self.task_conv2.append(_HipSepHeadOut(W2, B2, dtype=dtype) if hip_out else _FusedConv(W2, B2, 1, 1, relu=False, dtype=dtype))
            self.task_chans.append(tot_p)
            self.task_split.append((names, outs))
            ti = len(self.task_split) - 1
            ok = (self.lazy_head and hip_out and names[:5] == ["reg", "height", "dim", "rot", "vel"] and outs[:5] == [2, 1, 3, 2, 2]
                  and names[5:] in (["hm"], ["iou", "hm"]))
            self._lazy_ok.append(ok)
            if ok:
                dn = list(range(5, len(names)))                          # dense branches: [iou] hm
                W1d, b1d = torch.cat([w1s[j] for j in dn], 0), torch.cat([b1s[j] for j in dn])
                W2d = torch.zeros((16, hc * len(dn), 3, 3), dtype=torch.float32, device=W1.device)
                B2d = torch.zeros((16,), dtype=torch.float32, device=W1.device)
                o = 0
                for q, j in enumerate(dn):
                    W2d[o:o + outs[j], q * hc:(q + 1) * hc] = w2s[j]
                    B2d[o:o + outs[j]] = b2s[j]
                    o += outs[j]
                self.lazy_conv1.append(_HipConv3x3(W1d, b1d, 1, dtype=dtype))
                self.lazy_conv2.append(_HipSepHeadOut(W2d, B2d, dtype=dtype))
                # regression branches as matrices over (tap, channel): conv1 (576 -> 320), conv2 (9 positions x 320 -> 10, block-diagonal)
                W1z = torch.cat(w1s[:5], 0)                               # (320, 64, 3, 3)
                w1m = W1z.permute(2, 3, 1, 0).reshape(9 * hc, 5 * hc)     # rows (ky, kx, cin)
                w2m = torch.zeros((9 * 5 * hc, 10), dtype=torch.float32, device=W1.device)
                b2z = torch.zeros((10,), dtype=torch.float32, device=W1.device)
                o = 0
                for j in range(5):
                    w2 = w2s[j]                                          # (k, 64, 3, 3)
                    for pos in range(9):
                        w2m[pos * 5 * hc + j * hc: pos * 5 * hc + (j + 1) * hc, o:o + outs[j]] = w2[:, :, pos // 3, pos % 3].t()
                    b2z[o:o + outs[j]] = b2s[j]
                    o += outs[j]
                self.register_buffer(f"lazy_w1_{ti}", w1m.to(dtype).contiguous())
                self.register_buffer(f"lazy_b1_{ti}", torch.cat(b1s[:5]).float().contiguous())
                self.register_buffer(f"lazy_w2_{ti}", w2m.to(dtype).float().contiguous())   # the dense kernels hold W2 in the element type
                self.register_buffer(f"lazy_b2_{ti}", b2z)
                self.register_buffer(f"lazy_wf1_{ti}", ops.conv3x3_pack_weights(W1z, dtype=dtype))
                self.register_buffer(f"lazy_w2c_{ti}", ops.sephead_lazy_pack_w2(getattr(self, f"lazy_w2_{ti}")))
            else:
                self.lazy_conv1.append(nn.Identity())
                self.lazy_conv2.append(nn.Identity())
        self.lazy_head = self.lazy_head and all(self._lazy_ok)

    @torch.no_grad()
    def forward_preds(self, points, batch_size, marks=None, packed_out=None, taps=None, lazy=None, after_reader=None):
        """packed_out: a list that receives, per task, the packed NHWC head output -- or, with the lazy head (lazy=None: the model's
        setting), a LazyTask (dense [iou] hm map + deblocked features) for launch_decode()."""
        def mark(name):
            if marks is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((name, e))

        mark("start")
        ny, nx = (int(v) for v in self.reader.grid_size)
        planned = marks is None and taps is None and self._plan_ok()
        if planned:
            bb = self._backbone_plan(batch_size, points.device)
            self.reader.forward_dense(points, batch_size, dtype=self.dtype, out=bb["canvas"], occupancy=bb["occ"])
            if after_reader is not None:
                after_reader()
            bb["plan"].run()
            x, mask = bb["out"], bb["mask"]
        else:
            occ = torch.empty((batch_size, ny, nx), dtype=torch.uint8, device=points.device)
            x = self.reader.forward_dense(points, batch_size, dtype=self.dtype, occupancy=occ)
            mask = occ
            if after_reader is not None:
                after_reader()
        mark("reader")
        for si, (mods, (stride, subm)) in enumerate(zip(self.stages, self.stage_meta) if not planned else ()):
            if not subm:
                mask = ops.mask_pool3(mask, stride)
            ws, k = self._stage_workspace(si, mods, mask), 0
            tiles = self._stage_tiles(si, mods, mask, ws)

            def run(m, inp, res=None):
                nonlocal k
                if ws is None or not isinstance(m, _HipConv3x3):   # the strided entry conv of a 256-channel stage is MIOpen + epilogue
                    return m(inp, mask, residual=res)
                k += 1                                  # x, y, out of a block sit in three different buffers
                return m(inp, mask, residual=res, out=ws[(k - 1) % 3], tiles=tiles)

            x = run(mods[0], x)
            for j in range(1, len(mods), 2):
                y = run(mods[j], x)
                x = run(mods[j + 1], y, x)
            mark(f"backbone.stage{si}")
            if taps is not None:
                taps[f"stage{si}"] = x
        x = self.mapping(x, mask)
        # BasicBlock (utils/conv.py): act(block2(block1(x)) + x) where block2 already ends in a ReLU, so the residual is added
        # AFTER that ReLU; both terms are >= 0, which makes the trailing act() the identity.
        x = self.pre2(self.pre1(x), post_residual=x)
        parts = [F.conv2d(x, getattr(self, f"aspp_w{k}"), None, 1, d, d) for k, d in enumerate((1, 6, 12, 18))]
        parts = [p if p.is_contiguous(memory_format=torch.channels_last) else p.contiguous(memory_format=torch.channels_last) for p in parts]
        x = ops.sum_bias_act(parts, self.aspp_bias, relu=True)
        mark("mapping+neck")
        if taps is not None:
            taps["neck"] = x
        lazy = packed_out is not None and self.lazy_head and (lazy is None or lazy)
        if planned and lazy and self._head_plan_ok():
            packed_out.extend(self._run_head_plan(x))
            return []
        x = self.shared(x)
        preds = []
        for ti, (db, c1, c2, (names, outs_n)) in enumerate(zip(self.task_deblock, self.task_conv1, self.task_conv2, self.task_split)):
            up = db(x)
            if lazy:
                packed_out.append(LazyTask(self.lazy_conv2[ti](self.lazy_conv1[ti](up)), up))
                continue
            t = c2(c1(up))
            if packed_out is not None:
                packed_out.append(t)
                continue
            d, o = {}, 0
            for nme, k in zip(names, outs_n):
                d[nme] = t[:, o:o + k]
                o += k
            preds.append(d)
        mark("head")
        return preds

    # ------------------------------------------------------------------ launch plans (plan.py, include/pnx.h: pnx_enqueue)
    def _plan_ok(self):
        return (self.use_plan and self.sparse_ws and self.tile_lists and self.dtype in ops._HALF and not self.reader.training
                and self.reader._fused_supported() and all(isinstance(m, _HipConv3x3) for mods in self.stages for m in mods))

    def _head_plan_ok(self):
        return isinstance(self.shared, _HipConv3x3) and all(isinstance(d, _HipDeconv2x2) for d in self.task_deblock)

    def _backbone_plan(self, B, dev):
        """The backbone of a B-frame batch as ONE pnx_enqueue call: per stage [mask_pool3,] tile list, convolutions -- the same calls, buffers
        and order as the Python loop of forward_preds, frozen.  The reader writes the plan's persistent canvas / occupancy."""
        from .plan import LaunchPlan

        key = ("plan_bb", B, dev)
        st = self._ws.get(key)
        wkey = tuple(t.data_ptr() for mods in self.stages for m in mods for t in (m.wfrag, m.bias))   # .to() / a reload / a re-fold moves or replaces them
        if st is not None and st["weights_at"] == wkey:
            return st
        ny, nx = (int(v) for v in self.reader.grid_size)
        canvas = torch.empty((B, 64, ny, nx), dtype=self.dtype, device=dev, memory_format=torch.channels_last)
        occ = torch.empty((B, ny, nx), dtype=torch.uint8, device=dev)
        plan = LaunchPlan()
        x, mask = canvas, occ
        for si, (mods, (stride, subm)) in enumerate(zip(self.stages, self.stage_meta)):
            if not subm:
                H, W = mask.shape[1:]
                pooled = torch.empty((B, (H - 1) // stride + 1, (W - 1) // stride + 1), dtype=torch.uint8, device=dev)
                plan.mask_pool3(mask, pooled, stride)
                mask = pooled
            ws = self._stage_workspace(si, mods, mask)
            tiles, rows = self._stage_tile_buffers(si, mods, mask)
            if tiles is not None:
                plan.tile_list(mask, [w[1] for w in ws], rows, tiles)
            k = 0

            def run(m, inp, res=None):
                nonlocal k
                k += 1
                out = ws[(k - 1) % 3]
                plan.conv3x3(inp, m.wfrag, m.bias, m.cout, m.stride, mask, res, True, out=out, tiles=tiles if m.stride == 1 else None)
                return out[0]

            x = run(mods[0], x)
            for j in range(1, len(mods), 2):
                y = run(mods[j], x)
                x = run(mods[j + 1], y, x)
        st = self._ws[key] = {"canvas": canvas, "occ": occ, "plan": plan.freeze(), "out": x, "mask": mask, "weights_at": wkey}
        return st

    def _run_head_plan(self, x):
        """shared conv + per task (deblock, dense [iou] hm branches) as ONE pnx_enqueue call.  The deblocked maps and the dense maps are fresh
        tensors per step (the decoder's fallback may read them after later steps were enqueued); the intermediates are persistent."""
        from .plan import Dyn, LaunchPlan

        B, _, H, W = x.shape
        dev = x.device
        key = ("plan_head", B, H, W, dev, self.shared.wfrag.data_ptr())   # the weights' address: .to() after the plan was built moves them
        st = self._ws.get(key)
        T = len(self.task_deblock)

        def fresh():
            ups = [torch.empty((B, 64, 2 * H, 2 * W), dtype=self.dtype, device=dev, memory_format=torch.channels_last) for _ in range(T)]
            dense = [torch.empty((B, 16, 2 * H, 2 * W), dtype=self.dtype, device=dev, memory_format=torch.channels_last) for _ in range(T)]
            return ups, dense

        ups, dense = fresh()
        if st is None:
            plan = LaunchPlan()
            sh = torch.empty((B, self.shared.cout, H, W), dtype=self.dtype, device=dev, memory_format=torch.channels_last)
            plan.conv3x3(Dyn("x", x), self.shared.wfrag, self.shared.bias, self.shared.cout, 1, None, None, True, out=(sh, None))
            mids = {}
            for ti, db in enumerate(self.task_deblock):
                c1, c2 = self.lazy_conv1[ti], self.lazy_conv2[ti]
                plan.deconv2x2(sh, db.wfrag, db.bias, db.cout, Dyn(f"up{ti}", ups[ti]), db.relu)
                if c1.cout not in mids:      # the tasks run one after the other on the stream: one intermediate per width serves them all
                    mids[c1.cout] = torch.empty((B, c1.cout, 2 * H, 2 * W), dtype=self.dtype, device=dev, memory_format=torch.channels_last)
                plan.conv3x3(Dyn(f"up{ti}", ups[ti]), c1.wfrag, c1.bias, c1.cout, 1, None, None, True, out=(mids[c1.cout], None))
                plan.sephead_out(mids[c1.cout], c2.wfrag, c2.bias, Dyn(f"dense{ti}", dense[ti]))
            st = self._ws[key] = plan.freeze()
        st.bind("x", x)
        for ti in range(T):
            st.bind(f"up{ti}", ups[ti])
            st.bind(f"dense{ti}", dense[ti])
        st.run()
        return [LazyTask(d, u) for d, u in zip(dense, ups)]

    def _stage_tile_buffers(self, si, mods, mask):
        """(tile list, count) buffers of a stage and the tile rows of its kernels, without listing anything (see _stage_tiles)."""
        m = next((m for m in mods if isinstance(m, _HipConv3x3) and m.stride == 1), None)
        rows = ops.conv_tile_rows(m.cin, m.cout, 1) if m is not None else 0
        if rows <= 0:
            return None, 0
        key = ("tiles", si) + tuple(mask.shape) + (mask.device,)
        buf = self._ws.get(key)
        if buf is None:
            B, H, W = mask.shape
            n_tiles = B * ((H + rows - 1) // rows) * ((W + 31) // 32)
            buf = self._ws[key] = (torch.empty((n_tiles,), dtype=torch.int32, device=mask.device), torch.zeros((1,), dtype=torch.int32, device=mask.device))
        return buf, rows

    def _stage_workspace(self, si, mods, mask):
        """Three persistent (activation, row_dirty) pairs per backbone stage for its HIP convolutions: they then touch only the row
        segments that hold (or held, one frame ago) active sites -- ops.conv3x3_workspace / pnx.h row_dirty."""
        if not self.sparse_ws or not any(isinstance(m, _HipConv3x3) for m in mods):
            return None
        B, H, W = mask.shape
        key = (si, B, H, W, mask.device)
        if key not in self._ws:
            cout = next(m.cout for m in mods if isinstance(m, _HipConv3x3))
            self._ws[key] = [ops.conv3x3_workspace(B, cout, H, W, mask.device, self.dtype) for _ in range(3)]
        return self._ws[key]

    def _stage_tiles(self, si, mods, mask, ws):
        """Tile list of a stage (ops.conv_tile_list): the submanifold blocks share the stage's mask, so the tiles with an active site
        or a stale row in one of the stage's three buffers are listed once and every stride-1 convolution walks the list."""
        if ws is None or not self.tile_lists:
            return None
        m = next((m for m in mods if isinstance(m, _HipConv3x3) and m.stride == 1), None)
        rows = ops.conv_tile_rows(m.cin, m.cout, 1) if m is not None else 0
        if rows <= 0:
            return None
        key = ("tiles", si) + tuple(mask.shape) + (mask.device,)
        buf = self._ws.get(key)
        buf = ops.conv_tile_list(mask, [w[1] for w in ws], rows, out=buf)
        self._ws[key] = buf
        return buf

    def decoder(self):
        if self._decoder is None:
            from .decode import PackedDecoder

            hd = self.head_ref
            chans = list(self.task_chans)
            # decode.hip reads the packed channels as reg(2) height(1) dim(3) rot(2) vel(2) [iou(1)] hm(ncls): the order of the
            # YAML's common_heads dict decides the packing, so check it instead of decoding the wrong channels silently
            want = ["reg", "height", "dim", "rot", "vel"] + (["iou"] if hd.with_iou else []) + ["hm"]
            sizes = {"reg": 2, "height": 1, "dim": 3, "rot": 2, "vel": 2, "iou": 1}
            for (names, outs), ncls in zip(self.task_split, hd.num_classes):
                if list(names) != want or any(o != sizes.get(n, ncls) for n, o in zip(names, outs)) or ncls > 4:
                    raise ops.PnxError(f"fused decoder needs head branches {want} with sizes 2,1,3,2,2[,1],ncls<=4; got {list(zip(names, outs))} -- "
                                       "use SingleStageDetector (CenterHead.predict) for this head layout")
            self._decoder = PackedDecoder(hd.num_classes, hd.rectifier, self.post_processing, hd.with_iou, chans)
        return self._decoder

    @torch.no_grad()
    def lazy_eval(self, ups, local, seg_len, valid, segs):
        """The five regression branches of every task at the candidate cells (local (S, pre_max) = b*H*W + cell inside the list's task map,
        lists s = sample * nc_total + class) of the deblocked maps ups[t] (B,64,H,W channels_last bf16): conv3x3 (64 -> 5*64) + folded BN +
        ReLU at the 3 x 3 neighbours of every cell, rounded to bf16 like the dense kernel's intermediate, then the block-diagonal 3x3 conv
        (-> 10) at the cell; zero padding at the map border.  Returns (S, pre_max, 10) fp32 in the order reg 2, height 1, dim 3, rot 2,
        vel 2, rounded to bf16 like the dense output.  One HIP launch (ops.sephead_lazy); PNX_HEAD_LAZY_TORCH=1 runs the torch statement
        of the same arithmetic below instead (tests compare the two)."""
        if os.environ.get("PNX_HEAD_LAZY_TORCH", "0") != "1":
            tasks = [(ups[ti], getattr(self, f"lazy_wf1_{ti}"), getattr(self, f"lazy_b1_{ti}"), getattr(self, f"lazy_w2c_{ti}"),
                      getattr(self, f"lazy_b2_{ti}")) for ti in range(len(ups))]
            class_task = [ti for ti, (names, outs) in enumerate(self.task_split) for _ in range(outs[-1])]
            return ops.sephead_lazy(tasks, class_task, ups[0].shape[0], local, seg_len, local.shape[1])
        cand = torch.zeros((*local.shape, 10), dtype=torch.float32, device=local.device)
        for ti, st in enumerate(segs):
            cand[st] = self._lazy_eval_torch(ti, ups[ti], local[st].reshape(-1), valid[st].reshape(-1)).reshape(len(st), local.shape[1], 10)
        return cand

    def _lazy_eval_torch(self, ti, up, local, valid):
        B, C, H, W = up.shape
        n = local.shape[0]
        upf = up.permute(0, 2, 3, 1).reshape(B * H * W, C)
        local = torch.where(valid, local, torch.zeros_like(local))
        b, cell = local // (H * W), local % (H * W)
        y, x = cell // W, cell % W
        d = torch.arange(-2, 3, device=up.device)
        yy, xx = y[:, None] + d[None, :], x[:, None] + d[None, :]
        vy, vx = (yy >= 0) & (yy < H), (xx >= 0) & (xx < W)
        idx = (b[:, None, None] * H + yy.clamp(0, H - 1)[:, :, None]) * W + xx.clamp(0, W - 1)[:, None, :]
        patch = upf[idx.reshape(-1)].reshape(n, 5, 5, C) * (vy[:, :, None] & vx[:, None, :])[..., None].to(up.dtype)
        cols = patch.unfold(1, 3, 1).unfold(2, 3, 1).permute(0, 1, 2, 4, 5, 3).reshape(n * 9, 9 * C)
        t1 = torch.relu(cols.float() @ getattr(self, f"lazy_w1_{ti}").float() + getattr(self, f"lazy_b1_{ti}"))
        inside = (vy[:, 1:4, None] & vx[:, None, 1:4]).reshape(n * 9, 1)
        t1 = (t1 * inside).to(up.dtype).float().reshape(n, -1)
        out = t1 @ getattr(self, f"lazy_w2_{ti}") + getattr(self, f"lazy_b2_{ti}")
        return out.to(up.dtype).float() * valid[:, None]

    @torch.no_grad()
    def forward_async(self, example):
        """Enqueue the whole frame batch (reader -> ... -> NMS -> D2H copy) and return a decode.PendingDetections."""
        ref = self.__dict__.get("_deferred")    # the previous batch's decoder, if nobody asked for its result yet -- held by WEAK reference: a handle the
        prev = ref() if ref is not None else None  # caller dropped is not launched at all, and its ~1.2 GB of deblocked maps go back to the allocator at once
        packed = []
        self.forward_preds(example["points"], example["batch_size"], packed_out=packed,
                           after_reader=(lambda: prev.launch(behind_reader=True)) if prev is not None and not prev.launched else None)
        if not (self.decode_on_side_stream and packed and isinstance(packed[0], LazyTask)):
            return self.launch_decode(packed, example.get("token"))
        # The decoder (keys, radix select, candidate evaluation, boxes, NMS, gather, D2H) is a chain of small latency-bound launches.  On its own
        # stream, and started only once the NEXT batch's reader is through (so that the HBM-bound reader keeps the GPU to itself), it runs beside
        # that batch's convolutions instead of in front of them.  It reads only this step's fresh tensors (dense [iou] hm maps, deblocked maps) and
        # constants; its scratch is per stream (decode.PackedDecoder).  Nobody enqueues a next batch: result() launches it at once.
        d = _DeferredDecode(self, packed, example.get("token"))
        self.__dict__["_deferred"] = weakref.ref(d)
        return d

    def __getstate__(self):   # copy.deepcopy / torch.save of the module: the side stream and the pending decoder are per-process launch state, not model state
        st = dict(self.__dict__)
        for k in ("_decode_stream", "_deferred"):
            st.pop(k, None)
        st["_decoder"] = None
        st["_ws"] = {}
        return st

    def launch_decode(self, packed, tokens=None):
        if not (packed and isinstance(packed[0], LazyTask)):
            return self.decoder().launch(packed, tokens)
        ups = [p.up for p in packed]

        def dense_path():  # the exact fallback (decode.PendingDetections.result): every branch over the whole map
            return self.decoder().launch([c2(c1(u)) for u, c1, c2 in zip(ups, self.task_conv1, self.task_conv2)], tokens)

        dec = self.decoder()
        if self.use_plan and dec.use_topk and dec.pre_max <= 4096 and os.environ.get("PNX_HEAD_LAZY_TORCH", "0") != "1":
            tasks = [(ups[ti], getattr(self, f"lazy_wf1_{ti}"), getattr(self, f"lazy_b1_{ti}"), getattr(self, f"lazy_w2c_{ti}"),
                      getattr(self, f"lazy_b2_{ti}")) for ti in range(len(ups))]
            class_task = [ti for ti, (names, outs) in enumerate(self.task_split) for _ in range(outs[-1])]
            return dec.launch_lazy_fused([p.dense for p in packed], tasks, class_task, tokens, dense_path)
        return dec.launch_lazy([p.dense for p in packed], lambda *a: self.lazy_eval(ups, *a), tokens, dense_path)

    @staticmethod
    def detections(outputs):
        det = {}
        for o in outputs:
            tok = o.pop("token")
            det[tok] = o  # already on the host
        return det

    @torch.no_grad()
    def forward(self, example):
        return self.detections(self.forward_async(example).result())
