"""CenterPoint-style training losses of the reference (det3d/models/loss/centerloss.py), restated.

FastFocalLoss :8-37, RegLoss :40-60, IouLoss :63-87 (target = 2*IoU3D-1 from the native aligned IoU -- here the fused HIP
kernel pnx_boxes_aligned_iou3d), IouRegLoss :90-110 with the axis-aligned DIoU of :139-176.  Label tensors follow the
reference's assignment format (det3d/datasets/pipelines/assign.py:113-114): per task hm (B,C,H,W), ind (B,M) int64,
mask (B,M) uint8, cat (B,M) int64, anno_box (B,M,10), gt_boxes (B,M,7)."""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops


def gather_at(feat, ind):
    """(B,C,H,W), (B,M) -> (B,M,C): values of every channel at flat positions ind (centerloss.py:113-128)."""
    B, C = feat.shape[:2]
    flat = feat.permute(0, 2, 3, 1).reshape(B, -1, C)
    return flat.gather(1, ind.unsqueeze(2).expand(B, ind.shape[1], C))


class FastFocalLoss(nn.Module):
    def forward(self, out, target, ind, mask, cat):
        mask = mask.float()
        neg = (out.pow(2) * (1 - target).pow(4) * torch.log(1 - out)).sum()
        pos_pred = gather_at(out, ind).gather(2, cat.unsqueeze(2))          # (B,M,1)
        num_pos = mask.sum()
        pos = (torch.log(pos_pred) * (1 - pos_pred).pow(2) * mask.unsqueeze(2)).sum()
        if num_pos == 0:
            return -neg
        return -(pos + neg) / num_pos


class RegLoss(nn.Module):
    def forward(self, output, mask, ind, target):
        pred = gather_at(output, ind)
        m = mask.float().unsqueeze(2)
        nan = torch.isnan(target)
        target = torch.where(nan, pred.detach(), target)                     # NaN targets contribute nothing (:55-56)
        loss = (pred * m - target * m).abs() / (m.sum() + 1e-4)
        return loss.sum(dim=(0, 1))                                          # per box-code element


def diou_axis_aligned(p, g):
    """centerloss.py:139-176 (headings ignored): boxes (n,7) [x,y,z,dx,dy,dz,r]."""
    pmin, pmax = p[:, :2] - 0.5 * p[:, 3:5], p[:, :2] + 0.5 * p[:, 3:5]
    gmin, gmax = g[:, :2] - 0.5 * g[:, 3:5], g[:, :2] + 0.5 * g[:, 3:5]
    inter = torch.clamp(torch.minimum(pmax, gmax) - torch.maximum(pmin, gmin), min=0)
    outer = torch.clamp(torch.maximum(pmax, gmax) - torch.minimum(pmin, gmin), min=0)
    pz0, pz1 = p[:, 2] - 0.5 * p[:, 5], p[:, 2] + 0.5 * p[:, 5]
    gz0, gz1 = g[:, 2] - 0.5 * g[:, 5], g[:, 2] + 0.5 * g[:, 5]
    inter_h = torch.clamp(torch.minimum(pz1, gz1) - torch.maximum(pz0, gz0), min=0)
    outer_h = torch.clamp(torch.maximum(pz1, gz1) - torch.minimum(pz0, gz0), min=0)
    vi = inter[:, 0] * inter[:, 1] * inter_h
    vu = g[:, 3] * g[:, 4] * g[:, 5] + p[:, 3] * p[:, 4] * p[:, 5] - vi
    d_in = (g[:, :3] - p[:, :3]).pow(2).sum(-1)
    d_out = outer[:, 0] ** 2 + outer[:, 1] ** 2 + outer_h ** 2
    return torch.clamp(vi / vu - d_in / d_out, min=-1.0, max=1.0)


class IouRegLoss(nn.Module):
    def forward(self, box_pred, mask, ind, box_gt):
        if mask.sum() == 0:
            return box_pred.sum() * 0
        m = mask.bool()
        pred = gather_at(box_pred, ind)
        return (1.0 - diou_axis_aligned(pred[m], box_gt[m])).sum() / (m.sum() + 1e-4)


class IouLoss(nn.Module):
    def forward(self, iou_pred, mask, ind, box_pred, box_gt):
        if mask.sum() == 0:
            return iou_pred.sum() * 0
        m = mask.bool()
        pred = gather_at(iou_pred, ind)[m]
        pb = gather_at(box_pred, ind)[m].detach().float().contiguous()
        with torch.no_grad():
            target = 2 * ops.boxes_aligned_iou3d(pb, box_gt[m].float().contiguous()) - 1   # HIP: BEV overlap x height / union
        return F.l1_loss(pred, target, reduction="sum") / (m.sum() + 1e-4)


# --------------------------------------------------------------------------------------------- fused (HIP) form of the four losses
class _CenterLossFn(torch.autograd.Function):
    """All four CenterHead losses of one task in a handful of launches over the (B,M) lists (csrc/center_loss.hip): returns
    (hm_loss, box_loss_elem (10,), iou_loss, iou_reg_loss) exactly as FastFocalLoss / RegLoss / IouLoss / IouRegLoss above do."""

    @staticmethod
    def forward(ctx, hm, reg, height, dim, rot, vel, iou, hm_t, ind, mask, cat, anno, gtb, geom4, with_reg_iou):
        import ctypes

        from ._lib import check, lib, ptr, stream_ptr

        L = lib()
        maps = [t.contiguous() if t is not None else None for t in (hm, reg, height, dim, rot, vel, iou)]
        for t in maps:
            if t is not None and not (t.is_cuda and t.dtype == torch.float32):
                raise ops.PnxError("fused center loss needs fp32 CUDA head maps")
        B, C, H, W = maps[0].shape
        M = ind.shape[1]
        dev = maps[0].device
        # the kernels read the labels through raw pointers: int64 indices / classes, uint8 mask, fp32 targets, all on the maps' device
        # (an int32 `ind` or a label batch left on the host would be misread or fault; the module path raises a torch error there)
        hm_t, anno, gtb = (t.to(device=dev, dtype=torch.float32).contiguous() for t in (hm_t, anno, gtb))
        ind, cat = ind.to(device=dev, dtype=torch.int64).contiguous(), cat.to(device=dev, dtype=torch.int64).contiguous()
        mask = mask.to(device=dev, dtype=torch.uint8).contiguous()
        if ind.shape != cat.shape or ind.shape != mask.shape or anno.shape[:2] != ind.shape or gtb.shape[:2] != ind.shape:
            raise ops.PnxError(f"fused center loss: label shapes disagree (ind {tuple(ind.shape)}, cat {tuple(cat.shape)}, mask {tuple(mask.shape)}, "
                               f"anno_box {tuple(anno.shape)}, gt_boxes {tuple(gtb.shape)})")
        losses = torch.empty(16, dtype=torch.float32, device=dev)
        ws = torch.empty(int(L.pnx_center_loss_workspace_bytes(B, M)) + 256, dtype=torch.uint8, device=dev)
        arr = (ctypes.c_void_p * 7)(*[t.data_ptr() if t is not None else None for t in maps])
        g4 = (ctypes.c_float * 4)(*[float(v) for v in geom4])
        check(L.pnx_center_loss_forward(arr, ptr(hm_t), ptr(ind), ptr(mask), ptr(cat), ptr(anno), ptr(gtb), B, C, H, W, M, g4, int(bool(with_reg_iou)),
                                        ptr(losses), ptr(ws), ws.numel(), stream_ptr()), "pnx_center_loss_forward")
        ctx.save_for_backward(*maps, hm_t, ind, mask, cat, anno, gtb, losses, ws)
        ctx.misc = (tuple(float(v) for v in geom4), bool(with_reg_iou))
        return losses[0].clone(), losses[1:11].clone(), losses[11].clone(), losses[12].clone()

    @staticmethod
    def backward(ctx, g_hm, g_box, g_iou, g_ioureg):
        import ctypes

        from ._lib import check, lib, ptr, stream_ptr

        L = lib()
        saved = ctx.saved_tensors
        maps = list(saved[:7])
        hm_t, ind, mask, cat, anno, gtb, losses, ws = saved[7:]
        geom4, with_reg_iou = ctx.misc
        B, C, H, W = maps[0].shape
        M = ind.shape[1]
        dev = maps[0].device
        up = torch.cat([g_hm.reshape(1), g_box.reshape(10), g_iou.reshape(1), g_ioureg.reshape(1)]).float().contiguous()
        grads = [torch.empty_like(maps[0])] + [torch.zeros_like(t) if t is not None else None for t in maps[1:]]
        coef = torch.empty(4, dtype=torch.float32, device=dev)
        a_m = (ctypes.c_void_p * 7)(*[t.data_ptr() if t is not None else None for t in maps])
        a_g = (ctypes.c_void_p * 7)(*[t.data_ptr() if t is not None else None for t in grads])
        g4 = (ctypes.c_float * 4)(*geom4)
        check(L.pnx_center_loss_backward(a_m, a_g, ptr(hm_t), ptr(ind), ptr(mask), ptr(cat), ptr(anno), ptr(gtb), B, C, H, W, M, g4, int(with_reg_iou),
                                         ptr(losses), ptr(up), ptr(coef), ptr(ws), ws.numel(), stream_ptr()), "pnx_center_loss_backward")
        return (*grads, None, None, None, None, None, None, None, None)


def fused_center_loss(pd, hm_target, ind, mask, cat, anno_box, gt_boxes, geom4, with_reg_iou):
    """pd: the task's dict of head maps (hm = logits).  Returns (hm_loss, box_loss_elem, iou_loss, iou_reg_loss)."""
    f = lambda t: t.float()
    return _CenterLossFn.apply(f(pd["hm"]), f(pd["reg"]), f(pd["height"]), f(pd["dim"]), f(pd["rot"]), f(pd["vel"]),
                               f(pd["iou"]) if "iou" in pd else None, hm_target, ind, mask, cat, anno_box, gt_boxes, geom4, with_reg_iou)
