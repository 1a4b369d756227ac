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
