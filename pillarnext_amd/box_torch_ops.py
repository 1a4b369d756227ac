"""Mirror of det3d/core/bbox/box_torch_ops.py (rotate_nms_pcdet :5-31) and of
det3d/core/iou3d_nms/iou3d_nms_utils.py (boxes_iou3d_gpu :11-46, boxes_aligned_iou3d_gpu :49-89, nms_gpu :92-107)."""
import torch

from . import ops


def rotate_nms_pcdet(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """boxes (N,7) [x,y,z,dx,dy,dz,heading], scores (N) -> indices of the selected boxes (int64, device)."""
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous().float()
    if len(boxes) == 0:
        return order[:0]
    keep, num = ops.nms_single(boxes, float(thresh), rotated=True, post_max=int(post_max_size or 0))
    return order[keep.long()].contiguous()


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    assert boxes.shape[1] == 7
    return rotate_nms_pcdet(boxes, scores, thresh, pre_maxsize, None), None


def boxes_aligned_iou3d_gpu(boxes_a, boxes_b):
    """(N,7),(N,7) -> (N,1) 3-D IoU; one fused kernel instead of overlap kernel + 10 elementwise ops."""
    assert boxes_a.shape[0] == boxes_b.shape[0] and boxes_a.shape[1] == boxes_b.shape[1] == 7
    return ops.boxes_aligned_iou3d(boxes_a.contiguous().float(), boxes_b.contiguous().float())


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7),(M,7) -> (N,M) 3-D IoU."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a, b = boxes_a.contiguous().float(), boxes_b.contiguous().float()
    ov = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    ops.boxes_overlap_bev(a, b, ov)
    a_max = (a[:, 2] + a[:, 5] / 2).view(-1, 1)
    a_min = (a[:, 2] - a[:, 5] / 2).view(-1, 1)
    b_max = (b[:, 2] + b[:, 5] / 2).view(1, -1)
    b_min = (b[:, 2] - b[:, 5] / 2).view(1, -1)
    oh = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    o3 = ov * oh
    va = (a[:, 3] * a[:, 4] * a[:, 5]).view(-1, 1)
    vb = (b[:, 3] * b[:, 4] * b[:, 5]).view(1, -1)
    return o3 / torch.clamp(va + vb - o3, min=1e-6)
