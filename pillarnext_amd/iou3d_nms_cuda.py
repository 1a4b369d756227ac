"""Drop-in for the reference's pybind extension ``det3d.core.iou3d_nms.iou3d_nms_cuda``
(det3d/core/iou3d_nms/src/iou3d_nms_api.cpp:11-19): the same seven functions, same argument order, same
"caller allocates the output, function returns an int" contract -- implemented on libpnx_hip.so.

Differences, on purpose: errors raise PnxError instead of exit(-1) (iou3d_nms.cpp:14-38), kernels run on the
current torch stream instead of the legacy default stream, and the NMS greedy scan runs on the device; only
the final `keep` indices cross to the host, because the legacy signature wants them in a CPU LongTensor.
"""
import torch

from . import ops


def boxes_aligned_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    """iou3d_nms.cpp:40-63 -- ans_overlap (N,1) <- overlap area of pair i."""
    ops.boxes_aligned_overlap_bev(boxes_a, boxes_b, ans_overlap)
    return 1


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    """iou3d_nms.cpp:65-85 -- ans_overlap (N,M)."""
    ops.boxes_overlap_bev(boxes_a, boxes_b, ans_overlap)
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    """iou3d_nms.cpp:87-106 -- ans_iou (N,M)."""
    ops.boxes_iou_bev(boxes_a, boxes_b, ans_iou)
    return 1


def _nms(boxes, keep, thresh, rotated):
    if keep.is_cuda or keep.dtype != torch.int64 or not keep.is_contiguous():
        raise ops.PnxError("keep must be a contiguous CPU int64 tensor (iou3d_nms.cpp:121)")
    k, num = ops.nms_single(boxes, float(thresh), rotated=rotated)
    if num:
        keep[:num] = k.to(torch.int64).cpu()
    return num


def nms_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d_nms.cpp:113-159 -- boxes (N,7) CUDA, score-sorted; keep (N) CPU int64; returns num_to_keep."""
    return _nms(boxes, keep, nms_overlap_thresh, True)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d_nms.cpp:162-211 -- axis-aligned variant."""
    return _nms(boxes, keep, nms_overlap_thresh, False)


def _host(t, name):
    if t.is_cuda or t.dtype != torch.float32:
        raise ops.PnxError(f"{name} must be a CPU fp32 tensor (iou3d_cpu.cpp:236-238)")
    return t.contiguous()


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """iou3d_cpu.cpp:232-252 -- CPU tensors in/out, no GPU involved: pnx_boxes_iou_bev_cpu, the device kernel's arithmetic compiled for the host
    (csrc/iou3d_geom.h), so the CPU and GPU entry points agree bit for bit."""
    from ._lib import check, lib, ptr

    a, b = _host(boxes_a, "boxes_a"), _host(boxes_b, "boxes_b")
    out = ans_iou if ans_iou.is_contiguous() and ans_iou.dtype == torch.float32 and not ans_iou.is_cuda else torch.empty(ans_iou.shape, dtype=torch.float32)
    check(lib().pnx_boxes_iou_bev_cpu(ptr(a), a.shape[0], ptr(b), b.shape[0], ptr(out)), "pnx_boxes_iou_bev_cpu")
    if out is not ans_iou:
        ans_iou.copy_(out)
    return 1


def boxes_aligned_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """iou3d_cpu.cpp:254-273 -- pairwise IoU-BEV, CPU tensors in/out (see boxes_iou_bev_cpu)."""
    from ._lib import check, lib, ptr

    a, b = _host(boxes_a, "boxes_a"), _host(boxes_b, "boxes_b")
    if a.shape[0] != b.shape[0]:
        raise ops.PnxError("aligned IoU needs equally many boxes")
    out = torch.empty((a.shape[0],), dtype=torch.float32)
    check(lib().pnx_boxes_aligned_iou_bev_cpu(ptr(a), ptr(b), a.shape[0], ptr(out)), "pnx_boxes_aligned_iou_bev_cpu")
    ans_iou.copy_(out.view(ans_iou.shape))
    return 1
