"""boxes_iou3d_gpu / boxes_aligned_iou3d_gpu / nms_gpu of the reference's python helper module."""
from pillarnext_amd.box_torch_ops import boxes_aligned_iou3d_gpu, boxes_iou3d_gpu, nms_gpu  # noqa: F401
