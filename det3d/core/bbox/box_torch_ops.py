"""`rotate_nms_pcdet` on the HIP NMS."""
from pillarnext_amd.box_torch_ops import rotate_nms_pcdet  # noqa: F401
