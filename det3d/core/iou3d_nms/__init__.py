"""`from det3d.core.iou3d_nms import iou3d_nms_cuda` gives the seven legacy entry points on libpnx_hip.so."""
from pillarnext_amd import iou3d_nms_cuda  # noqa: F401
