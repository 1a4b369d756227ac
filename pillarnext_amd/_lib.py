"""ctypes binding of libpnx_hip.so (include/pnx.h).  There is no fallback: if the library is missing or a
call fails, a PnxError is raised -- the product path never routes through a CPU/eager substitute."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNX_LIB") or os.path.join(_HERE, "libpnx_hip.so")  # PNX_LIB: an instrumented build of the same library (tools/)

PNX_F32, PNX_BF16, PNX_F16 = 0, 1, 2
PNX_NHWC, PNX_NCHW = 0, 1


class PnxError(RuntimeError):
    pass


class PnxGeom(ctypes.Structure):
    _fields_ = [("pc_min", ctypes.c_float * 3), ("voxel", ctypes.c_float * 3), ("gx", ctypes.c_int32), ("gy", ctypes.c_int32)]


PNX_GROUP_VOXEL, PNX_GROUP_PILLAR_CLAMP, PNX_GROUP_CYLINDER_CLAMP = 0, 1, 2


class PnxGroupGeom(ctypes.Structure):   # include/pnx.h: pnx_group_geom
    _fields_ = [("min", ctypes.c_float * 3), ("voxel", ctypes.c_float * 3), ("grid", ctypes.c_int32 * 3), ("mode", ctypes.c_int32),
                ("prefilter", ctypes.c_int32), ("keep_min", ctypes.c_float * 3), ("keep_max", ctypes.c_float * 3)]


_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_f32 = ctypes.c_float
_sz = ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/pnx.h declares (tests/test_capi_symbols.py checks the two agree)
PROTOTYPES = {
    "pnx_last_error": (ctypes.c_char_p, []),
    "pnx_version": (ctypes.c_char_p, []),
    "pnx_geom_init": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(PnxGeom)]),
    "pnx_pfn_fold_bn": (ctypes.c_int, [_i32] + [_vp] * 10 + [_f32, _vp, _vp]),
    "pnx_reader_workspace_bytes": (_sz, [_i64, _i32, ctypes.POINTER(PnxGeom)]),
    "pnx_reader_forward": (ctypes.c_int, [_vp, _i64, _i32, _i32, ctypes.POINTER(PnxGeom), _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp,
                                          _vp, _vp, _sz, _vp]),
    "pnx_reader_fill_split": (None, [ctypes.POINTER(ctypes.c_int32)]),
    "pnx_profile_begin": (ctypes.c_int, [_i32]),
    "pnx_profile_end": (ctypes.c_int, [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)]),
    "pnx_profile_last_pfn_us": (ctypes.c_float, []),
    "pnx_profile_last_voxelize_us": (ctypes.c_float, []),
    "pnx_pfn_train_param_floats": (_sz, [_i32]),
    "pnx_pfn_train_partial_floats": (_sz, [_i32, _i32]),
    "pnx_pfn_forward_train": (ctypes.c_int, [_i32, _vp, _i64, _i32, _i32, ctypes.POINTER(PnxGeom), _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pnx_pfn_backward": (ctypes.c_int, [_i32, _i64, _i32, _i32, ctypes.POINTER(PnxGeom), _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pnx_voxelize": (ctypes.c_int, [_vp, _i64, _i32, _i32, ctypes.POINTER(PnxGeom), _vp, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pnx_group_workspace_bytes": (_sz, [_i64, _i32, _i32, ctypes.POINTER(PnxGroupGeom)]),
    "pnx_group_points": (ctypes.c_int, [_vp, _i64, _i32, _i32, ctypes.POINTER(PnxGroupGeom), _vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pnx_pfn_layer_eval": (ctypes.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i64, _i64, _vp, _i32, _vp, _vp]),
    "pnx_bilinear_gather": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _vp,
                                           _vp, _i32, _i64, _vp, _i32, _vp]),
    "pnx_scatter_max_workspace_bytes": (_sz, [_i64, _i64]),
    "pnx_scatter_max": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _sz, _vp]),
    "pnx_scatter_max_backward": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "pnx_scatter_canvas": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "pnx_merge_sweeps_desc_bytes": (_sz, []),
    "pnx_merge_sweeps_workspace_bytes": (_sz, [_i64]),
    "pnx_merge_sweeps": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "pnx_bias_act_mask": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "pnx_sum_bias_act": (ctypes.c_int, [_vp, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "pnx_deconv2x2_bf16": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pnx_deconv2x2_f16": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pnx_sephead_out_bf16": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pnx_sephead_out_f16": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pnx_masked_bn_blocks": (_i32, []),
    "pnx_masked_bn_reduce": (ctypes.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "pnx_masked_bn_finalize": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pnx_masked_bn_bwd_finalize": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pnx_masked_bn_stats": (ctypes.c_int, [_vp, _i32, _vp, _i64, _i32, _vp, _vp, _vp]),
    "pnx_masked_bn_apply": (ctypes.c_int, [_vp, _vp, _i32, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp]),
    "pnx_masked_bn_bwd_stats": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "pnx_masked_bn_bwd_apply": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "pnx_sephead_lazy_bf16": (ctypes.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "pnx_sephead_lazy_f16": (ctypes.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "pnx_conv3x3_bf16": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pnx_conv3x3_f16": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pnx_conv3x3_tile_rows": (ctypes.c_int, [_i32, _i32, _i32]),
    "pnx_conv3x3_pack_weights": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "pnx_conv3x3_wgrad_workspace_bytes": (_sz, [_i32, _i32]),
    "pnx_conv3x3_wgrad_bf16": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "pnx_conv3x3_wgrad_x3": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "pnx_conv3x3_dgrad_s2_bf16": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pnx_conv3x3_dgrad_s2_x3": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pnx_conv3x3_smallk": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pnx_conv3x3_smallk_wgrad_workspace_bytes": (_sz, [_i32]),
    "pnx_conv3x3_smallk_wgrad": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "pnx_conv_tile_list": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "pnx_mask_pool3": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "pnx_split_f32": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _vp]),
    "pnx_conv3x3_x3": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pnx_decode_task_desc_bytes": (_sz, []),
    "pnx_decode_topk_workspace_bytes": (_sz, [_i64, _i32]),
    "pnx_decode_topk": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pnx_decode_keys": (ctypes.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "pnx_sort_keys_workspace_bytes": (ctypes.c_size_t, [_i64]),
    "pnx_sort_keys": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "pnx_decode_boxes": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pnx_decode_boxes_lazy": (ctypes.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pnx_gather_kept": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "pnx_enqueue": (ctypes.c_int, [_vp, _i32, _vp]),
    "pnx_decode_lazy_enqueue": (ctypes.c_int, [_vp, _vp]),
    "pnx_op_bytes": (_sz, []),
    "pnx_lazy_decode_bytes": (_sz, []),
    "pnx_center_loss_workspace_bytes": (_sz, [_i32, _i32]),
    "pnx_center_loss_forward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _sz, _vp]),
    "pnx_center_loss_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pnx_boxes_overlap_bev": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "pnx_boxes_iou_bev": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "pnx_boxes_aligned_overlap_bev": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "pnx_boxes_aligned_iou3d": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "pnx_boxes_iou_bev_cpu": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp]),
    "pnx_boxes_aligned_iou_bev_cpu": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "pnx_nms_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "pnx_debug_nms_pair_cap": (_i32, [_i32]),
    "pnx_nms_rotated_batched": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "pnx_nms_normal_batched": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
}

_LIB = None


def lib():
    """Load libpnx_hip.so (built in-tree by pillarnext_amd/build.py or __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PnxError(f"{LIB_PATH} is missing: run `python -m pillarnext_amd.build` (hipcc, gfx950). "
                           "There is no CPU fallback for the product path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError here = the .so is stale w.r.t. include/pnx.h
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc, what=""):
    if rc != 0:
        msg = lib().pnx_last_error().decode(errors="replace")
        raise PnxError(f"{what} failed with status {rc}: {msg}")


def make_geom(pc_range, voxel_size):
    pr = (ctypes.c_double * 6)(*[float(v) for v in pc_range])
    vs = (ctypes.c_double * 3)(*[float(v) for v in voxel_size])
    g = PnxGeom()
    check(lib().pnx_geom_init(pr, vs, ctypes.byref(g)), "pnx_geom_init")
    return g


def ptr(t):
    """data pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
