"""Thin torch-tensor front-ends over the C ABI (include/pnx.h).  PyTorch is used for device memory and the
current stream only; every computation below happens in libpnx_hip.so."""
import ctypes

import torch

from . import _lib
from ._lib import PNX_BF16, PNX_F16, PNX_F32, PNX_NCHW, PNX_NHWC, PnxError, check, lib, ptr, stream_ptr

_DT = {torch.float32: PNX_F32, torch.bfloat16: PNX_BF16, torch.float16: PNX_F16}


def _need_cuda(t, name):
    if not t.is_cuda:
        raise PnxError(f"{name} must be a CUDA (ROCm) tensor; the PillarNeXt hot path has no CPU implementation")
    if not t.is_contiguous():
        raise PnxError(f"{name} must be contiguous")


class Workspace:
    """Grow-only device scratch buffer, one per (device, stream user)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        return self.buf


# --------------------------------------------------------------------------------------------- reader
def fold_bn(F, w0, bn0, w1, bn1, eps, out=None):
    """bn = (gamma, beta, running_mean, running_var) fp32 CUDA tensors -> folded parameter buffer."""
    n = 32 * (F + 5) + 32 + 64 * 64 + 64 + 64 * 121 + 64 * 71  # PNX_PFN_FOLDED_FLOATS(F)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=w0.device)
    ts = [w0, *bn0, w1, *bn1]
    for t in ts:
        _need_cuda(t, "PFN parameter")
        if t.dtype != torch.float32:
            raise PnxError("PFN parameters must be fp32")
    if tuple(w0.shape) != (32, F + 5) or tuple(w1.shape) != (64, 64):
        raise PnxError(f"PFN weights {tuple(w0.shape)}, {tuple(w1.shape)}: kernels are built for num_filters=[64, 64] only")
    check(lib().pnx_pfn_fold_bn(F, *[ptr(t) for t in ts[:5]], *[ptr(t) for t in ts[5:]], ctypes.c_float(eps), ptr(out), stream_ptr()),
          "pnx_pfn_fold_bn")
    return out


def reader_forward(points, batch, geom, folded, ws, canvas=None, canvas_layout=PNX_NHWC, occupancy=None, feat_max=None, coords=None,
                   unq_inv=None, pillar_of_point=None, counts=None):
    _need_cuda(points, "points")
    if points.dtype != torch.float32 or points.dim() != 2:
        raise PnxError("points must be (N, 1+F) fp32")
    n, stride = points.shape
    nbytes = lib().pnx_reader_workspace_bytes(n, batch, ctypes.byref(geom))
    buf = ws.get(nbytes, points.device)
    cap = 0
    if feat_max is not None:
        cap = feat_max.shape[0]
    if coords is not None:
        cap = coords.shape[0] if cap == 0 else min(cap, coords.shape[0])
    cdt = _DT[canvas.dtype] if canvas is not None else PNX_F32
    check(lib().pnx_reader_forward(ptr(points), n, stride, batch, ctypes.byref(geom), ptr(folded), ptr(canvas), cdt, canvas_layout,
                                   ptr(occupancy), ptr(feat_max), ptr(coords), cap, ptr(unq_inv), ptr(pillar_of_point), ptr(counts), ptr(buf),
                                   buf.numel(), stream_ptr()), "pnx_reader_forward")


def voxelize(points, batch, geom, ws, features=None, coords=None, unq_inv=None, pillar_of_point=None, counts=None):
    _need_cuda(points, "points")
    n, stride = points.shape
    nbytes = lib().pnx_reader_workspace_bytes(n, batch, ctypes.byref(geom))
    buf = ws.get(nbytes, points.device)
    cap = coords.shape[0] if coords is not None else 0
    check(lib().pnx_voxelize(ptr(points), n, stride, batch, ctypes.byref(geom), ptr(features), ptr(coords), cap, ptr(unq_inv),
                             ptr(pillar_of_point), ptr(counts), ptr(buf), buf.numel(), stream_ptr()), "pnx_voxelize")


def scatter_canvas(feat_max, coords, num_pillars_dev, batch, gy, gx, canvas, layout=PNX_NHWC):
    _need_cuda(feat_max, "feat_max")
    check(lib().pnx_scatter_canvas(ptr(feat_max), ptr(coords), ptr(num_pillars_dev), feat_max.shape[0], batch, gy, gx, ptr(canvas),
                                   _DT[canvas.dtype], layout, stream_ptr()), "pnx_scatter_canvas")


_SM_WS = Workspace()


class ScatterMax(torch.autograd.Function):
    """torch_scatter.scatter_max(x, index, dim=0)[0] with the argmax-routed gradient (pillar_encoder.py:43,180)."""

    @staticmethod
    def forward(ctx, x, index, num_pillars):
        _need_cuda(x, "x")
        x = x.float()
        n, C = x.shape
        out = torch.empty((num_pillars, C), dtype=torch.float32, device=x.device)
        arg = torch.empty((num_pillars, C), dtype=torch.int64, device=x.device)
        nbytes = lib().pnx_scatter_max_workspace_bytes(n, num_pillars)
        buf = _SM_WS.get(nbytes, x.device)
        check(lib().pnx_scatter_max(ptr(x), ptr(index), n, C, num_pillars, ptr(out), ptr(arg), ptr(buf), buf.numel(), stream_ptr()),
              "pnx_scatter_max")
        ctx.save_for_backward(arg)
        ctx.n = n
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        (arg,) = ctx.saved_tensors
        P, C = arg.shape
        g = grad_out.contiguous().float()
        gx = torch.empty((ctx.n, C), dtype=torch.float32, device=g.device)
        check(lib().pnx_scatter_max_backward(ptr(g), ptr(arg), ctx.n, C, P, ptr(gx), stream_ptr()), "pnx_scatter_max_backward")
        return gx, None, None


def scatter_max(x, index, num_pillars):
    return ScatterMax.apply(x.contiguous(), index.contiguous(), int(num_pillars))


# --------------------------------------------------------------------------------------------- voxel / multi-view readers (csrc/group.hip)
def group_geom(pc_range, voxel_size, mode, keep_range=None):
    """pnx_group_geom from the YAML lists: fp32 casts of min / voxel (what the reference applies, voxel_encoder.py:43-45), grid = np.round((max - min) /
    voxel) in fp64 (:40-41); keep_range (6 values) = MVFFeatureNet.forward's range mask on the raw x, y, z (mvf_encoder.py:290-297)."""
    import numpy as np

    pr, vs = np.asarray(pc_range, np.float64), np.asarray(voxel_size, np.float64)
    gs = (pr[3:] - pr[:3]) / vs
    gs = np.round(gs, 0, gs).astype(np.int64)
    g = _lib.PnxGroupGeom()
    for k in range(3):
        g.min[k], g.voxel[k], g.grid[k] = float(np.float32(pr[k])), float(np.float32(vs[k])), int(gs[k])
    g.mode, g.prefilter = int(mode), 0
    if keep_range is not None:
        g.prefilter = 1
        kr = torch.tensor([float(v) for v in keep_range], dtype=torch.float32)   # torch.tensor(self.pc_range, dtype=points.dtype): fp32 casts
        for k in range(3):
            g.keep_min[k], g.keep_max[k] = float(kr[k]), float(kr[3 + k])
    return g


_GROUP_WS = Workspace()


def group_points(points, batch, geom, want_features=True, want_mean=False, features_out=None):
    """pnx_group_points.  points (N, 1+F) fp32 CUDA.  Returns dict(features (N', C) or None, coords (G, 3|4) int32, unq_inv (N') int64,
    mean (G, M) or None, G, Nk) -- ONE host sync for the two counts (the reference's torch.unique syncs as well).  features_out = (buffer (>= N, ld)
    fp32, column offset): write the feature rows into columns [off, off + C) of an existing buffer (the concat of the two MVF views)."""
    _need_cuda(points, "points")
    if points.dtype != torch.float32 or points.dim() != 2:
        raise PnxError("points must be (N, 1+F) fp32")
    n, stride = points.shape
    dev = points.device
    voxel = geom.mode == _lib.PNX_GROUP_VOXEL
    C = stride - 1 if voxel else stride + 4
    M = stride - 1 if voxel else 3
    cells = batch * geom.grid[0] * geom.grid[1] * (geom.grid[2] if voxel else 1)
    cap = max(min(n, cells), 1)
    feats, ld, fptr = None, 0, None
    if features_out is not None:
        buf, off = features_out
        if not (buf.is_cuda and buf.dtype == torch.float32 and buf.dim() == 2 and buf.is_contiguous() and buf.shape[0] >= n and off + C <= buf.shape[1]):
            raise PnxError("group_points: features_out must be a contiguous fp32 (>= N, ld) buffer with room for the columns")
        feats, ld, fptr = buf, buf.shape[1], ctypes.c_void_p(buf.data_ptr() + 4 * off)
    elif want_features:
        feats = torch.empty((max(n, 1), C), dtype=torch.float32, device=dev)
        ld, fptr = C, ptr(feats)
    coords = torch.empty((cap, 4 if voxel else 3), dtype=torch.int32, device=dev)
    inv = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    mean = torch.empty((cap, M), dtype=torch.float32, device=dev) if want_mean else None
    counts = torch.zeros((2,), dtype=torch.int32, device=dev)
    nbytes = lib().pnx_group_workspace_bytes(n, stride, batch, ctypes.byref(geom))
    if nbytes == 0:
        raise PnxError("group_points: bad geometry")
    ws = _GROUP_WS.get(nbytes, dev)
    check(lib().pnx_group_points(ptr(points), n, stride, batch, ctypes.byref(geom), fptr, ld, ptr(coords), cap, ptr(inv), ptr(mean), ptr(counts), ptr(ws),
                                 ws.numel(), stream_ptr()), "pnx_group_points")
    G, Nk = (int(v) for v in counts.tolist())
    return {"features": None if feats is None else (feats if features_out is not None else feats[:Nk]), "coords": coords[:G], "unq_inv": inv[:Nk],
            "mean": None if mean is None else mean[:G], "G": G, "Nk": Nk}


def pfn_layer_eval(xa, gb, inv, wt, shift, num_groups, store=True, want_max=True):
    """pnx_pfn_layer_eval: relu(W' [xa | gb[inv]] + shift) per point (returned when store) and its per-cell maximum (num_groups, cout) (when want_max).
    xa (N, ca) fp32 -- may be a column slice of a wider contiguous buffer; gb (G, cb) fp32 or None; wt (ca + cb, cout) = W' transposed."""
    _need_cuda(wt, "wt")
    n = xa.shape[0]
    ca, cb, cout = xa.shape[1], 0 if gb is None else gb.shape[1], wt.shape[1]
    if not (xa.is_cuda and xa.dtype == torch.float32 and xa.stride(1) == 1 and wt.shape[0] == ca + cb and wt.dtype == torch.float32 and shift.numel() == cout):
        raise PnxError("pfn_layer_eval: xa (N, ca) fp32 with unit column stride, wt (ca + cb, cout) fp32, shift (cout)")
    if gb is not None:
        _need_cuda(gb, "gb")
    y = torch.empty((n, cout), dtype=torch.float32, device=xa.device) if store else None
    gmax = torch.empty((num_groups, cout), dtype=torch.float32, device=xa.device) if want_max else None
    check(lib().pnx_pfn_layer_eval(ptr(xa), xa.stride(0) if n > 1 else max(ca, 1), ca, ptr(gb), cb, ptr(inv), ptr(wt), ptr(shift), cout, n, num_groups, ptr(y), cout,
                                   ptr(gmax), stream_ptr()), "pnx_pfn_layer_eval")
    return y, gmax


def bilinear_gather(image, pos, pos_min, pos_voxel, cell_coords, unq_inv, ds_rate):
    """pnx_bilinear_gather: image (B, C, H, W) channels_last fp32 / bf16 / fp16; pos (N, >= 2) fp32 columns (may be a slice of a wider buffer); -> (N, C) fp32."""
    if not (image.is_cuda and image.dim() == 4 and image.is_contiguous(memory_format=torch.channels_last) and image.dtype in _DT):
        raise PnxError("bilinear_gather needs a channels_last fp32 / bf16 / fp16 CUDA map")
    if not (pos.is_cuda and pos.dtype == torch.float32 and pos.stride(1) == 1 and cell_coords.dtype == torch.int32 and unq_inv.dtype == torch.int64):
        raise PnxError("bilinear_gather: pos fp32 rows, int32 coords, int64 unq_inv")
    B, C, H, W = image.shape
    n = pos.shape[0]
    out = torch.empty((n, C), dtype=torch.float32, device=image.device)
    mn = (ctypes.c_float * 2)(float(pos_min[0]), float(pos_min[1]))
    vs = (ctypes.c_float * 2)(float(pos_voxel[0]), float(pos_voxel[1]))
    check(lib().pnx_bilinear_gather(ptr(image), _DT[image.dtype], B, H, W, C, ptr(pos), pos.stride(0) if n > 1 else 2, mn, vs, ptr(cell_coords.contiguous()),
                                    ptr(unq_inv), int(ds_rate), n, ptr(out), C, stream_ptr()), "pnx_bilinear_gather")
    return out


# --------------------------------------------------------------------------------------------- IoU / NMS
def _boxes(t, name):
    _need_cuda(t, name)
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != 7:
        raise PnxError(f"{name} must be (N, 7) fp32")


def boxes_overlap_bev(a, b, out):
    _boxes(a, "boxes_a"), _boxes(b, "boxes_b"), _need_cuda(out, "out")
    check(lib().pnx_boxes_overlap_bev(ptr(a), a.shape[0], ptr(b), b.shape[0], ptr(out), stream_ptr()), "pnx_boxes_overlap_bev")


def boxes_iou_bev(a, b, out):
    _boxes(a, "boxes_a"), _boxes(b, "boxes_b"), _need_cuda(out, "out")
    check(lib().pnx_boxes_iou_bev(ptr(a), a.shape[0], ptr(b), b.shape[0], ptr(out), stream_ptr()), "pnx_boxes_iou_bev")


def boxes_aligned_overlap_bev(a, b, out):
    _boxes(a, "boxes_a"), _boxes(b, "boxes_b"), _need_cuda(out, "out")
    if a.shape[0] != b.shape[0]:
        raise PnxError("aligned overlap needs equally many boxes")
    check(lib().pnx_boxes_aligned_overlap_bev(ptr(a), ptr(b), a.shape[0], ptr(out), stream_ptr()), "pnx_boxes_aligned_overlap_bev")


def boxes_aligned_iou3d(a, b):
    _boxes(a, "boxes_a"), _boxes(b, "boxes_b")
    if a.shape[0] != b.shape[0]:
        raise PnxError("aligned IoU needs equally many boxes")
    out = torch.empty((a.shape[0], 1), dtype=torch.float32, device=a.device)
    check(lib().pnx_boxes_aligned_iou3d(ptr(a), ptr(b), a.shape[0], ptr(out), stream_ptr()), "pnx_boxes_aligned_iou3d")
    return out


_NMS_WS = Workspace()


def nms_batched(boxes, seg_offsets, thresh, max_seg_len, post_max=0, rotated=True, seg_len=None):
    """boxes (T,7) score-sorted inside each segment; seg_offsets int32 (S+1) device; thresh fp32 (S) device.
    Returns keep (T) int32 [segment-local indices, ascending, first keep_count[s] valid per segment] and keep_count (S)."""
    _boxes(boxes, "boxes")
    S = seg_offsets.numel() - 1
    T = boxes.shape[0]
    keep = torch.empty((max(T, 1),), dtype=torch.int32, device=boxes.device)
    cnt = (torch.empty if S > 0 and max_seg_len > 0 else torch.zeros)((max(S, 1),), dtype=torch.int32, device=boxes.device)   # the greedy scan writes every segment's count
    nbytes = lib().pnx_nms_workspace_bytes(T, S, max_seg_len)
    buf = _NMS_WS.get(nbytes, boxes.device)
    fn = lib().pnx_nms_rotated_batched if rotated else lib().pnx_nms_normal_batched
    check(fn(ptr(boxes), ptr(seg_offsets), ptr(seg_len), S, int(max_seg_len), ptr(thresh), int(post_max), ptr(keep), ptr(cnt), ptr(buf), buf.numel(),
             stream_ptr()), "pnx_nms_batched")
    return keep, cnt


def nms_single(boxes, thresh, rotated=True, post_max=0):
    """One score-sorted list -> (keep int32 device tensor, count python int). Syncs once, like nms_gpu does."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int32, device=boxes.device), 0
    off = torch.tensor([0, n], dtype=torch.int32, device=boxes.device)
    thr = torch.tensor([float(thresh)], dtype=torch.float32, device=boxes.device)
    keep, cnt = nms_batched(boxes, off, thr, n, post_max, rotated)
    k = int(cnt[0].item())
    return keep[:k], k


# --------------------------------------------------------------------------------------------- dense epilogue
def bias_act_mask_(x, bias, mask=None, residual=None, relu=True):
    """In place on a channels_last bf16 / fp16 (B,C,H,W) tensor: x = [relu](x + bias[c] [+ residual]) * mask[b,h,w];
    relu=2: (relu(x + bias[c]) + residual) * mask."""
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.is_contiguous(memory_format=torch.channels_last)):
        raise PnxError("bias_act_mask_ needs a channels_last bf16 / fp16 CUDA tensor")
    B, C, H, W = x.shape
    if residual is not None and not (residual.dtype == x.dtype and residual.shape == x.shape and residual.is_contiguous(memory_format=torch.channels_last)):
        raise PnxError("residual must match x (dtype, channels_last)")
    check(lib().pnx_bias_act_mask(ptr(x), ptr(residual), ptr(bias), ptr(mask), ptr(x), B * H * W, C, _DT[x.dtype], int(relu), stream_ptr()),
          "pnx_bias_act_mask")
    return x


def sum_bias_act(parts, bias, relu=True):
    """[relu](sum(parts) + bias[c]) for channels_last bf16 (B,C,H,W) tensors of one shape, summed in fp32 in one pass."""
    x = parts[0]
    for t in parts:
        if not (t.is_cuda and t.dtype in (torch.bfloat16, torch.float16) and t.dtype == x.dtype and t.shape == x.shape
                and t.is_contiguous(memory_format=torch.channels_last)):
            raise PnxError("sum_bias_act needs channels_last bf16 / fp16 CUDA tensors of one shape and dtype")
    B, C, H, W = x.shape
    out = torch.empty_like(x, memory_format=torch.channels_last)
    arr = (ctypes.c_void_p * len(parts))(*[t.data_ptr() for t in parts])
    check(lib().pnx_sum_bias_act(arr, len(parts), ptr(bias), ptr(out), B * H * W, C, _DT[x.dtype], 1 if relu else 0, stream_ptr()), "pnx_sum_bias_act")
    return out


def mask_pool3(mask, stride):
    """uint8 (B,H,W) occupancy -> occupancy after a 3x3/stride/pad-1 sparse conv."""
    B, H, W = mask.shape
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    out = torch.empty((B, Ho, Wo), dtype=torch.uint8, device=mask.device)
    check(lib().pnx_mask_pool3(ptr(mask), B, H, W, stride, ptr(out), stream_ptr()), "pnx_mask_pool3")
    return out


# (Cin, Cout) pairs pnx_conv3x3_bf16 has kernels for.  Stride 1: backbone blocks, + the merged SepHead branches (5/6/7 x 64).
# Stride 2: the entry convolutions of backbone stages 1-3 (no residual).
CONV3X3_SHAPES_S1 = {(64, 64), (64, 128), (128, 128), (256, 256), (256, 64), (64, 320), (64, 384), (64, 448)}
CONV3X3_SHAPES_S2 = {(64, 64), (64, 128), (128, 128), (128, 256), (256, 256)}


_HALF = (torch.bfloat16, torch.float16)   # element types of the convolution kernels (csrc/conv3x3.hip is built for both)


def _conv_fn(name, dtype):
    """The bf16 or IEEE-half entry point of a convolution call (pnx_<name>_bf16 / pnx_<name>_f16)."""
    return getattr(lib(), f"pnx_{name}_{'f16' if dtype == torch.float16 else 'bf16'}")


def conv3x3_pack_weights(w, transposed=False, dtype=torch.bfloat16):
    """(Cout, Cin, 3, 3) -> bf16 (or, dtype=torch.float16, fp16) MFMA-fragment order [tap][cin/16][cout/32][lane = kb*32 + n][8]  (csrc/conv3x3.hip).  transposed: the weights of
    the stride-1 data gradient instead, i.e. pack(w.flip(2, 3).transpose(0, 1)).  CUDA fp32 / bf16 weights take one HIP launch
    (pnx_conv3x3_pack_weights); anything else the torch statement below, which is also what the tests compare the kernel with."""
    co, ci = w.shape[:2]
    if dtype == torch.bfloat16 and w.is_cuda and w.dtype in (torch.float32, torch.bfloat16) and tuple(w.shape[2:]) == (3, 3) and co % 32 == 0 and ci % 32 == 0:
        wc = w.detach().contiguous()
        out = torch.empty((9 * co * ci,), dtype=torch.bfloat16, device=w.device)
        check(lib().pnx_conv3x3_pack_weights(ptr(wc), _DT[wc.dtype], co, ci, 1 if transposed else 0, ptr(out), stream_ptr()), "pnx_conv3x3_pack_weights")
        return out
    if transposed:
        w = w.flip(2, 3).transpose(0, 1)
        co, ci = ci, co
    v = w.detach().float().reshape(co // 32, 32, ci // 16, 2, 8, 3, 3)         # (mt, n, cb, kb, e, ky, kx)
    v = v.permute(5, 6, 2, 0, 3, 1, 4).contiguous()                               # (ky, kx, cb, mt, kb, n, e)
    return v.reshape(-1).to(dtype).contiguous()


def deconv2x2_pack_weights(w, dtype=torch.bfloat16):
    """ConvTranspose2d weight (Cin, Cout, 2, 2) -> bf16 / fp16 MFMA-fragment order [ky*2+kx][cin/16][cout/32][lane = kb*32 + n][8]."""
    ci, co = w.shape[:2]
    v = w.detach().float().permute(1, 0, 2, 3).reshape(co // 32, 32, ci // 16, 2, 8, 2, 2)   # (mt, n, cb, kb, e, ky, kx)
    v = v.permute(5, 6, 2, 0, 3, 1, 4).contiguous()                                          # (ky, kx, cb, mt, kb, n, e)
    return v.reshape(-1).to(dtype).contiguous()


def deconv2x2(x, wfrag, bias, cout, relu=True):
    """x (B,Cin,H,W) channels_last bf16 / fp16 -> [relu](conv_transpose2d(x, W, stride 2) + bias) as (B,Cout,2H,2W) channels_last, same dtype
    (wfrag packed in that dtype)."""
    if not (x.is_cuda and x.dtype in _HALF and x.is_contiguous(memory_format=torch.channels_last) and wfrag.dtype == x.dtype):
        raise PnxError("deconv2x2 needs a channels_last bf16 / fp16 CUDA tensor and weights of the same dtype")
    B, ci, H, W = x.shape
    y = torch.empty((B, cout, 2 * H, 2 * W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    check(_conv_fn("deconv2x2", x.dtype)(ptr(x), ptr(wfrag), ptr(bias), ptr(y), B, H, W, ci, cout, 1 if relu else 0, stream_ptr()), "pnx_deconv2x2_bf16")
    return y


def sephead_pack_weights(w2, dtype=torch.bfloat16):
    """Block-diagonal (16, nb*64, 3, 3) -> bf16 / fp16 fragment order [branch][tap][kc][lane = q*16 + o][8]  (csrc/conv3x3.hip::k_sephead_out)."""
    co, ci = w2.shape[:2]
    if co != 16 or ci % 64:
        raise PnxError("sephead_pack_weights wants a (16, nb*64, 3, 3) weight")
    v = w2.detach().float().reshape(16, ci // 64, 2, 4, 8, 3, 3)                # (o, j, kc, q, e, ky, kx)
    v = v.permute(1, 5, 6, 2, 3, 0, 4).contiguous()                               # (j, ky, kx, kc, q, o, e)
    return v.reshape(-1).to(dtype).contiguous()


def sephead_out(x, wfrag, bias):
    """x (B, nb*64, H, W) channels_last bf16 / fp16 -> (B, 16, H, W) channels_last, same dtype: the last 3x3 conv of every SepHead branch of a task."""
    if not (x.is_cuda and x.dtype in _HALF and x.is_contiguous(memory_format=torch.channels_last) and wfrag.dtype == x.dtype):
        raise PnxError("sephead_out needs a channels_last bf16 / fp16 CUDA tensor and weights of the same dtype")
    B, ci, H, W = x.shape
    y = torch.empty((B, 16, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    check(_conv_fn("sephead_out", x.dtype)(ptr(x), ptr(wfrag), ptr(bias), ptr(y), B, H, W, ci // 64, stream_ptr()), "pnx_sephead_out_bf16")
    return y


def sephead_lazy_pack_w2(w2m):
    """(9*320, 10) matrix of the five regression branches' second convolutions (rows (pos, channel), block diagonal over the branches
    reg 2 | height 1 | dim 3 | rot 2 | vel 2) -> fp32 [M tile 10][pos 9][channel 32][3]: per 32-channel tile the <= 3 outputs of its branch."""
    off, k = [0, 2, 3, 6, 8], [2, 1, 3, 2, 2]
    v = w2m.detach().float().reshape(9, 10, 32, 10)                               # (pos, mt, cl, o)
    out = torch.zeros((10, 9, 32, 3), dtype=torch.float32, device=w2m.device)
    for mt in range(10):
        j = mt // 2
        out[mt, :, :, :k[j]] = v[:, mt, :, off[j]:off[j] + k[j]]
    return out.contiguous()


class _PnxLazyTask(ctypes.Structure):
    _fields_ = [("up", ctypes.c_void_p), ("wfrag1", ctypes.c_void_p), ("bias1", ctypes.c_void_p), ("w2c", ctypes.c_void_p), ("bias2", ctypes.c_void_p),
                ("h", ctypes.c_int32), ("w", ctypes.c_int32)]


def sephead_lazy(tasks, class_task, batch, local, seg_len, pre_max):
    """The regression branches of every task at the candidate cells (csrc/conv3x3.hip::k_sephead_lazy, one launch).
    tasks: list of (up (B,64,H,W) channels_last bf16 / fp16 (one dtype for all tasks), wfrag1 in that dtype, bias1, w2c, bias2); class_task: task index of every class (host list);
    local int64 (batch*len(class_task), pre_max) = b*H*W + cell per slot; seg_len int32 (batch*len(class_task),).  -> (lists, pre_max, 10) fp32,
    rows behind seg_len zero."""
    arr = (_PnxLazyTask * len(tasks))()
    dt = tasks[0][0].dtype
    for i, (up, wf, b1, w2c, b2) in enumerate(tasks):
        if not (up.is_cuda and up.dtype in _HALF and up.dtype == dt and wf.dtype == dt and up.shape[1] == 64
                and up.is_contiguous(memory_format=torch.channels_last) and up.shape[0] == batch):
            raise PnxError("sephead_lazy needs 64-channel channels_last bf16 / fp16 CUDA tensors of one dtype (weights included)")
        arr[i] = _PnxLazyTask(up.data_ptr(), wf.data_ptr(), b1.data_ptr(), w2c.data_ptr(), b2.data_ptr(), up.shape[2], up.shape[3])
    nc = len(class_task)
    ct = (ctypes.c_int32 * nc)(*[int(v) for v in class_task])
    S = batch * nc
    if not (local.is_cuda and local.dtype == torch.int64 and local.numel() == S * pre_max and seg_len.dtype == torch.int32 and seg_len.numel() == S):
        raise PnxError("sephead_lazy: local must be int64 (lists*pre_max), seg_len int32 (lists)")
    local, seg_len = local.contiguous(), seg_len.contiguous()
    out = torch.empty((S, pre_max, 10), dtype=torch.float32, device=local.device)
    check(_conv_fn("sephead_lazy", dt)(arr, len(tasks), ct, nc, batch, ptr(local), ptr(seg_len), pre_max, ptr(out), stream_ptr()), "pnx_sephead_lazy_bf16")
    return out


_WGRAD_WS = {}


def conv3x3_wgrad(x, dy, mask, stride=1):
    """Weight gradient (Cout, Cin, 3, 3) fp32 of the masked 3x3 convolution (pad 1, stride 1 or 2): sum over the sites of `mask` (B,Ho,Wo uint8,
    the OUTPUT's active set) of dy[p] (x) x[stride * p + tap]; x, dy channels_last bf16 (csrc/conv_wgrad.hip).  Deterministic."""
    for tns, what in ((x, "x"), (dy, "dy")):
        if not (tns.is_cuda and tns.dtype == torch.bfloat16 and tns.dim() == 4 and tns.is_contiguous(memory_format=torch.channels_last)):
            raise PnxError(f"conv3x3_wgrad: {what} must be a channels_last bf16 CUDA tensor")
    B, ci, H, W = x.shape
    co = dy.shape[1]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if stride not in (1, 2) or tuple(dy.shape) != (B, co, Ho, Wo) or tuple(mask.shape) != (B, Ho, Wo) or mask.dtype != torch.uint8:
        raise PnxError("conv3x3_wgrad: stride 1 or 2, dy (B,Cout,Ho,Wo) and a uint8 (B,Ho,Wo) mask of the output sites")
    nbytes = int(lib().pnx_conv3x3_wgrad_workspace_bytes(ci, co))
    if nbytes == 0:
        raise PnxError(f"conv3x3_wgrad: no kernel for {ci} -> {co} channels")
    key = (nbytes, x.device, torch.cuda.current_stream().cuda_stream)   # the partials of a call live until its reduction ran: one buffer per stream
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = _WGRAD_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dw = torch.empty((co, ci, 3, 3), dtype=torch.float32, device=x.device)
    check(lib().pnx_conv3x3_wgrad_bf16(ptr(x), ptr(dy), ptr(mask), ptr(dw), B, H, W, ci, co, stride, ptr(ws), ws.numel(), stream_ptr()), "pnx_conv3x3_wgrad_bf16")
    return dw


def split_f32(x, mask=None):
    """fp32 tensor -> (hi, lo) bf16 tensors of the same shape and memory layout: hi = RNE(x), lo = RNE(x - hi) (pnx_split_f32).
    mask: uint8 (B,H,W) for a channels_last (B,C,H,W) x that is zero where mask is 0 -- those sites are not read."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.numel() % 8 == 0):
        raise PnxError("split_f32 needs an fp32 CUDA tensor with a multiple of 8 elements")
    if not (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))):
        raise PnxError("split_f32 needs a dense tensor (contiguous or channels_last)")
    c = 0
    if mask is not None:
        if not (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and mask.dtype == torch.uint8 and mask.is_contiguous()
                and tuple(mask.shape) == (x.shape[0], x.shape[2], x.shape[3]) and x.shape[1] % 8 == 0):
            raise PnxError("split_f32: mask needs a channels_last (B,C,H,W) tensor with C % 8 == 0 and a contiguous uint8 (B,H,W) mask")
        c = x.shape[1]
    hi, lo = torch.empty_like(x, dtype=torch.bfloat16), torch.empty_like(x, dtype=torch.bfloat16)
    check(lib().pnx_split_f32(ptr(x), ptr(hi), ptr(lo), x.numel(), ptr(mask), c, stream_ptr()), "pnx_split_f32")
    return hi, lo


CONV3X3_X3_SHAPES = {1: {(64, 64), (128, 128), (256, 256)}, 2: {(64, 128), (128, 256), (256, 256)}}   # stride -> (Cin, Cout) of pnx_conv3x3_x3


def conv3x3_x3(x_hi, x_lo, wfrag_hi, wfrag_lo, cout, stride, mask, bias=None):
    """fp32 (B,Cout,Ho,Wo) channels_last = masked 3x3 convolution of x_hi + x_lo with W_hi + W_lo (three bf16 products accumulated in fp32 in one
    launch, pnx_conv3x3_x3) [+ bias, fp32 (Cout,)]; mask uint8 (B,Ho,Wo) of the OUTPUT sites or None (dense); zeros at inactive sites."""
    if bias is not None and not (bias.is_cuda and bias.dtype == torch.float32 and bias.numel() == cout and bias.is_contiguous()):
        raise PnxError("conv3x3_x3: bias must be a contiguous fp32 CUDA vector of Cout values")
    for tns in (x_hi, x_lo):
        if not (tns.is_cuda and tns.dtype == torch.bfloat16 and tns.dim() == 4 and tns.is_contiguous(memory_format=torch.channels_last)):
            raise PnxError("conv3x3_x3 needs channels_last bf16 CUDA halves")
    if x_hi.shape != x_lo.shape or wfrag_hi.dtype != torch.bfloat16 or wfrag_lo.dtype != torch.bfloat16:
        raise PnxError("conv3x3_x3: the two halves must have one shape, the weights must be packed bf16")
    B, ci, H, W = x_hi.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if mask is not None and (tuple(mask.shape) != (B, Ho, Wo) or mask.dtype != torch.uint8):
        raise PnxError("conv3x3_x3: mask must be uint8 (B,Ho,Wo)")
    y = torch.empty((B, cout, Ho, Wo), dtype=torch.float32, device=x_hi.device, memory_format=torch.channels_last)
    check(lib().pnx_conv3x3_x3(ptr(x_hi), ptr(x_lo), ptr(wfrag_hi), ptr(wfrag_lo), ptr(bias), ptr(mask), ptr(y), B, H, W, ci, cout, stride, stream_ptr()), "pnx_conv3x3_x3")
    return y


def conv3x3_wgrad_x3(x_hi, x_lo, dy_hi, dy_lo, mask, stride=1):
    """conv3x3_wgrad of x_hi + x_lo with dy_hi + dy_lo (three products, one pass: pnx_conv3x3_wgrad_x3): (Cout, Cin, 3, 3) fp32."""
    for tns in (x_hi, x_lo, dy_hi, dy_lo):
        if not (tns.is_cuda and tns.dtype == torch.bfloat16 and tns.dim() == 4 and tns.is_contiguous(memory_format=torch.channels_last)):
            raise PnxError("conv3x3_wgrad_x3 needs channels_last bf16 CUDA halves")
    B, ci, H, W = x_hi.shape
    co = dy_hi.shape[1]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if (stride not in (1, 2) or x_lo.shape != x_hi.shape or dy_lo.shape != dy_hi.shape or tuple(dy_hi.shape) != (B, co, Ho, Wo)
            or tuple(mask.shape) != (B, Ho, Wo) or mask.dtype != torch.uint8):
        raise PnxError("conv3x3_wgrad_x3: stride 1 or 2, halves of one shape, dy (B,Cout,Ho,Wo) and a uint8 (B,Ho,Wo) mask of the output sites")
    nbytes = int(lib().pnx_conv3x3_wgrad_workspace_bytes(ci, co))
    if nbytes == 0:
        raise PnxError(f"conv3x3_wgrad_x3: no kernel for {ci} -> {co} channels")
    key = (nbytes, x_hi.device, torch.cuda.current_stream().cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = _WGRAD_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=x_hi.device)
    dw = torch.empty((co, ci, 3, 3), dtype=torch.float32, device=x_hi.device)
    check(lib().pnx_conv3x3_wgrad_x3(ptr(x_hi), ptr(x_lo), ptr(dy_hi), ptr(dy_lo), ptr(mask), ptr(dw), B, H, W, ci, co, stride, ptr(ws), ws.numel(), stream_ptr()),
          "pnx_conv3x3_wgrad_x3")
    return dw


def conv3x3_dgrad_s2(g, wfrag_t, cin, in_hw, mask_in, g_lo=None, wfrag_t_lo=None):
    """Data gradient (B,cin,H,W) channels_last of a stride-2 masked 3x3 convolution from the upstream gradient g (B,cout,Ho,Wo) channels_last bf16 and the
    TRANSPOSED weight pack (conv3x3_pack_weights(w, transposed=True)); mask_in uint8 (B,H,W): the layer's input active set.  With g_lo / wfrag_t_lo (the
    bf16 low halves, split_f32) the three-product fp32 form: fp32 out."""
    H, W = in_hw
    B, co, Ho, Wo = g.shape
    x3 = g_lo is not None
    for tns in (g,) + ((g_lo,) if x3 else ()):
        if not (tns.is_cuda and tns.dtype == torch.bfloat16 and tns.dim() == 4 and tns.is_contiguous(memory_format=torch.channels_last) and tns.shape == g.shape):
            raise PnxError("conv3x3_dgrad_s2 needs channels_last bf16 CUDA gradients")
    if (Ho, Wo) != ((H - 1) // 2 + 1, (W - 1) // 2 + 1) or tuple(mask_in.shape) != (B, H, W) or mask_in.dtype != torch.uint8 or not mask_in.is_contiguous():
        raise PnxError("conv3x3_dgrad_s2: g (B,cout,Ho,Wo) with Ho = (H - 1) // 2 + 1 and a contiguous uint8 (B,H,W) mask of the input sites")
    dx = torch.empty((B, cin, H, W), dtype=torch.float32 if x3 else torch.bfloat16, device=g.device, memory_format=torch.channels_last)
    if x3:
        check(lib().pnx_conv3x3_dgrad_s2_x3(ptr(g), ptr(g_lo), ptr(wfrag_t), ptr(wfrag_t_lo), ptr(mask_in), ptr(dx), B, H, W, cin, co, stream_ptr()), "pnx_conv3x3_dgrad_s2_x3")
    else:
        check(lib().pnx_conv3x3_dgrad_s2_bf16(ptr(g), ptr(wfrag_t), ptr(mask_in), ptr(dx), B, H, W, cin, co, stream_ptr()), "pnx_conv3x3_dgrad_s2_bf16")
    return dx


def _smallk_check(x, what):
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous(memory_format=torch.channels_last)):
        raise PnxError(f"{what} must be a channels_last fp32 / bf16 CUDA tensor")


def conv3x3_smallk(x, weight, bias=None):
    """nn.Conv2d(64, k, 3, padding=1) with k <= 4 on a channels_last fp32 / bf16 map (pnx_conv3x3_smallk): (B,k,H,W) channels_last of x's dtype;
    weight (k,64,3,3) and bias (k) fp32."""
    _smallk_check(x, "conv3x3_smallk: x")
    k = weight.shape[0]
    if not (weight.is_cuda and weight.dtype == torch.float32 and tuple(weight.shape[1:]) == (64, 3, 3) and weight.is_contiguous() and 1 <= k <= 4
            and x.shape[1] == 64 and (bias is None or (bias.is_cuda and bias.dtype == torch.float32 and bias.numel() == k and bias.is_contiguous()))):
        raise PnxError("conv3x3_smallk: weight must be a contiguous fp32 (k,64,3,3) with k <= 4, bias fp32 (k)")
    B, _, H, W = x.shape
    y = torch.empty((B, k, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if k == 1:   # (B,1,H,W): channels_last and contiguous coincide, torch may report either stride set
        y = torch.empty((B, H, W, 1), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
    check(lib().pnx_conv3x3_smallk(ptr(x), ptr(weight), ptr(bias), ptr(y), B, H, W, 64, k, 0 if x.dtype == torch.float32 else 1, stream_ptr()), "pnx_conv3x3_smallk")
    return y


def conv3x3_smallk_wgrad(x, dy, want_bias=True):
    """(dw (k,64,3,3), dbias (k) or None), fp32, of conv3x3_smallk from x and the upstream gradient dy (B,k,H,W) (pnx_conv3x3_smallk_wgrad)."""
    _smallk_check(x, "conv3x3_smallk_wgrad: x")
    B, c, H, W = x.shape
    k = dy.shape[1]
    if not (c == 64 and 1 <= k <= 4 and dy.is_cuda and dy.dtype == x.dtype and tuple(dy.shape) == (B, k, H, W)):
        raise PnxError("conv3x3_smallk_wgrad: x (B,64,H,W), dy (B,k,H,W) of the same dtype, k <= 4")
    dyc = dy.permute(0, 2, 3, 1).contiguous()    # NHWC with k channels (a no-op for a channels_last gradient)
    nbytes = int(lib().pnx_conv3x3_smallk_wgrad_workspace_bytes(k))
    key = (nbytes, x.device, torch.cuda.current_stream().cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = _WGRAD_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dw = torch.empty((k, 64, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((k,), dtype=torch.float32, device=x.device) if want_bias else None
    check(lib().pnx_conv3x3_smallk_wgrad(ptr(x), ptr(dyc), ptr(dw), ptr(db), B, H, W, 64, k, 0 if x.dtype == torch.float32 else 1, ptr(ws), ws.numel(), stream_ptr()),
          "pnx_conv3x3_smallk_wgrad")
    return dw, db


def conv3x3_workspace(batch, cout, ho, wo, device, dtype=torch.bfloat16):
    """A persistent (output buffer, row_dirty flags) pair for conv3x3_masked(out=...): both start zeroed (pnx.h: row_dirty)."""
    y = torch.zeros((batch, cout, ho, wo), dtype=dtype, device=device).contiguous(memory_format=torch.channels_last)
    return y, torch.zeros((batch, ho, (wo + 31) // 32), dtype=torch.uint8, device=device)


def conv_tile_rows(cin, cout, stride=1):
    """Rows of a tile of the kernel that serves (cin, cout, stride); 0 = that kernel walks all tiles (takes no tile list)."""
    return int(lib().pnx_conv3x3_tile_rows(cin, cout, stride))


def conv_tile_list(mask, dirties, tile_rows, out=None):
    """(list int32[n_tiles], count int32[1]) of the tile_rows x 32 tiles of `mask` (B,H,W uint8) that hold an active site or a stale row
    of one of the `dirties` (row_dirty arrays of conv3x3_workspace buffers); pass it as tiles= to every stride-1 conv over this mask."""
    B, H, W = mask.shape
    n_tiles = B * ((H + tile_rows - 1) // tile_rows) * ((W + 31) // 32)
    if out is None:
        out = (torch.empty((n_tiles,), dtype=torch.int32, device=mask.device), torch.zeros((1,), dtype=torch.int32, device=mask.device))
    arr = (ctypes.c_void_p * max(len(dirties), 1))(*[d.data_ptr() for d in dirties])
    check(lib().pnx_conv_tile_list(ptr(mask), arr, len(dirties), B, H, W, tile_rows, ptr(out[0]), ptr(out[1]), stream_ptr()), "pnx_conv_tile_list")
    return out


def conv3x3_masked(x, wfrag, bias, cout, stride=1, mask=None, residual=None, relu=True, out=None, tiles=None):
    """x (B,Cin,H,W) channels_last bf16 / fp16 -> (B,Cout,Ho,Wo) channels_last, same dtype (wfrag packed in it); mask uint8 (B,Ho,Wo) of the OUTPUT sites.
    out = (y, row_dirty) from conv3x3_workspace: write into the persistent buffer, touching only row segments that are or were active.
    tiles = conv_tile_list(mask, ...) of the same mask (stride 1 only): walk the listed tiles instead of all of them."""
    if not (x.is_cuda and x.dtype in _HALF and x.is_contiguous(memory_format=torch.channels_last) and wfrag.dtype == x.dtype
            and (residual is None or residual.dtype == x.dtype)):
        raise PnxError("conv3x3_masked needs a channels_last bf16 / fp16 CUDA tensor, with weights and residual of the same dtype")
    B, ci, H, W = x.shape
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    if out is not None:
        y, dirty = out
        if mask is None or tuple(y.shape) != (B, cout, Ho, Wo) or tuple(dirty.shape) != (B, Ho, (Wo + 31) // 32) or y.dtype != x.dtype:
            raise PnxError("conv3x3_masked: out= needs a mask and a workspace of the output shape")
    else:
        y, dirty = torch.empty((B, cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last), None
    tl, tc = tiles if tiles is not None else (None, None)
    check(_conv_fn("conv3x3", x.dtype)(ptr(x), ptr(wfrag), ptr(bias), ptr(residual), ptr(mask), ptr(y), B, H, W, ci, cout, stride, 1 if relu else 0,
                                       ptr(dirty), ptr(tl), ptr(tc), stream_ptr()), "pnx_conv3x3")
    return y
