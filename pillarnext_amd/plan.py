"""Launch tables for include/pnx.h::pnx_enqueue: the C-ABI calls of a model section with their arguments frozen once, replayed per frame
batch by ONE call (csrc/enqueue.hip).  The builder methods mirror the wrappers of ops.py (same checks, same shapes) but record the call
instead of issuing it; tensors named as `dynamic` are re-bound per step (bind), everything else must stay alive and in place -- the
plan keeps references."""
import ctypes

import torch

from ._lib import PnxError, check, lib, stream_ptr

OP_MASK_POOL3, OP_TILE_LIST, OP_CONV3X3, OP_DECONV2X2, OP_SEPHEAD_OUT = 1, 2, 3, 4, 5


class PnxOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("i", ctypes.c_int32 * 9), ("p", ctypes.c_void_p * 11)]


assert ctypes.sizeof(PnxOp) == 128, "pnx_op layout drifted (include/pnx.h)"


class Dyn:
    """A tensor argument that changes from step to step: `like` gives the shape / dtype / layout every bound tensor must have."""

    def __init__(self, name, like):
        self.name, self.like = name, like


def _t(a):
    return a.like if isinstance(a, Dyn) else a


def _nhwc_half(x, what, like=None):
    """channels_last bf16 / fp16 (and, with `like`, of like's dtype: a convolution call runs in one element type)."""
    x = _t(x)
    if not (x.dtype in (torch.bfloat16, torch.float16) and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        raise PnxError(f"{what} needs a channels_last bf16 / fp16 tensor")
    if like is not None and x.dtype != _t(like).dtype:
        raise PnxError(f"{what}: dtype differs from the input's")
    return x


def _dt(x):
    return 2 if _t(x).dtype == torch.float16 else 1   # PNX_F16 / PNX_BF16: selects the pnx_*_f16 twin in csrc/enqueue.hip


class LaunchPlan:
    def __init__(self):
        self._ops, self._keep, self._dyn, self._arr = [], [], {}, None

    def __len__(self):
        return len(self._ops)

    def _add(self, kind, ints, ptrs):
        if self._arr is not None:
            raise PnxError("the plan is frozen")
        op = PnxOp()
        op.kind = kind
        for k, v in enumerate(ints):
            op.i[k] = int(v)
        for k, a in enumerate(ptrs):
            if isinstance(a, Dyn):    # bound per step: only the description of the tensor is kept
                meta = (a.like.shape, a.like.dtype, a.like.stride(), a.like.device)
                self._dyn.setdefault(a.name, (meta, []))[1].append((len(self._ops), k))
                op.p[k] = a.like.data_ptr()
            elif a is not None:
                op.p[k] = a.data_ptr()
                self._keep.append(a)
        self._ops.append(op)

    # ---- builders (ops.py: mask_pool3, conv_tile_list, conv3x3_masked, deconv2x2, sephead_out)
    def mask_pool3(self, mask_in, mask_out, stride):
        B, H, W = _t(mask_in).shape
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        if tuple(_t(mask_out).shape) != (B, Ho, Wo) or _t(mask_out).dtype != torch.uint8:
            raise PnxError("mask_pool3: mask_out must be uint8 (B, Ho, Wo)")
        self._add(OP_MASK_POOL3, [B, H, W, stride], [mask_in, mask_out])

    def tile_list(self, mask, dirties, tile_rows, out):
        B, H, W = _t(mask).shape
        n_tiles = B * ((H + tile_rows - 1) // tile_rows) * ((W + 31) // 32)
        if len(dirties) > 8 or out[0].numel() < n_tiles or out[0].dtype != torch.int32 or out[1].dtype != torch.int32:
            raise PnxError("tile_list: at most 8 row_dirty arrays, int32 list of >= n_tiles entries + int32 count")
        self._add(OP_TILE_LIST, [len(dirties), B, H, W, tile_rows], [mask, out[0], out[1]] + list(dirties))

    def conv3x3(self, x, wfrag, bias, cout, stride=1, mask=None, residual=None, relu=True, out=None, tiles=None):
        """out = (y, row_dirty) workspace pair, or (y, None) for a plain output buffer."""
        xs = _nhwc_half(x, "conv3x3")
        B, ci, H, W = xs.shape
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        y, dirty = out
        ys = _nhwc_half(y, "conv3x3 output", x)
        if wfrag.dtype != xs.dtype or (residual is not None and _t(residual).dtype != xs.dtype):
            raise PnxError("conv3x3: weights / residual of another dtype than the input")
        if tuple(ys.shape) != (B, cout, Ho, Wo) or (dirty is not None and (mask is None or tuple(dirty.shape) != (B, Ho, (Wo + 31) // 32))):
            raise PnxError("conv3x3: output / workspace of the wrong shape")
        tl, tc = tiles if tiles is not None else (None, None)
        self._add(OP_CONV3X3, [B, H, W, ci, cout, stride, 1 if relu else 0, _dt(x)], [x, wfrag, bias, residual, mask, y, dirty, tl, tc])

    def deconv2x2(self, x, wfrag, bias, cout, y, relu=True):
        xs = _nhwc_half(x, "deconv2x2")
        B, ci, H, W = xs.shape
        if tuple(_nhwc_half(y, "deconv2x2 output", x).shape) != (B, cout, 2 * H, 2 * W) or wfrag.dtype != xs.dtype:
            raise PnxError("deconv2x2: output of the wrong shape / weights of another dtype")
        self._add(OP_DECONV2X2, [B, H, W, ci, cout, 1 if relu else 0, _dt(x)], [x, wfrag, bias, y])

    def sephead_out(self, x, wfrag, bias, y):
        xs = _nhwc_half(x, "sephead_out")
        B, ci, H, W = xs.shape
        if tuple(_nhwc_half(y, "sephead_out output", x).shape) != (B, 16, H, W) or wfrag.dtype != xs.dtype:
            raise PnxError("sephead_out: output of the wrong shape / weights of another dtype")
        self._add(OP_SEPHEAD_OUT, [B, H, W, ci // 64, _dt(x)], [x, wfrag, bias, y])

    # ---- replay
    def freeze(self):
        self._arr = (PnxOp * max(len(self._ops), 1))(*self._ops)   # copies: the entries of the array are what bind() patches
        self._bound = {}
        return self

    def bind(self, name, t):
        if self._arr is None:
            self.freeze()
        meta, where = self._dyn[name]
        if (t.shape, t.dtype, t.stride(), t.device) != meta:
            raise PnxError(f"plan: tensor bound to '{name}' differs from the one the plan was built for")
        a = t.data_ptr()
        for k, j in where:
            self._arr[k].p[j] = a
        self._bound[name] = t   # alive until the next binding: the launches read / write it asynchronously

    def run(self):
        if self._arr is None:
            self.freeze()
        missing = set(self._dyn) - set(self._bound)
        if missing:   # an unbound slot would replay the build-time pointer, which nothing keeps alive
            raise PnxError(f"plan: dynamic tensors never bound: {sorted(missing)}")
        if not all(a.is_cuda for a in self._keep) or not all(t.is_cuda for t in self._bound.values()):
            raise PnxError("plan: every tensor of a launch table must live on the GPU")   # the library has no CPU path
        check(lib().pnx_enqueue(self._arr, len(self._ops), stream_ptr()), "pnx_enqueue")
