"""Fused CenterHead.predict for packed head outputs (host side of csrc/decode.hip).

Semantics = det3d/models/heads/centerhead.py:231-384 (decode, score/range mask, IoU-rectified score, per-class
sort + top pre_max, rotated NMS, top post_max, task merge with label offsets), restructured so that the whole frame
batch needs one key kernel per task, ONE device sort, one box kernel, ONE batched NMS and ONE device->host copy."""
import ctypes
import struct

import weakref

import torch

from . import ops
from ._lib import PNX_BF16, PNX_F16, PNX_F32, PnxError, check, lib, ptr, stream_ptr


def _get(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


def pack_task(C, has_iou, ncls, cls_off, H, W, osf, vs, pc_range, score_thr, lim, rect, lazy=False):
    """Host descriptor matching `struct DecodeTask` in csrc/decode.hip.  lazy: the packed tensor is [iou] hm only."""
    lim6 = [float(v) for v in lim] if len(lim) > 0 else [0.0] * 6
    r4 = [float(v) for v in rect] + [0.0] * (4 - len(rect))
    o_iou = 0 if lazy else 10
    o_hm = (1 if has_iou else 0) if lazy else 10 + int(bool(has_iou))
    blob = struct.pack("6i6f6fi4f3i", int(C), int(bool(has_iou)), int(ncls), int(cls_off), int(H), int(W),
                       float(osf), float(vs[0]), float(vs[1]), float(pc_range[0]), float(pc_range[1]), float(score_thr),
                       *lim6, int(len(lim) > 0), *r4, int(o_hm), int(o_iou), int(bool(lazy)))
    assert len(blob) == lib().pnx_decode_task_desc_bytes(), "DecodeTask layout drifted"
    return blob


class _PnxLazyDecode(ctypes.Structure):
    """include/pnx.h: pnx_lazy_decode."""
    _fields_ = ([(n, ctypes.c_int32) for n in ("n_tasks", "n_classes_total", "batch", "pre_max", "post_max", "dtype")]
                + [(n, ctypes.c_void_p) for n in ("dense_host", "task_descs_host", "task_descs_dev", "task_key_off_host", "task_key_off_dev",
                                                 "list_key_off_dev", "lazy_tasks_host", "class_task_host", "seg_off_dev", "nms_thresh_dev", "keys",
                                                 "sorted_keys", "order", "seg_start", "local", "seg_len", "seg_total", "cand", "boxes9", "boxes7",
                                                 "scores", "flag", "keep", "keep_count", "topk_ws")]
                + [("topk_ws_bytes", ctypes.c_size_t), ("nms_ws", ctypes.c_void_p), ("nms_ws_bytes", ctypes.c_size_t)]
                + [(n, ctypes.c_void_p) for n in ("out", "out_host", "keep_count_host", "flag_host")])


class PackedDecoder:
    def __init__(self, num_classes, rectifier, test_cfg, has_iou, channels_per_task):
        self.num_classes = list(num_classes)
        self.nc_total = sum(num_classes)
        self.rectifier = rectifier
        self.has_iou = has_iou
        self.channels = list(channels_per_task)
        nms = _get(test_cfg, "nms")
        self.pre_max = int(_get(nms, "nms_pre_max_size"))
        self.post_max = int(_get(nms, "nms_post_max_size"))
        self.thr = [float(v) for t in _get(nms, "nms_iou_threshold") for v in t]
        self.cfg = test_cfg
        self._dev = {}
        self._topk_ws = None
        import os

        # PNX_DECODE_TOPK=1 (default): exact radix-select top-k in HIP (pnx_decode_topk: most significant digit first over the composite
        # (score key, key index), lists that hold fewer than pre_max valid keys finish after the first pass) -- exactly the selection of
        # a stable sort + cut (tests/test_gpu_decode.py::test_segmented_topk_equals_the_full_stable_sort).  PNX_DECODE_TOPK=0 selects the
        # bit-ranged stable key sort of all keys (pnx_sort_keys), PNX_DECODE_TORCH_SORT=1 the generic 64-bit torch.sort (cross-checks).
        self.use_topk = os.environ.get("PNX_DECODE_TOPK", "1") == "1"
        self.use_torch_sort = os.environ.get("PNX_DECODE_TORCH_SORT", "0") == "1"   # the generic 64-bit torch.sort (cross-check)
        self._sort_ws = None
        self._tptr = {}
        self._pin = {}

    def _descs(self, shapes):
        cfg = self.cfg
        out, off = [], 0
        for t, (H, W) in enumerate(shapes):
            out.append(pack_task(self.channels[t], self.has_iou, self.num_classes[t], off, H, W, _get(cfg, "out_size_factor")[t],
                                 _get(cfg, "voxel_size"), _get(cfg, "pc_range"), _get(cfg, "score_threshold"),
                                 _get(cfg, "post_center_limit_range"), self.rectifier[t]))
            off += self.num_classes[t]
        return out

    @torch.no_grad()
    def launch(self, packed, tokens=None):
        """packed: list (one per task) of (B, C_t, H, W) channels_last tensors, fp32 or bf16 (all the same dtype).
        Enqueues decode + NMS + the D2H copy and returns a PendingDetections."""
        B = packed[0].shape[0]
        dev = packed[0].device
        dt = {torch.float32: PNX_F32, torch.bfloat16: PNX_BF16, torch.float16: PNX_F16}[packed[0].dtype]
        T = len(packed)
        shapes = [(p.shape[2], p.shape[3]) for p in packed]
        for p in packed:
            assert p.is_contiguous(memory_format=torch.channels_last), "packed head outputs must be channels_last"
        descs = self._descs(shapes)
        sizes = [B * h * w for h, w in shapes]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        keys = torch.empty((offs[-1],), dtype=torch.int64, device=dev)
        L = lib()
        for t, p in enumerate(packed):
            kp = ctypes.c_void_p(keys.data_ptr() + 8 * offs[t])
            check(L.pnx_decode_keys(ptr(p), dt, B, self.nc_total, descs[t], kp, stream_ptr()), "pnx_decode_keys")
        S = B * self.nc_total
        ck = (B, T, dt, tuple(shapes), dev)
        if ck not in self._dev:
            tdesc = torch.frombuffer(bytearray(b"".join(descs)), dtype=torch.uint8).to(dev)
            koff = torch.tensor(offs, dtype=torch.int64, device=dev)
            seg_off = (torch.arange(S + 1, dtype=torch.int32, device=dev) * self.pre_max).contiguous()
            thr = torch.tensor(self.thr, dtype=torch.float32, device=dev).repeat(B)
            bounds = (torch.arange(S + 1, device=dev, dtype=torch.int64) << 32) ^ (-0x8000000000000000)
            self._dev[ck] = (tdesc, koff, seg_off, thr, bounds)
        tdesc, koff, seg_off, thr, bounds = self._dev[ck]
        if self.use_topk and self.pre_max <= 4096:
            # segmented top-k in HIP (csrc/decode.hip): exact radix select on (score key, key index) -> collect -> LDS sort of the <= pre_max
            # survivors of every list; no device sort of all keys, no searchsorted
            n_rows0 = S * self.pre_max
            skeys = torch.empty((n_rows0,), dtype=torch.int64, device=dev)
            order = torch.empty((n_rows0,), dtype=torch.int64, device=dev)
            seg_start = torch.empty((S,), dtype=torch.int64, device=dev)
            seg_len = torch.empty((S,), dtype=torch.int32, device=dev)
            wsb = int(L.pnx_decode_topk_workspace_bytes(offs[-1], S)) + 256
            if self._topk_ws is None or self._topk_ws.numel() < wsb or self._topk_ws.device != dev:
                self._topk_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            check(L.pnx_decode_topk(ptr(keys), offs[-1], S, self.pre_max, ptr(skeys), ptr(order), ptr(seg_start), ptr(seg_len), None, ptr(self._topk_ws),
                                    self._topk_ws.numel(), stream_ptr()), "pnx_decode_topk")
        elif self.use_torch_sort:
            # keys are non-negative when valid ... as int64 the all-ones key is -1: sort as unsigned by flipping the sign bit
            skeys, order = torch.sort(keys ^ (-0x8000000000000000), stable=True)
            skeys = skeys ^ (-0x8000000000000000)
            seg_start = torch.searchsorted(skeys ^ (-0x8000000000000000), bounds)
            seg_len = torch.clamp(seg_start[1:] - seg_start[:-1], max=self.pre_max).to(torch.int32)
        else:
            # the same stable sort over the 32 + bit_length(S) bits that can differ (pnx_sort_keys: 5 radix passes instead of 8)
            skeys = torch.empty_like(keys)
            order = torch.empty_like(keys)
            wsb = int(L.pnx_sort_keys_workspace_bytes(offs[-1]))
            if self._sort_ws is None or self._sort_ws.numel() < wsb or self._sort_ws.device != dev:
                self._sort_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            check(L.pnx_sort_keys(ptr(keys), offs[-1], S, ptr(skeys), ptr(order), ptr(self._sort_ws), self._sort_ws.numel(), stream_ptr()),
                  "pnx_sort_keys")
            seg_start = torch.searchsorted(skeys ^ (-0x8000000000000000), bounds)
            seg_len = torch.clamp(seg_start[1:] - seg_start[:-1], max=self.pre_max).to(torch.int32)
        # pointer table of the task tensors: a torch.tensor(list, device=...) is a blocking copy from pageable memory (it would make
        # the host wait for the whole network before it could enqueue the sort/NMS); cached per pointer tuple, else a pinned async copy
        pk = tuple(p.data_ptr() for p in packed)
        tptr = self._tptr.get(pk)
        if tptr is None:
            hp = torch.tensor(pk, dtype=torch.int64).pin_memory()
            tptr = torch.empty((T,), dtype=torch.int64, device=dev)
            tptr.copy_(hp, non_blocking=True)
            if len(self._tptr) > 64:
                self._tptr.clear()
            self._tptr[pk] = tptr
            self._tptr_host = hp  # keep the pinned source alive until the copy ran
        n_rows = S * self.pre_max
        boxes9 = torch.empty((n_rows, 9), dtype=torch.float32, device=dev)
        boxes7 = torch.zeros((n_rows, 7), dtype=torch.float32, device=dev)
        scores = torch.empty((n_rows,), dtype=torch.float32, device=dev)
        check(L.pnx_decode_boxes(ptr(tptr), ptr(tdesc), ptr(koff), T, dt, B, ptr(skeys), ptr(order), ptr(seg_start), ptr(seg_len), S,
                                 self.pre_max, ptr(boxes9), ptr(boxes7), ptr(scores), stream_ptr()), "pnx_decode_boxes")
        return self._finish(boxes9, boxes7, scores, seg_off, thr, seg_len, S, B, dev, tokens)

    def _finish(self, boxes9, boxes7, scores, seg_off, thr, seg_len, S, B, dev, tokens, flag=None, fallback=None):
        L = lib()
        keep, cnt = ops.nms_batched(boxes7, seg_off, thr, self.pre_max, post_max=self.post_max, seg_len=seg_len)
        out = torch.empty((S, self.post_max, 10), dtype=torch.float32, device=dev)
        check(L.pnx_gather_kept(ptr(boxes9), ptr(scores), ptr(keep), ptr(cnt), S, self.pre_max, self.post_max, ptr(out), stream_ptr()),
              "pnx_gather_kept")
        # the one device->host hand-off of the frame batch: asynchronous into pinned memory; PendingDetections.result() waits
        slot = self._pinned_slot(out.shape, S, out.dtype, cnt.dtype)
        out_h, cnt_h = slot["out"], slot["cnt"]
        out_h.copy_(out, non_blocking=True)
        cnt_h.copy_(cnt[:S], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pend = PendingDetections(ev, out_h, cnt_h, B, self.nc_total, tokens if tokens else [None] * B)
        if flag is not None:  # lazy head: the range-test flag rides on the same stream; result() falls back to the dense path if it is set
            if "flag" not in slot:
                slot["flag"] = torch.zeros((1,), dtype=torch.int32, pin_memory=True)
            slot["flag"].copy_(flag, non_blocking=True)
            ev.record()
            pend.flag_h, pend.fallback = slot["flag"], fallback
        slot["owner"] = weakref.ref(pend)
        return pend

    @torch.no_grad()
    def launch_lazy(self, dense, evaluator, tokens=None, fallback=None):
        """Lazy head: dense = list (one per task) of (B, 16, H, W) channels_last maps holding [iou] hm only;
        evaluator(local (S, pre_max) int64, seg_len (S,) int32, valid (S, pre_max) bool, segs = per task the rows of its lists) returns the
        (S, pre_max, 10) fp32 regression values [reg 2, height 1, dim 3, rot 2, vel 2] at the cells local = b*H*W + cell of each list's task
        (slots behind seg_len are ignored).  Same selection as launch(): scores come from the dense maps, the candidates are the
        first pre_max of every (sample, class) list; the centre range test runs on the evaluated candidates (pnx_decode_boxes_lazy) and
        `fallback()` (the dense path) is taken by result() in the one case where that could change the selection."""
        B = dense[0].shape[0]
        dev = dense[0].device
        dt = {torch.float32: PNX_F32, torch.bfloat16: PNX_BF16, torch.float16: PNX_F16}[dense[0].dtype]
        T = len(dense)
        shapes = [(p.shape[2], p.shape[3]) for p in dense]
        cfg = self.cfg
        descs, off = [], 0
        for t, (H, W) in enumerate(shapes):
            descs.append(pack_task(dense[t].shape[1], self.has_iou, self.num_classes[t], off, H, W, _get(cfg, "out_size_factor")[t],
                                   _get(cfg, "voxel_size"), _get(cfg, "pc_range"), _get(cfg, "score_threshold"),
                                   _get(cfg, "post_center_limit_range"), self.rectifier[t], lazy=True))
            off += self.num_classes[t]
        sizes = [B * h * w for h, w in shapes]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        keys = torch.empty((offs[-1],), dtype=torch.int64, device=dev)
        L = lib()
        for t, p in enumerate(dense):
            assert p.is_contiguous(memory_format=torch.channels_last), "dense head outputs must be channels_last"
            kp = ctypes.c_void_p(keys.data_ptr() + 8 * offs[t])
            check(L.pnx_decode_keys(ptr(p), dt, B, self.nc_total, descs[t], kp, stream_ptr()), "pnx_decode_keys")
        S = B * self.nc_total
        ck = ("lazy", B, T, dt, tuple(shapes), dev)
        if ck not in self._dev:
            tdesc = torch.frombuffer(bytearray(b"".join(descs)), dtype=torch.uint8).to(dev)
            koff = torch.tensor(offs, dtype=torch.int64, device=dev)
            seg_off = (torch.arange(S + 1, dtype=torch.int32, device=dev) * self.pre_max).contiguous()
            thr = torch.tensor(self.thr, dtype=torch.float32, device=dev).repeat(B)
            bounds = (torch.arange(S + 1, device=dev, dtype=torch.int64) << 32) ^ (-0x8000000000000000)
            seg_cls = torch.arange(S, device=dev) % self.nc_total
            segs, kofs, c0 = [], torch.zeros((S,), dtype=torch.int64, device=dev), 0
            for t in range(T):
                m = (seg_cls >= c0) & (seg_cls < c0 + self.num_classes[t])
                segs.append(torch.nonzero(m).flatten())
                kofs[m] = offs[t]
                c0 += self.num_classes[t]
            jj = torch.arange(self.pre_max, device=dev, dtype=torch.int64)
            self._dev[ck] = (tdesc, koff, seg_off, thr, bounds, segs, kofs, jj)
        tdesc, koff, seg_off, thr, bounds, segs, kofs, jj = self._dev[ck]
        if self.use_topk and self.pre_max <= 4096:
            n_rows0 = S * self.pre_max
            skeys = torch.empty((n_rows0,), dtype=torch.int64, device=dev)
            order = torch.zeros((n_rows0,), dtype=torch.int64, device=dev)     # slots behind seg_len are read (and ignored) by the evaluator's glue
            seg_start = torch.empty((S,), dtype=torch.int64, device=dev)
            seg_len = torch.empty((S,), dtype=torch.int32, device=dev)
            seg_total = torch.empty((S,), dtype=torch.int32, device=dev)
            wsb = int(L.pnx_decode_topk_workspace_bytes(offs[-1], S)) + 256
            if self._topk_ws is None or self._topk_ws.numel() < wsb or self._topk_ws.device != dev:
                self._topk_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            check(L.pnx_decode_topk(ptr(keys), offs[-1], S, self.pre_max, ptr(skeys), ptr(order), ptr(seg_start), ptr(seg_len), ptr(seg_total),
                                    ptr(self._topk_ws), self._topk_ws.numel(), stream_ptr()), "pnx_decode_topk")
            valid = jj[None, :] < seg_len[:, None]
            local = torch.where(valid, order.view(S, self.pre_max) - kofs[:, None], torch.zeros_like(kofs[:, None]))
        else:
            skeys = torch.empty_like(keys)
            order = torch.empty_like(keys)
            wsb = int(L.pnx_sort_keys_workspace_bytes(offs[-1]))
            if self._sort_ws is None or self._sort_ws.numel() < wsb or self._sort_ws.device != dev:
                self._sort_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            check(L.pnx_sort_keys(ptr(keys), offs[-1], S, ptr(skeys), ptr(order), ptr(self._sort_ws), self._sort_ws.numel(), stream_ptr()), "pnx_sort_keys")
            seg_start = torch.searchsorted(skeys ^ (-0x8000000000000000), bounds)
            seg_total = (seg_start[1:] - seg_start[:-1]).to(torch.int32)
            seg_len = torch.clamp(seg_total, max=self.pre_max)
            # candidate cells per slot (s, j): local index b*H*W + cell inside the slot's task
            pos = torch.clamp(seg_start[:S, None] + jj[None, :], max=offs[-1] - 1)
            valid = jj[None, :] < seg_len[:, None]
            local = order[pos] - kofs[:, None]
        n_rows = S * self.pre_max
        cand = evaluator(local, seg_len, valid, segs)                          # (S, pre_max, 10) fp32
        boxes9 = torch.empty((n_rows, 9), dtype=torch.float32, device=dev)
        boxes7 = torch.zeros((n_rows, 7), dtype=torch.float32, device=dev)
        scores = torch.empty((n_rows,), dtype=torch.float32, device=dev)
        flag = torch.zeros((1,), dtype=torch.int32, device=dev)
        check(L.pnx_decode_boxes_lazy(ptr(tdesc), ptr(koff), T, self.nc_total, ptr(skeys), ptr(order), ptr(seg_start), ptr(seg_len), ptr(seg_total), S,
                                      self.pre_max, ptr(cand), ptr(boxes9), ptr(boxes7), ptr(scores), ptr(flag), stream_ptr()), "pnx_decode_boxes_lazy")
        return self._finish(boxes9, boxes7, scores, seg_off, thr, seg_len, S, B, dev, tokens, flag=flag, fallback=fallback)

    @torch.no_grad()
    def launch_lazy_fused(self, dense, lazy_tasks, class_task, tokens=None, fallback=None):
        """launch_lazy with the HIP evaluator as ONE C call (include/pnx.h: pnx_decode_lazy_enqueue): keys per task, top-k, candidate cells,
        regression branches at the candidates, boxes, batched NMS, gather and the copies into pinned memory are enqueued from C++ out of a
        descriptor that is built once per (batch, map shapes); per step only the pointers of the dense maps, of the deblocked maps and of
        the pinned result slot change.  lazy_tasks: per task (up, wfrag1, bias1, w2c, bias2) as ops.sephead_lazy takes them.
        Every scratch buffer is persistent: the calls of consecutive steps are ordered by the stream."""
        B, dev, T = dense[0].shape[0], dense[0].device, len(dense)
        dt = {torch.float32: PNX_F32, torch.bfloat16: PNX_BF16, torch.float16: PNX_F16}[dense[0].dtype]
        shapes = [(p.shape[2], p.shape[3]) for p in dense]
        S = B * self.nc_total
        L = lib()
        ck = ("lazy_fused", B, T, dt, tuple(shapes), dev, torch.cuda.current_stream().cuda_stream)   # persistent scratch: one set per stream
        st = self._dev.get(ck)
        if st is None:
            cfg = self.cfg
            descs, off, offs = [], 0, [0]
            for t, (H, W) in enumerate(shapes):
                descs.append(pack_task(dense[t].shape[1], self.has_iou, self.num_classes[t], off, H, W, _get(cfg, "out_size_factor")[t],
                                       _get(cfg, "voxel_size"), _get(cfg, "pc_range"), _get(cfg, "score_threshold"),
                                       _get(cfg, "post_center_limit_range"), self.rectifier[t], lazy=True))
                off += self.num_classes[t]
                offs.append(offs[-1] + B * H * W)
            n_keys, n_rows = offs[-1], S * self.pre_max
            kofs = torch.tensor([offs[class_task[s % self.nc_total]] for s in range(S)], dtype=torch.int64, device=dev)
            i64, i32, f32 = torch.int64, torch.int32, torch.float32
            bufs = {
                "descs_host": ctypes.create_string_buffer(b"".join(descs)),
                "tdesc": torch.frombuffer(bytearray(b"".join(descs)), dtype=torch.uint8).to(dev),
                "koff_host": (ctypes.c_int64 * (T + 1))(*offs),
                "koff": torch.tensor(offs, dtype=i64, device=dev),
                "kofs": kofs,
                "seg_off": (torch.arange(S + 1, dtype=i32, device=dev) * self.pre_max).contiguous(),
                "thr": torch.tensor(self.thr, dtype=f32, device=dev).repeat(B),
                "dense_arr": (ctypes.c_void_p * T)(),
                "tasks_arr": (ops._PnxLazyTask * T)(),
                "class_task": (ctypes.c_int32 * self.nc_total)(*[int(v) for v in class_task]),
                "keys": torch.empty((n_keys,), dtype=i64, device=dev), "skeys": torch.empty((n_rows,), dtype=i64, device=dev),
                "order": torch.empty((n_rows,), dtype=i64, device=dev), "seg_start": torch.empty((S,), dtype=i64, device=dev),
                "local": torch.empty((n_rows,), dtype=i64, device=dev), "seg_len": torch.empty((S,), dtype=i32, device=dev),
                "seg_total": torch.empty((S,), dtype=i32, device=dev), "cand": torch.empty((n_rows, 10), dtype=f32, device=dev),
                "boxes9": torch.empty((n_rows, 9), dtype=f32, device=dev), "boxes7": torch.empty((n_rows, 7), dtype=f32, device=dev),
                "scores": torch.empty((n_rows,), dtype=f32, device=dev), "flag": torch.empty((1,), dtype=i32, device=dev),
                "keep": torch.empty((n_rows,), dtype=i32, device=dev), "cnt": torch.empty((S,), dtype=i32, device=dev),
                "out": torch.empty((S, self.post_max, 10), dtype=f32, device=dev),
            }
            bufs["topk_ws"] = torch.empty(int(L.pnx_decode_topk_workspace_bytes(n_keys, S)) + 256, dtype=torch.uint8, device=dev)
            bufs["nms_ws"] = torch.empty(max(int(L.pnx_nms_workspace_bytes(n_rows, S, self.pre_max)), 1), dtype=torch.uint8, device=dev)
            d = _PnxLazyDecode()
            d.n_tasks, d.n_classes_total, d.batch, d.pre_max, d.post_max, d.dtype = T, self.nc_total, B, self.pre_max, self.post_max, dt
            d.dense_host = ctypes.cast(bufs["dense_arr"], ctypes.c_void_p)
            d.task_descs_host = ctypes.cast(bufs["descs_host"], ctypes.c_void_p)
            d.task_key_off_host = ctypes.cast(bufs["koff_host"], ctypes.c_void_p)
            d.lazy_tasks_host = ctypes.cast(bufs["tasks_arr"], ctypes.c_void_p)
            d.class_task_host = ctypes.cast(bufs["class_task"], ctypes.c_void_p)
            for f, k in (("task_descs_dev", "tdesc"), ("task_key_off_dev", "koff"), ("list_key_off_dev", "kofs"), ("seg_off_dev", "seg_off"),
                         ("nms_thresh_dev", "thr"), ("keys", "keys"), ("sorted_keys", "skeys"), ("order", "order"), ("seg_start", "seg_start"),
                         ("local", "local"), ("seg_len", "seg_len"), ("seg_total", "seg_total"), ("cand", "cand"), ("boxes9", "boxes9"),
                         ("boxes7", "boxes7"), ("scores", "scores"), ("flag", "flag"), ("keep", "keep"), ("keep_count", "cnt"),
                         ("topk_ws", "topk_ws"), ("nms_ws", "nms_ws"), ("out", "out")):
                setattr(d, f, bufs[k].data_ptr())
            d.topk_ws_bytes, d.nms_ws_bytes = bufs["topk_ws"].numel(), bufs["nms_ws"].numel()
            st = self._dev[ck] = (d, bufs)
        d, bufs = st
        for t, (p, (up, wf, b1, w2c, b2)) in enumerate(zip(dense, lazy_tasks)):
            if not (p.is_contiguous(memory_format=torch.channels_last) and up.is_contiguous(memory_format=torch.channels_last)
                    and up.dtype == p.dtype and wf.dtype == p.dtype and up.shape[1] == 64 and up.shape[0] == B and tuple(up.shape[2:]) == shapes[t]):
                raise PnxError("lazy decode: dense maps, 64-channel deblocked maps and packed weights must be channels_last bf16 / fp16 (one dtype) of the task's shape")
            bufs["dense_arr"][t] = p.data_ptr()
            bufs["tasks_arr"][t] = ops._PnxLazyTask(up.data_ptr(), wf.data_ptr(), b1.data_ptr(), w2c.data_ptr(), b2.data_ptr(), up.shape[2], up.shape[3])
        slot = self._pinned_slot((S, self.post_max, 10), S, torch.float32, torch.int32)
        if "flag" not in slot:
            slot["flag"] = torch.zeros((1,), dtype=torch.int32, pin_memory=True)
        d.out_host, d.keep_count_host, d.flag_host = slot["out"].data_ptr(), slot["cnt"].data_ptr(), slot["flag"].data_ptr()
        check(L.pnx_decode_lazy_enqueue(ctypes.byref(d), stream_ptr()), "pnx_decode_lazy_enqueue")
        ev = torch.cuda.Event()
        ev.record()
        pend = PendingDetections(ev, slot["out"], slot["cnt"], B, self.nc_total, tokens if tokens else [None] * B)
        pend.flag_h, pend.fallback = slot["flag"], fallback
        slot["owner"] = weakref.ref(pend)
        return pend

    def _pinned_slot(self, out_shape, S, out_dtype, cnt_dtype):
        """One of the rotating pinned result buffers of a shape (a serving loop has at most two batches in flight: bench.py / forward_async);
        a slot is only reused once the PendingDetections that owns it was resolved (result()); otherwise the ring grows."""
        ring = self._pin.setdefault((tuple(out_shape), S), [])
        slot = next((sl for sl in ring if sl["owner"] is None or sl["owner"]() is None or sl["owner"]().done), None)
        if slot is None:
            slot = {"out": torch.empty(out_shape, dtype=out_dtype, pin_memory=True), "cnt": torch.empty((S,), dtype=cnt_dtype, pin_memory=True),
                    "owner": None}
            ring.append(slot)
        return slot

    def __call__(self, packed, tokens=None):
        return self.launch(packed, tokens).result()


class PendingDetections:
    """Detections of one frame batch on their way to the host.  Splitting launch() from result() lets a serving loop enqueue
    the next batch's GPU work before it blocks on this one (the GPU never idles on the host's post-processing)."""

    def __init__(self, event, out_h, cnt_h, batch, nc_total, tokens):
        self.event, self.out_h, self.cnt_h, self.batch, self.nc_total, self.tokens = event, out_h, cnt_h, batch, nc_total, tokens
        self.done = False  # the pinned buffers may be handed to a later launch once this is True (or this object is gone)
        self._res = None
        self.flag_h, self.fallback = None, None

    def result(self):
        if self._res is not None:
            return self._res
        self.event.synchronize()
        if self.flag_h is not None and int(self.flag_h[0]) != 0:
            if self.fallback is None:
                raise PnxError("lazy head: a candidate list cut at pre_max lost a candidate to the centre range test and no dense fallback "
                               "was supplied -- the selection may differ from CenterHead.post_processing (centerhead.py:341-363)")
            # a segment cut at pre_max lost a candidate to the centre range test: the dense path decides (exact, and rare)
            fb, self.fallback = self.fallback, None
            self._res, self.done = fb().result(), True
            return self._res
        self.fallback = None   # the closure holds every task's deblocked map (~1.2 GB at C2 x 12 frames): let the allocator have them back
        out_c, cnt_c = self.out_h, self.cnt_h.tolist()
        res = []
        for b in range(self.batch):
            bb, ss, ll = [], [], []
            for c in range(self.nc_total):
                k = cnt_c[b * self.nc_total + c]
                blk = out_c[b * self.nc_total + c, :k]
                bb.append(blk[:, :9])
                ss.append(blk[:, 9])
                ll.append(torch.full((k,), c, dtype=torch.int64))
            res.append({"box3d_lidar": torch.cat(bb), "scores": torch.cat(ss), "label_preds": torch.cat(ll), "token": self.tokens[b]})
        self._res, self.done = res, True  # torch.cat made copies: the pinned slot is free again
        return res
