"""Input contract of the hot path.

The reference collates a batch on the CPU -- every key containing 'point' gets the sample index prepended as a float column and
the samples are concatenated (det3d/datasets/loader/collate.py:15-22) -- and then uploads it with a blocking copy
(trainer/trainer/trainer.py:8-19,111).  Here the collation writes straight into ONE pinned staging buffer and the upload is an
async copy on a side stream, double-buffered, so frame k+1 travels over PCIe while frame k is on the GPU."""
import numpy as np
import torch


def collate_points(clouds, out=None):
    """clouds: list of (n_i, F) fp32 arrays -> (sum n_i, 1+F) fp32 with the batch index in column 0 (collate.py:15-22)."""
    n = sum(len(c) for c in clouds)
    F = clouds[0].shape[1]
    if out is None:
        out = np.empty((n, 1 + F), np.float32)
    o = 0
    for b, c in enumerate(clouds):
        out[o:o + len(c), 0] = b
        out[o:o + len(c), 1:] = c
        o += len(c)
    return out[:n]


class PointUploader:
    """Double-buffered pinned staging + async H2D of collated point batches."""

    def __init__(self, max_points, num_features=5, device="cuda", depth=2):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.host = [torch.empty((max_points, 1 + num_features), dtype=torch.float32).pin_memory() for _ in range(depth)]
        self.dev = [torch.empty((max_points, 1 + num_features), dtype=torch.float32, device=self.device) for _ in range(depth)]
        self.done = [torch.cuda.Event() for _ in range(depth)]      # slot's H2D copy finished (host buffer reusable)
        self.marks = [None] * (depth + 1)                           # compute-stream events of the last depth+1 calls
        self.k = 0

    def upload(self, clouds):
        """Returns (device tensor view (N,1+F), batch size); the copy is ordered before later work on the CURRENT stream."""
        return self._go(sum(len(c) for c in clouds), lambda buf: collate_points(clouds, buf)), len(clouds)

    def upload_rows(self, rows):
        """rows: (n, 1+F) fp32 host array (e.g. the raw sweeps of a batch, back to back) -> device view of the same rows."""
        def fill(buf):
            buf[:len(rows)] = rows
        return self._go(len(rows), fill)

    def _go(self, n, fill):
        i = self.k % len(self.host)
        self.k += 1
        self.done[i].synchronize()                      # the pinned staging buffer of this slot is free again (its copy ran)
        fill(self.host[i].numpy())
        # The DEVICE buffer of the slot may still be read by reader kernels of the batch that used it `depth` uploads ago.  Every call
        # records an event on the compute stream first: the event of call j covers the consumers of batches < j, so the consumers of
        # batch j - depth are covered by the event of call j - depth + 1 -- waiting for THAT one (not for "now") keeps the copy of
        # batch j overlapped with the compute of batch j - 1.
        j = self.k - 1
        mark = torch.cuda.Event()
        mark.record(torch.cuda.current_stream(self.device))
        self.marks[j % len(self.marks)] = mark
        need = j - len(self.host) + 1
        if need >= 0 and self.marks[need % len(self.marks)] is not None:
            self.stream.wait_event(self.marks[need % len(self.marks)])
        with torch.cuda.stream(self.stream):
            self.dev[i][:n].copy_(self.host[i][:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        torch.cuda.current_stream(self.device).wait_event(ev)
        self.done[i] = ev
        return self.dev[i][:n]


# --------------------------------------------------------------------------------------------- multi-sweep merge on the device
def pack_segments(segments):
    """segments: list of dicts {begin, end, batch, time, radius=0.0, transform=None (4x4 or 3x4 array)} in row order ->
    bytes matching `struct SegDesc` of csrc/merge.hip."""
    import struct

    blob = b""
    for s in segments:
        T = s.get("transform")
        t12 = [0.0] * 12 if T is None else [float(v) for v in np.asarray(T, np.float64)[:3, :4].reshape(-1)]
        blob += struct.pack("12d2q2f2i", *t12, int(s["begin"]), int(s["end"]), float(s.get("radius", 0.0)), float(s["time"]), int(s["batch"]),
                            0 if T is None else 1)
    return blob


class SweepMerger:
    """nusc.py:76-121 / waymo.py:49-67 + collate.py:15-22 on the GPU: raw sweeps (already uploaded, one (M, C) fp32 tensor) ->
    the collated (N, 2+n_copy) point buffer of the reader, in one stable compaction (csrc/merge.hip)."""

    def __init__(self):
        self._ws = None

    def descriptors(self, segments, device):
        """Segment descriptors on the device (build once per segment layout, reuse every step)."""
        from . import _lib

        blob = pack_segments(segments)
        assert len(blob) == len(segments) * _lib.lib().pnx_merge_sweeps_desc_bytes(), "SegDesc layout drifted"
        return torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device), len(segments)

    def __call__(self, raw, segments, n_copy=4, out=None, n_out=None):
        """segments: list of dicts (pack_segments) or the (tensor, count) pair of descriptors().  Returns (out (n, n_copy+2), n_out (1,) int32):
        rows [0, n_out) are the merged cloud, rows [n_out, n) carry batch index -1 (the reader drops them), so `out` can be handed to
        the reader as a whole without a host sync."""
        from . import _lib

        L = _lib.lib()
        assert raw.is_cuda and raw.dtype == torch.float32 and raw.is_contiguous()
        n, stride = raw.shape
        desc, nseg = segments if isinstance(segments, tuple) else self.descriptors(segments, raw.device)
        need = int(L.pnx_merge_sweeps_workspace_bytes(n)) + 256
        if self._ws is None or self._ws.numel() < need or self._ws.device != raw.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=raw.device)
        if out is None:
            out = torch.empty((n, n_copy + 2), dtype=torch.float32, device=raw.device)
        assert out.shape[0] >= n and out.shape[1] == n_copy + 2 and out.is_contiguous()
        if n_out is None:
            n_out = torch.zeros(1, dtype=torch.int32, device=raw.device)
        _lib.check(L.pnx_merge_sweeps(_lib.ptr(raw), n, stride, n_copy, _lib.ptr(desc), nseg, _lib.ptr(out), _lib.ptr(n_out), _lib.ptr(self._ws),
                                      self._ws.numel(), _lib.stream_ptr()), "pnx_merge_sweeps")
        return out[:n], n_out
