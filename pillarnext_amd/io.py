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
        i = self.k % len(self.host)
        self.k += 1
        self.done[i].synchronize()                      # the pinned staging buffer of this slot is free again (its copy ran)
        n = sum(len(c) for c in clouds)
        collate_points(clouds, self.host[i].numpy())
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
        return self.dev[i][:n], len(clouds)
