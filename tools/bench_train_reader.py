#!/usr/bin/env python3
"""Training-mode reader step (forward with batch statistics + backward to the 6 PFN parameters), fused passes vs the torch-autograd
path, C2 geometry: time per step and peak extra memory."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import synth  # noqa: E402
from pillarnext_amd.reader import PillarFeatureNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = synth.CONFIGS["C2"]
tp = torch.from_numpy(synth.make_batch("C2", B, "sweep")).cuda()
for mode, name in (("1", "fused passes (csrc/pfn_train.hip)"), ("0", "torch autograd + HIP scatter-max (round 1)")):
    os.environ["PNX_TRAIN_FUSED"] = mode
    net = PillarFeatureNet(5, [64, 64], list(cfg["voxel_size"]), list(cfg["pc_range"])).cuda().train()
    w = torch.linspace(-1, 1, 64, device="cuda")

    def step():
        fm, _, _ = net(tp, B)
        (fm * w).sum().backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:46s} {B} frames ({tp.shape[0]} points): {dt * 1e3:8.2f} ms per forward+backward, peak extra memory "
          f"{(torch.cuda.max_memory_allocated() - base) / 2**20:8.0f} MiB", flush=True)
