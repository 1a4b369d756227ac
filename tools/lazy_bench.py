#!/usr/bin/env python3
"""Time the lazy SepHead evaluator (csrc/conv3x3.hip::k_sephead_lazy) at the bench's size: 6 tasks, 8 frames, 360 x 360 maps, 10 classes,
pre_max 1000 full lists.  With an instrumented build (PNX_CONV_TIMERS=1) prints the share of wave time per section."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import ops  # noqa: E402
from pillarnext_amd._lib import lib  # noqa: E402

B, H, W, pre_max = 8, 360, 360, 1000
ncls = [1, 2, 2, 1, 2, 2]
g = torch.Generator(device="cuda").manual_seed(0)
tasks = []
for ti in range(6):
    up = torch.relu(torch.randn((B, 64, H, W), device="cuda", generator=g)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    W1 = (torch.randn((320, 64, 3, 3), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    w2m = (torch.randn((2880, 10), device="cuda", generator=g) * 0.05).to(torch.bfloat16).float()
    tasks.append((up, ops.conv3x3_pack_weights(W1), torch.zeros(320, device="cuda"), ops.sephead_lazy_pack_w2(w2m), torch.zeros(10, device="cuda")))
class_task = [t for t, n in enumerate(ncls) for _ in range(n)]
S = B * len(class_task)
local = torch.randint(0, B * H * W, (S, pre_max), device="cuda", generator=g)
seg_len = torch.full((S,), pre_max, dtype=torch.int32, device="cuda")
for _ in range(3):
    ops.sephead_lazy(tasks, class_task, B, local, seg_len, pre_max)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.sephead_lazy(tasks, class_task, B, local, seg_len, pre_max)
e1.record()
torch.cuda.synchronize()
print(f"k_sephead_lazy: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us for {S * pre_max} candidates ({S * ((pre_max + 31) // 32)} workgroups)")
L = lib()
if hasattr(L, "pnx_debug_conv_timers"):
    buf = (ctypes.c_ulonglong * 8)()
    L.pnx_debug_conv_timers(buf)
    ops.sephead_lazy(tasks, class_task, B, local, seg_len, pre_max)
    L.pnx_debug_conv_timers(buf)
    tot = float(sum(buf)) or 1.0
    names = ["cells + second-conv weights -> LDS", "patch gathers + LDS writes", "barrier", "taps (bias, fragments, MFMAs)", "per-pixel contraction",
             "partials -> LDS", "barrier", "output rows"]
    for k in range(8):
        print(f"   section {names[k]:38s} {100 * buf[k] / tot:5.1f} %")
