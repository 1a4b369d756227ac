#!/usr/bin/env python3
"""Micro-benchmark of the SepHead output convolution (k_sephead_out) on head-shaped input."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import ops
nb, B, H = 6, int(os.environ.get("B", "8")), 360
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.relu(torch.randn((B, nb * 64, H, H), device="cuda", generator=g)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
W2 = (torch.randn((16, nb * 64, 3, 3), device="cuda", generator=g) / 24).to(torch.bfloat16)
wf, bias = ops.sephead_pack_weights(W2), torch.zeros(16, device="cuda")
for _ in range(3): ops.sephead_out(x, wf, bias)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.sephead_out(x, wf, bias)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"sephead_out {nb}x64 -> 16 @ {H}^2 x{B}: {us:.1f} us  ({x.numel()*2/us/1e3:.0f} GB/s of input)")
