#!/usr/bin/env python3
"""Micro-benchmark of the masked 3x3 conv kernel vs MIOpen (+ epilogue) on a backbone-shaped tensor."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=64); ap.add_argument("--cout", type=int, default=64)
ap.add_argument("--hw", type=int, default=1440); ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--density", type=float, default=1.0); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--miopen", action="store_true"); ap.add_argument("--res", action="store_true")
a = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((a.batch, a.cin, a.hw, a.hw), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn((a.cout, a.cin, 3, 3), device="cuda", generator=g) / 24).to(torch.bfloat16)
bias = torch.randn((a.cout,), device="cuda", generator=g)
mask = (torch.rand((a.batch, a.hw, a.hw), device="cuda", generator=g) < a.density).to(torch.uint8) if a.density < 1 else None
wf = ops.conv3x3_pack_weights(w)
res = (x[:, :a.cout] * 1.0).contiguous(memory_format=torch.channels_last) if a.res and a.cout <= a.cin else None
wcl = w.contiguous(memory_format=torch.channels_last)
torch.backends.cudnn.benchmark = True
def run():
    if a.miopen:
        y = torch.nn.functional.conv2d(x, wcl, None, 1, 1)
        return ops.bias_act_mask_(y, bias, mask, res, True)
    return ops.conv3x3_masked(x, wf, bias, a.cout, 1, mask, res, True)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
fl = 2.0 * a.batch * a.hw * a.hw * a.cin * a.cout * 9
from pillarnext_amd import _lib
import ctypes
L = _lib.lib() if hasattr(_lib, "lib") else None
if L is not None and hasattr(L, "pnx_debug_conv_timers") and not a.miopen:
    buf = (ctypes.c_ulonglong * 8)()
    L.pnx_debug_conv_timers(buf)          # reset
    run(); L.pnx_debug_conv_timers(buf)
    names = ["rowmask+sync", "deal rows/zero rows", "residual loads + stage issue+write", "stage barrier", "taps + epilogue", "-", "-", "tile head"]
    tot = sum(buf)
    print("  section share of wave time:", ", ".join(f"{n} {100.0*v/tot:.1f}%" for n, v in zip(names, buf) if v))
print(f"{'miopen+epilogue' if a.miopen else 'pnx_conv3x3'} {a.cin}->{a.cout} {a.hw}^2 b{a.batch} density {a.density}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s (dense-equivalent)")
