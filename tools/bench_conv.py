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
ap.add_argument("--stride", type=int, default=1); ap.add_argument("--tiles", action="store_true"); ap.add_argument("--miopen", action="store_true"); ap.add_argument("--res", action="store_true")
ap.add_argument("--dilate", action="store_true", help="with --lidar: the stage's active set after its entry SparseConv2d (3x3 dilation of the pooled occupancy) -- what the stage's blocks run on")
ap.add_argument("--data", default="randn", help="randn | relu (post-ReLU: half the values zero) | zero -- operand toggling sets the power draw and with it the clock")
ap.add_argument("--lidar", type=int, default=-1, help="backbone stage (0..3): active sites = the C2 sweep occupancy pooled to that stage (sets --hw)")
a = ap.parse_args()
lidar_mask = None
if a.lidar >= 0:
    import numpy as np
    from pillarnext_amd import synth
    cfg = synth.CONFIGS["C2"]
    r, v = cfg["pc_range"], cfg["voxel_size"]
    pts = torch.from_numpy(synth.make_batch("C2", a.batch, "sweep")).cuda()
    nx, ny = int(round((r[3] - r[0]) / v[0])), int(round((r[4] - r[1]) / v[1]))
    bi, xi, yi = pts[:, 0].long(), ((pts[:, 1] - r[0]) / v[0]).floor().long(), ((pts[:, 2] - r[1]) / v[1]).floor().long()
    ok = (xi >= 0) & (xi < nx) & (yi >= 0) & (yi < ny)
    occ = torch.zeros((a.batch, ny, nx), dtype=torch.uint8, device="cuda")
    occ[bi[ok], yi[ok], xi[ok]] = 1
    occ_in = occ
    for _ in range(a.lidar):
        occ_in, occ = occ, ops.mask_pool3(occ, 2)
    if a.dilate:
        occ = ops.mask_pool3(occ, 1)
    lidar_mask, a.hw = occ, occ.shape[-1]
    if a.stride == 1:
        occ_in = occ
    seg = torch.nn.functional.max_pool1d(occ.float(), 32, 32, ceil_mode=True)
    print(f"stage {a.lidar}: {a.hw}^2, active sites {occ.float().mean().item():.3f}, active 32-px row segments {seg.mean().item():.3f}")
g = torch.Generator(device="cuda").manual_seed(0)
hw_in = a.hw * a.stride
x = torch.randn((a.batch, a.cin, hw_in, hw_in), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
if a.data == "relu":
    x = torch.relu(x)
elif a.data == "zero":
    x = torch.zeros_like(x)
w = (torch.randn((a.cout, a.cin, 3, 3), device="cuda", generator=g) / 24).to(torch.bfloat16)
bias = torch.randn((a.cout,), device="cuda", generator=g)
mask = (torch.rand((a.batch, a.hw, a.hw), device="cuda", generator=g) < a.density).to(torch.uint8) if a.density < 1 else None
if lidar_mask is not None:
    mask = lidar_mask
    x = (x * occ_in.unsqueeze(1)).contiguous(memory_format=torch.channels_last)
ws = ops.conv3x3_workspace(a.batch, a.cout, a.hw, a.hw, "cuda") if mask is not None and os.environ.get("PNX_BENCH_WS", "1") != "0" else None
wf = ops.conv3x3_pack_weights(w)
res = (x[:, :a.cout] * 1.0).contiguous(memory_format=torch.channels_last) if a.res and a.cout <= a.cin else None
wcl = w.contiguous(memory_format=torch.channels_last)
torch.backends.cudnn.benchmark = True
tiles = None
if a.tiles and ws is not None and a.stride == 1:
    tiles = ops.conv_tile_list(mask, [ws[1]], ops.conv_tile_rows(a.cin, a.cout, 1))
    print(f"tile list: {int(tiles[1].item())} of {tiles[0].numel()} tiles")
def run():
    if a.miopen:
        y = torch.nn.functional.conv2d(x, wcl, None, a.stride, 1)
        return ops.bias_act_mask_(y, bias, mask, res, True)
    return ops.conv3x3_masked(x, wf, bias, a.cout, a.stride, mask, res, True, out=ws, tiles=tiles)
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
    if os.environ.get("PNX_CONV_PC", "1") != "0" and a.stride == 1:   # producer / consumer kernel (csrc/conv_pc.h): 8 consumer + 4 producer waves
        names = ["c:info+deal", "c:taps", "c:barrier", "c:epilogue", "-", "p:prepare+issue", "p:wait+barrier", "prologue"]
        ctot, ptot = sum(buf[0:4]), sum(buf[5:7])
        print("  consumer waves:", ", ".join(f"{n} {100.0*v/ctot:.1f}%" for n, v in zip(names[:4], buf[:4]) if v), "| producer waves:",
              ", ".join(f"{n} {100.0*v/ptot:.1f}%" for n, v in zip(names[5:7], buf[5:7]) if v), f"| consumer ticks per launch {ctot/8:.0f} per wave-slot x 256 CUs")
    else:
        names = ["rowmask+sync", "deal rows/zero rows", "residual loads + stage issue+write", "stage barrier", "taps + epilogue", "-", "-", "tile head"]
        tot = sum(buf)
        print("  section share of wave time:", ", ".join(f"{n} {100.0*v/tot:.1f}%" for n, v in zip(names, buf) if v))
print(f"{'miopen+epilogue' if a.miopen else 'pnx_conv3x3'} {a.cin}->{a.cout} {a.hw}^2 (stride {a.stride}) b{a.batch} density {a.density}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s (dense-equivalent)")
