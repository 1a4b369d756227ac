#!/usr/bin/env python3
"""Time one 64 -> 64 submanifold layer of backbone stage 0 on the C2 sweep cloud (8 frames): the sparse kernel (feature rows + occupancy
words, ops.subm64_sparse) against the dense-layout kernel (ops.conv3x3_masked with workspace + tile list), on the stage's real active set
(the 3 x 3 dilation of the pillar occupancy: its first layer is a SparseConv2d, sparse_resnet.py:53-54)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import ops, synth  # noqa: E402
from pillarnext_amd.reader import PillarFeatureNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dilate", type=int, default=1)
a = ap.parse_args()
cfg = synth.CONFIGS["C2"]
net = PillarFeatureNet(5, (64, 64), cfg["voxel_size"], cfg["pc_range"]).cuda().eval()
B = a.batch
pts = torch.from_numpy(synth.make_batch("C2", B, "sweep")).cuda()
ny, nx = (int(v) for v in net.grid_size)
occ = torch.empty((B, ny, nx), dtype=torch.uint8, device="cuda")
net.forward_dense(pts, B, occupancy=occ)
mask = ops.mask_pool3(occ, 1) if a.dilate else occ                                       # (B, ny, nx) canvas frame
mt = mask.transpose(1, 2).contiguous()
wfull, wpr = ops.sparse_index_from_mask(mt)
P = int(mask.sum())
g = torch.Generator(device="cuda").manual_seed(0)
rows = torch.relu(torch.randn((P, 64), device="cuda", generator=g)).to(torch.bfloat16)
res = torch.relu(torch.randn((P, 64), device="cuda", generator=g)).to(torch.bfloat16)
w = (torch.randn((64, 64, 3, 3), device="cuda", generator=g) / 24).to(torch.bfloat16)
bias = torch.randn((64,), device="cuda", generator=g)
wt = ops.conv3x3_pack_weights(w.transpose(2, 3))
out = torch.empty_like(rows)
tiles = ops.sparse_tile_list(wfull, B, nx, wpr)
print(f"active sites {P} ({P / mask.numel():.3f}), tiles listed {int(tiles[1])}")


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3


print(f"sparse  subm64            {timed(lambda: ops.subm64_sparse(rows, wfull, B, nx, wpr, wt, bias, out=out, tiles=tiles)):8.1f} us")
from pillarnext_amd._lib import lib  # noqa: E402
import ctypes  # noqa: E402

L = lib()
if hasattr(L, "pnx_debug_conv_timers"):   # instrumented build (PNX_CONV_TIMERS=1): share of wave time per section of the sparse kernel
    buf = (ctypes.c_ulonglong * 8)()
    L.pnx_debug_conv_timers(buf)
    ops.subm64_sparse(rows, wfull, B, nx, wpr, wt, bias, out=out, tiles=tiles)
    L.pnx_debug_conv_timers(buf)
    tot = float(sum(buf)) or 1.0
    names = ["barrier 1 -> ticket, next words", "locate + residual request", "staging (issue + LDS writes)", "barrier 2", "rest of the round loop", "taps (bias, residual add, MFMAs, pack)", "stores",
             "top of tile -> barrier 1"]
    for k in (7, 0, 1, 2, 3, 5, 6, 4):
        print(f"   section {names[k]:32s} {100 * buf[k] / tot:5.1f} %")
print(f"sparse  subm64 + residual {timed(lambda: ops.subm64_sparse(rows, wfull, B, nx, wpr, wt, bias, residual=res, out=out, tiles=tiles)):8.1f} us")
print(f"sparse  tile list         {timed(lambda: ops.sparse_tile_list(wfull, B, nx, wpr, out=tiles)):8.1f} us")
# dense-layout kernel on the same active set (rows are in rank order = (b, xi, yi) order = nonzero() order of the transposed mask)
b, xi, yi = torch.nonzero(mt, as_tuple=True)
xd = torch.zeros((B, ny, nx, 64), dtype=torch.bfloat16, device="cuda")
xd[b, yi, xi] = rows
xd = xd.permute(0, 3, 1, 2)
rd = torch.zeros((B, ny, nx, 64), dtype=torch.bfloat16, device="cuda")
rd[b, yi, xi] = res
rd = rd.permute(0, 3, 1, 2)
wf = ops.conv3x3_pack_weights(w)
ws = ops.conv3x3_workspace(B, 64, ny, nx, "cuda")
tl = ops.conv_tile_list(mask, [ws[1]], ops.conv_tile_rows(64, 64, 1))
for name, env in (("row kernel", "0"), ("pixel-gather kernel", "1")):
    os.environ["PNX_CONV_GATHER"] = env
    print(f"dense   {name:20s} {timed(lambda: ops.conv3x3_masked(xd, wf, bias, 64, 1, mask, None, True, out=ws, tiles=tl)):8.1f} us   "
          f"+ residual {timed(lambda: ops.conv3x3_masked(xd, wf, bias, 64, 1, mask, rd, True, out=ws, tiles=tl)):8.1f} us")
os.environ.pop("PNX_CONV_GATHER")
got = ops.subm64_sparse(rows, wfull, B, nx, wpr, wt, bias, residual=res, out=out, tiles=tiles)
ref = ops.conv3x3_masked(xd, wf, bias, 64, 1, mask, rd, True, out=ws, tiles=tl).permute(0, 2, 3, 1)[b, yi, xi]
d = (got.float() - ref.float()).abs()
print(f"sparse vs dense kernel: max |diff| {float(d.max()):.4f}, mean {float(d.mean()):.6f}, equal {float((d == 0).float().mean()):.4f}")
