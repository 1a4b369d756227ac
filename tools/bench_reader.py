#!/usr/bin/env python3
"""Micro-benchmark of the reader (points -> dense bf16 canvas) on one GPU; run under rocprofv3 for per-kernel times."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import synth  # noqa: E402
from pillarnext_amd.reader import PillarFeatureNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--dist", default="uniform")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bfloat16")
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    net = PillarFeatureNet(5, [64, 64], list(cfg["voxel_size"]), list(cfg["pc_range"])).cuda().eval()
    pts = torch.from_numpy(synth.make_batch(a.config, a.batch, a.dist)).cuda()
    ny, nx = net.grid_size
    out = torch.empty((a.batch, 64, int(ny), int(nx)), dtype=getattr(torch, a.dtype), device="cuda", memory_format=torch.channels_last)
    counts = torch.zeros(2, dtype=torch.int32, device="cuda")
    for _ in range(a.warmup):
        net.forward_dense(pts, a.batch, out=out, counts=counts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import ctypes
    from pillarnext_amd import _lib
    L = _lib.lib()
    L.pnx_profile_begin(a.iters)
    e0.record()
    for _ in range(a.iters):
        net.forward_dense(pts, a.batch, out=out, counts=counts)
    e1.record()
    torch.cuda.synchronize()
    r_us, c_us, ns = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int32(0)
    os.environ.setdefault("PNX_DEBUG", "1")
    L.pnx_profile_end(ctypes.byref(r_us), ctypes.byref(c_us), ctypes.byref(ns))
    print(f"events: reader {r_us.value:.1f} us, canvas kernel {c_us.value:.1f} us")
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    P, m = counts.tolist()
    esz = out.element_size()
    alg = 24 * pts.shape[0] + out.numel() * esz
    print(f"{a.config} {a.dist} B={a.batch} N={pts.shape[0]} N'={m} P={P}: {us:.1f} us/call  algorithmic {alg/1e6:.1f} MB -> {alg/us/1e6:.2f} TB/s "
          f"({alg/us/1e6/8*100:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
