#!/usr/bin/env python3
"""Timing of the voxel and multi-view readers on one GPU, synthetic clouds, HIP events: the grouping calls (pnx_group_points: all of
its kernels + the one host sync for the two counts), the eval PFN stack and bilinear gather of a view, and MVFFeatureNet end to end at the
reference's Waymo MVF geometry (configs/models/reader/mvf_encoder.yaml: 0.075 m pillars over +-76.8 m = 2048 x 2048, cylinder cells 0.140625 deg x 0.2 m =
2560 x 100) with the per-view sparse ResNets as fp32 torch modules and on the masked HIP convolution kernels (use_hip_convs).
Algorithmic bytes of a grouping call = read every 24-byte row once + write the feature rows, unq_inv and coords once (no credit for the key /
bitmap / scan scratch).  usage: python tools/bench_readers.py [--points 180000] [--batch 2]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import ops, synth  # noqa: E402
from pillarnext_amd._lib import PNX_GROUP_CYLINDER_CLAMP, PNX_GROUP_PILLAR_CLAMP, PNX_GROUP_VOXEL  # noqa: E402
from pillarnext_amd.mvf_encoder import MVFFeatureNet  # noqa: E402


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=180_000)
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    B = a.batch
    pts = synth.make_batch("C4", B, "sweep", n=a.points)                    # Waymo-shaped sweep cloud, +-75.2 m
    t = torch.from_numpy(pts).cuda()
    n = t.shape[0]
    pr, vs = [-76.8, -76.8, -10.0, 76.8, 76.8, 10.0], [0.075, 0.075, 20]
    cr, cs = [-180, -10.0, 0, 180, 10.0, 107], [0.140625, 0.2, 107]
    print(f"# {n} points in {B} frames ({a.points} per frame), MI355X; wall time per call incl. the host sync for the counts")
    for name, g, cols in (("voxel 0.075 x 0.075 x 0.2 (drop)", ops.group_geom([-76.8, -76.8, -4.0, 76.8, 76.8, 4.0], [0.075, 0.075, 0.2], PNX_GROUP_VOXEL), 5),
                          ("pillar view 2048 x 2048 (clamp)", ops.group_geom(pr, vs, PNX_GROUP_PILLAR_CLAMP, pr), 10),
                          ("cylinder view 2560 x 100 (clamp)", ops.group_geom(cr, cs, PNX_GROUP_CYLINDER_CLAMP, pr), 10)):
        r = ops.group_points(t, B, g, want_mean=True)
        us = timed(lambda: ops.group_points(t, B, g, want_mean=True))
        by = 24 * n + r["Nk"] * (cols * 4 + 8) + r["G"] * (r["coords"].shape[1] * 4 + r["mean"].shape[1] * 4)
        print(f"pnx_group_points {name}: {us:7.1f} us   cells {r['G']}, kept {r['Nk']}, algorithmic {by / 1e6:.1f} MB -> {by / us / 1e3:.1f} GB/s")
    torch.manual_seed(0)
    m = MVFFeatureNet(in_channels=5, voxel_size=vs, pc_range=pr, cylinder_size=cs, cylinder_range=cr, num_filters=[48, 48], layer_nums=[2, 2, 2, 2],
                      ds_layer_strides=[1, 2, 2, 2], ds_num_filters=[48, 96, 192, 192], kernel_size=[3, 3, 3, 3], out_channels=256).cuda().eval()
    with torch.no_grad():
        feat, rp, rc, _ = m.group_views(t, B)
        us = timed(lambda: m.group_views(t, B))
        print(f"MVFFeatureNet.group_views (both groupings into one (N', 20) buffer): {us:7.1f} us")
        us = timed(lambda: m.pillarview.cell_features(feat, rp["unq_inv"], rp["G"]))
        print(f"SingleView PFN stack, eval (2 x pnx_pfn_layer_eval: 20 -> 24 (+) 24, 48 -> 48 + cell max): {us:7.1f} us  ({feat.shape[0]} points, {rp['G']} cells)")
        for label, dt in (("fp32 torch modules", None), ("HIP conv kernels bf16", torch.bfloat16), ("HIP conv kernels fp16", torch.float16)):
            m.use_hip_convs(dt)
            torch.cuda.reset_peak_memory_stats()
            ms = timed(lambda: m(t, batch_size=B), iters=5, warm=2) / 1e3
            print(f"MVFFeatureNet.forward, per-view sparse ResNets on {label}: {ms:8.2f} ms per {B} frames = {B / ms * 1e3:6.1f} frames/s, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
