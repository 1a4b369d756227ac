#!/usr/bin/env python3
"""A/B timing of the reader's pipelines on one GPU (points resident -> bf16 NHWC canvas + occupancy), HIP events from libpnx_hip.so.

Variants are environment settings read by pnx_reader_forward on every call:
  PNX_READER_IMPL=4            default: chunk sort (chunk_sort.hip) + span PFN (pfn_spans.hip)
    PNX_SPAN_QUOTA, PNX_SPAN_SOLO     carve rule of the spans (spans.h)
    PNX_BINS_LDS, PNX_BINS_CAP        LDS budget / record slots of a span workgroup
    PNX_FILL_BLOCKS, PNX_PFN_BLOCKS   workgroups of the zero-fill kernel (second stream) and of the span kernel (0 fill blocks: timing only,
                                      the canvas is wrong); PNX_FILL_NT=0|1 plain / nontemporal fill stores
  PNX_READER_IMPL=2            the general pipeline: binned grouping (reader_bins.h) + k_bin_sort + PFN v3 (pfn_v3.hip), records through HBM
    PNX_FILL_SPLIT=a,b,c       percent of the fill tiles carried by k_bin_count / k_bin_scatter / k_bin_sort
  PNX_PFN_F16X3=0|1            layer 1 as fp32 MFMA | fp16 hi/lo splits (0 sends the call to the general pipeline)
Every variant must equal the first variant's canvas bit for bit.
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import _lib, synth  # noqa: E402
from pillarnext_amd.reader import PillarFeatureNet  # noqa: E402

VARIANTS = [
    ("spans default", {}),
    ("spans default again", {}),
    ("spans default 4", {}),
    ("spans q512", {"PNX_SPAN_QUOTA": "512"}),
    ("spans q704", {"PNX_SPAN_QUOTA": "704"}),
    ("spans q768", {"PNX_SPAN_QUOTA": "768"}),
    ("spans q832", {"PNX_SPAN_QUOTA": "832"}),
    ("spans q896", {"PNX_SPAN_QUOTA": "896"}),
    ("spans q1024", {"PNX_SPAN_QUOTA": "1024"}),
    ("spans q768 pfn768", {"PNX_SPAN_QUOTA": "768", "PNX_PFN_BLOCKS": "768"}),
    ("spans q768 lds76k", {"PNX_SPAN_QUOTA": "768", "PNX_BINS_LDS": "76000"}),
    ("spans q896 pfn768", {"PNX_SPAN_QUOTA": "896", "PNX_PFN_BLOCKS": "768"}),
    ("spans pfn640", {"PNX_PFN_BLOCKS": "640"}),
    ("spans pfn1024", {"PNX_PFN_BLOCKS": "1024"}),
    ("spans default 3", {}),
    ("spans q384 s128", {"PNX_SPAN_QUOTA": "384", "PNX_SPAN_SOLO": "128"}),
    ("spans pfn768", {"PNX_PFN_BLOCKS": "768"}),
    ("spans fill192", {"PNX_FILL_BLOCKS": "192"}),
    ("spans fill128", {"PNX_FILL_BLOCKS": "128"}),
    ("spans lds76k", {"PNX_BINS_LDS": "76000"}),
    ("spans seg96", {"PNX_BINS_CAP": "96"}),
    ("spans unfilled (timing only)", {"PNX_FILL_BLOCKS": "0"}),
    ("spans lds78000", {"PNX_BINS_LDS": "78000"}),
    ("spans lds81000", {"PNX_BINS_LDS": "81000"}),
    ("spans lds81400", {"PNX_BINS_LDS": "81400"}),
    ("spans fill nt off", {"PNX_FILL_NT": "0"}),
    ("binned", {"PNX_READER_IMPL": "2"}),
    ("binned split 10,10", {"PNX_READER_IMPL": "2", "PNX_FILL_SPLIT": "10,10,0"}),
    ("fp32 layer 1 (binned)", {"PNX_PFN_F16X3": "0"}),
]
KEYS = ["PNX_READER_IMPL", "PNX_FILL_BLOCKS", "PNX_PFN_BLOCKS", "PNX_FILL_SPLIT", "PNX_PFN_F16X3", "PNX_BINS_CAP", "PNX_BINS_LDS", "PNX_SPAN_QUOTA", "PNX_SPAN_SOLO", "PNX_FILL_NT"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--dist", default="sweep")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--only", default="", help="run the variants whose name contains this string")
    ap.add_argument("--exact", default="", help="run exactly this variant")
    ap.add_argument("--no-occ", action="store_true", help="do not ask for the occupancy output")
    ap.add_argument("--between", default="", choices=["", "mfma", "dirty", "both", "idle"],
                    help="what the GPU does between two timed reader calls, as the detector's convolutions do in bench.py: mfma = ~12 ms of bf16 "
                         "GEMMs (power state), dirty = 1 GiB of plain stores (dirty lines in L2 / Infinity Cache), idle = a 15 ms host sleep")
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    net = PillarFeatureNet(5, [64, 64], list(cfg["voxel_size"]), list(cfg["pc_range"])).cuda().eval()
    # four different batches rotate through the loop, as in bench.py
    batches = [torch.from_numpy(synth.make_batch(a.config, a.batch, a.dist, frame0=k * a.batch)).cuda() for k in range(4)]
    ny, nx = (int(v) for v in net.grid_size)
    out = torch.empty((a.batch, 64, ny, nx), dtype=torch.bfloat16, device="cuda", memory_format=torch.channels_last)
    occ = None if a.no_occ else torch.empty((a.batch, ny, nx), dtype=torch.uint8, device="cuda")
    counts = torch.zeros(2, dtype=torch.int32, device="cuda")
    L = _lib.lib()
    ref = None
    if a.between in ("mfma", "both"):
        ga = torch.randn((8192, 8192), device="cuda").to(torch.bfloat16)
        gb, gc = ga.clone(), torch.empty_like(ga)
    if a.between in ("dirty", "both"):
        junk = torch.zeros((1 << 28,), dtype=torch.int32, device="cuda")
    alg = 24 * batches[0].shape[0] + out.numel() * 2
    print(f"# {a.config} {a.dist} B={a.batch}: algorithmic {alg/1e6:.1f} MB per launch (24*N + canvas)")
    for name, env in VARIANTS:
        if (a.only and a.only not in name) or (a.exact and a.exact != name):
            continue
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        for i in range(a.warmup):
            net.forward_dense(batches[i % 4], a.batch, out=out, counts=counts, occupancy=occ)
        net.forward_dense(batches[0], a.batch, out=out, counts=counts, occupancy=occ)
        torch.cuda.synchronize()
        if ref is None:
            ref_canvas, ref_occ = out.clone(), (occ.clone() if occ is not None else None)
            ref = True
        same = torch.equal(out, ref_canvas) and (occ is None or torch.equal(occ, ref_occ))
        L.pnx_profile_begin(a.iters)
        for i in range(a.iters):
            if a.between in ("mfma", "both"):
                for _ in range(12):
                    torch.mm(ga, gb, out=gc)
            if a.between in ("dirty", "both"):
                junk.add_(1)
            if a.between == "idle":
                torch.cuda.synchronize()
                import time
                time.sleep(0.015)
            net.forward_dense(batches[i % 4], a.batch, out=out, counts=counts, occupancy=occ)
        torch.cuda.synchronize()
        r_us, c_us, ns = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int32(0)
        L.pnx_profile_end(ctypes.byref(r_us), ctypes.byref(c_us), ctypes.byref(ns))
        P, m = counts.tolist()
        print(f"{name:26s} reader {r_us.value:8.1f} us  voxelize {L.pnx_profile_last_voxelize_us():7.1f}  pfn {L.pnx_profile_last_pfn_us():7.1f}  "
              f"canvas {c_us.value:7.1f}  -> {alg / r_us.value / 1e6:5.2f} TB/s ({alg / r_us.value / 1e6 / 8 * 100:4.1f}% of 8)  P={P} N'={m}  "
              f"{'== first variant' if same else 'MISMATCH vs first variant'}", flush=True)


if __name__ == "__main__":
    main()
