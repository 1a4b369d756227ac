#!/usr/bin/env python3
"""A/B timing of the reader's pipelines on one GPU (points resident -> bf16 NHWC canvas + occupancy), HIP events from libpnx_hip.so.

Variants are environment settings read by pnx_reader_forward on every call:
  PNX_READER_IMPL=1            round-1 pipeline (global-atomic slots, 32-byte records, DPP-scan PFN, separate fill kernel)
  PNX_READER_IMPL=2            round-2 pipeline: binned grouping (reader_bins.h) + k_bin_sort + PFN v3 (pfn_v3.hip), records through HBM
  PNX_READER_IMPL=4            default: chunk sort (chunk_sort.hip) + span PFN (pfn_spans.hip)
  PNX_READER_IMPL=3            round 3: binned grouping + ONE launch that sorts every bin in LDS and runs the PFN on it (pfn_bins.hip)
    PNX_READER_FUSE=0|1|3      zero-fill as its own kernel | as extra blocks of the PFN launch and of the grouping kernels |
                               as a persistent grid-capped kernel on a second stream (PNX_FILL_SIDE percent, PNX_FILL_SIDE_BLOCKS)
    PNX_BIN_NWG, PNX_BIN_THREADS, PNX_BIN_SH   chunks / threads of k_bin_count and k_bin_scatter, pillars per bin (2^sh)
    PNX_FILL_SPLIT=a,b,c       percent of the fill tiles carried by k_bin_count / k_bin_scatter / k_bin_sort
    PNX_PFN_F16X3=0|1          layer 1 as fp32 MFMA | fp16 hi/lo splits
    PNX_FILL_BLOCKS, PNX_PFN_BLOCKS   block counts of the two roles
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
    ("r2 binned default", {"PNX_READER_IMPL": "2"}),
    ("lds default", {"PNX_READER_IMPL": "3"}),
    ("spans default", {"PNX_READER_IMPL": "4"}),
    ("spans pfn768", {"PNX_READER_IMPL": "4", "PNX_PFN_BLOCKS": "768"}),
    ("spans fill192", {"PNX_READER_IMPL": "4", "PNX_FILL_BLOCKS": "192"}),
    ("spans fill128", {"PNX_READER_IMPL": "4", "PNX_FILL_BLOCKS": "128"}),
    ("spans unfilled", {"PNX_READER_IMPL": "4", "PNX_FILL_BLOCKS": "0"}),
    ("spans pre10", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "640", "PNX_PREFILL": "10"}),
    ("spans pre15", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "640", "PNX_PREFILL": "15"}),
    ("spans pre20", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "640", "PNX_PREFILL": "20"}),
    ("spans pre25", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "640", "PNX_PREFILL": "25"}),
    ("spans pre30", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "640", "PNX_PREFILL": "30"}),
    ("spans pre20 b128", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "640", "PNX_PREFILL": "20", "PNX_PREFILL_BLOCKS": "128"}),
    ("spans pre100", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "640", "PNX_PREFILL": "100"}),
    ("spans side", {"PNX_READER_IMPL": "4", "PNX_FILL_SIDE": "1"}),
    ("spans side lds78k", {"PNX_READER_IMPL": "4", "PNX_FILL_SIDE": "1", "PNX_BINS_LDS": "78000"}),
    ("spans side lds76k", {"PNX_READER_IMPL": "4", "PNX_FILL_SIDE": "1", "PNX_BINS_LDS": "76000"}),
    ("spans side lds72k", {"PNX_READER_IMPL": "4", "PNX_FILL_SIDE": "1", "PNX_BINS_LDS": "72000"}),
    ("spans side lds76k b512", {"PNX_READER_IMPL": "4", "PNX_FILL_SIDE": "1", "PNX_BINS_LDS": "76000", "PNX_FILL_BLOCKS": "512"}),
    ("spans side lds76k b128", {"PNX_READER_IMPL": "4", "PNX_FILL_SIDE": "1", "PNX_BINS_LDS": "76000", "PNX_FILL_BLOCKS": "128"}),
    ("spans lds76k", {"PNX_READER_IMPL": "4", "PNX_BINS_LDS": "76000"}),
    ("spans q384 s128", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "384", "PNX_SPAN_SOLO": "128"}),
    ("spans q512 s256", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "512", "PNX_SPAN_SOLO": "256"}),
    ("spans q640", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "640"}),
    ("spans q768", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "768"}),
    ("spans q640 unfilled", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "640", "PNX_FILL_BLOCKS": "0"}),
    ("spans q384 s128 unfilled", {"PNX_READER_IMPL": "4", "PNX_SPAN_QUOTA": "384", "PNX_SPAN_SOLO": "128", "PNX_FILL_BLOCKS": "0"}),
    ("lds fill384", {"PNX_READER_IMPL": "3", "PNX_FILL_BLOCKS": "384"}),
    ("lds fill192", {"PNX_READER_IMPL": "3", "PNX_FILL_BLOCKS": "192"}),
    ("lds pfnblocks768", {"PNX_READER_IMPL": "3", "PNX_PFN_BLOCKS": "768"}),
    ("lds unfused", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "0"}),
    ("lds side100", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3"}),
    ("lds side100 b512", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE_BLOCKS": "512"}),
    ("lds side100 b128", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE_BLOCKS": "128"}),
    ("lds side60", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE": "60"}),
    ("lds side40", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE": "40"}),
    ("lds side25", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE": "25"}),
    ("lds side100 atpfn", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE_AT": "pfn"}),
    ("lds side100 atpfn b512", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE_AT": "pfn", "PNX_FILL_SIDE_BLOCKS": "512"}),
    ("lds side50 atpfn", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE_AT": "pfn", "PNX_FILL_SIDE": "50"}),
    ("lds sideat lds78k", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE_AT": "pfn", "PNX_BINS_LDS": "78000"}),
    ("lds sideat lds76k", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE_AT": "pfn", "PNX_BINS_LDS": "76000"}),
    ("lds sideat lds72k", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE_AT": "pfn", "PNX_BINS_LDS": "72000"}),
    ("lds sideat lds64k", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE_AT": "pfn", "PNX_BINS_LDS": "64000"}),
    ("lds sideat lds76k prio0", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE_AT": "pfn", "PNX_BINS_LDS": "76000", "PNX_FILL_PRIO": "0"}),
    ("lds sideat lds76k b512", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_FILL_SIDE_AT": "pfn", "PNX_BINS_LDS": "76000", "PNX_FILL_SIDE_BLOCKS": "512"}),
    ("lds sidebm lds76k", {"PNX_READER_IMPL": "3", "PNX_READER_FUSE": "3", "PNX_BINS_LDS": "76000"}),
    ("lds fused lds76k", {"PNX_READER_IMPL": "3", "PNX_BINS_LDS": "76000"}),
    ("lds nwg256 t512", {"PNX_READER_IMPL": "3", "PNX_BIN_NWG": "256", "PNX_BIN_THREADS": "512"}),
    ("lds nwg256 t1024", {"PNX_READER_IMPL": "3", "PNX_BIN_NWG": "256", "PNX_BIN_THREADS": "1024"}),
    ("lds nwg128 t512", {"PNX_READER_IMPL": "3", "PNX_BIN_NWG": "128", "PNX_BIN_THREADS": "512"}),
    ("lds sh9", {"PNX_READER_IMPL": "3", "PNX_BIN_SH": "9"}),
    ("lds split 10,10", {"PNX_READER_IMPL": "3", "PNX_FILL_SPLIT": "10,10,0"}),
    ("r2 binned unfused", {"PNX_READER_IMPL": "2", "PNX_READER_FUSE": "0"}),
    ("r2 side-stream fill", {"PNX_READER_IMPL": "2", "PNX_READER_FUSE": "2"}),
    ("round1", {"PNX_READER_IMPL": "1"}),
]
KEYS = ["PNX_READER_IMPL", "PNX_READER_FUSE", "PNX_FILL_BLOCKS", "PNX_PFN_BLOCKS", "PNX_FILL_SPLIT", "PNX_PFN_F16X3", "PNX_FILL_SIDE",
        "PNX_FILL_SIDE_BLOCKS", "PNX_FILL_SIDE_AT", "PNX_BIN_NWG", "PNX_BIN_THREADS", "PNX_BIN_SH", "PNX_BINS_CAP", "PNX_BINS_LDS", "PNX_FILL_PRIO", "PNX_SPAN_QUOTA", "PNX_SPAN_SOLO", "PNX_FILL_SIDE", "PNX_PREFILL", "PNX_PREFILL_BLOCKS"]


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
