#!/usr/bin/env python3
"""bench.py -- PillarNeXt-B inference frames/s on synthetic nuScenes-shaped clouds (BASELINE.json metric).

A step = one batch of `--batch` frames through the whole path with inputs already resident in HBM:
  HIP reader (voxelize + PFN + dense bf16 canvas)  ->  dense masked ResNet-18 + ASPP + CenterHead (PyTorch-ROCm,
  bf16, channels_last, MIOpen)  ->  decode + batched rotated NMS (HIP).
Rank 0 prints ONE JSON line (see the driver contract).  Extra objects:
  roofline      the reader's dominant kernel (dense-canvas writer), HBM-bound; algorithmic bytes per launch =
                (24*N + nx*ny*64*2) * frames per launch, duration from HIP events recorded on the
                kernel's own stream inside libpnx_hip.so (pnx_profile_begin/end) during the timed steps
  cpu_baseline  the CPU oracle ("port", one core) on a bounded sample of the same frames, hot path only
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PNX_BENCH_BATCH", "8")),
                    help="frames per GPU per step (measured on one MI355X: 4 -> 292, 8 -> 325, 16 -> 326 frames/s)")
    ap.add_argument("--config", default="C2")
    ap.add_argument("--dist", default="sweep", choices=["uniform", "sweep"],
                    help="sweep = ring-structured 10-sweep cloud (BASELINE configs[1]); uniform = worst case, ~1.2 points per pillar")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames of the bounded CPU-baseline sample (0 = skip)")
    return ap.parse_args()


def cpu_baseline(cfg, config, dist_name, frames):
    """CPU oracle (plain-C port of the reference algorithm, single thread) on `frames` frames: voxelize+PFN+scatter."""
    from oracle import oracle as O
    from pillarnext_amd import synth

    layers = synth.pfn_params()
    t_total = 0.0
    for f in range(frames):
        pts = synth.make_batch(config, 1, dist_name, frame0=f)
        t0 = time.perf_counter()
        O.reader_forward(pts, cfg["pc_range"], cfg["voxel_size"], [64, 64], layers, B=1, want_canvas=True)
        t_total += time.perf_counter() - t0
    return {"value": round(frames / t_total, 3), "unit": "frames/s (voxelize+PFN+scatter only, fp32 canvas)", "cores": 1, "kind": "port",
            "sample": f"{frames} frames of {config}/{dist_name}, oracle/pnx_oracle.c orc_reader_forward, 1 thread, host {os.cpu_count()} cores"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", init_method="env://")
    torch.cuda.set_device(local)
    if not os.environ.get("PNX_NO_MIOPEN_BENCH"):
        torch.backends.cudnn.benchmark = True  # MIOpen times its applicable solvers once per conv shape (during warm-up)
    dev = torch.device("cuda", local)

    from pillarnext_amd import _lib, synth
    from pillarnext_amd.models import build_pillarnext_b

    cfg = synth.CONFIGS[a.config]
    torch.manual_seed(0)
    model = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"]).to(dev).eval()
    if os.environ.get("PNX_BENCH_UNFUSED"):
        for m in (model.backbone, model.neck, model.head):
            m.to(memory_format=torch.channels_last, dtype=torch.bfloat16)
    else:
        from pillarnext_amd.models import FusedPillarNeXt

        model = FusedPillarNeXt(model).to(dev).eval()  # same network: eval-BN folded, HIP epilogues, merged head branches
    # frames are sharded across ranks: rank r gets frames r*B .. r*B+B-1 (replicas only, no collective on the path)
    pts = torch.from_numpy(synth.make_batch(a.config, a.batch, a.dist, frame0=rank * a.batch)).to(dev)
    example = {"points": pts, "token": [f"r{rank}f{i}" for i in range(a.batch)], "batch_size": a.batch}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(max(a.warmup, 1)):  # at least one untimed pass: library handles, MIOpen find mode, workspace allocation
            model(example)
        barrier()
        L = _lib.lib()
        _lib.check(L.pnx_profile_begin(max(a.steps, 1)), "pnx_profile_begin")
        t0 = time.perf_counter()
        if hasattr(model, "forward_async") and not os.environ.get("PNX_BENCH_SYNC"):
            # serving loop: batch i+1 is enqueued before the host blocks on (and unpacks) the detections of batch i; every
            # batch's detections are on the host, as dicts, before the timed region ends
            pending = None
            for _ in range(a.steps):
                nxt = model.forward_async(example)
                if pending is not None:
                    out = model.detections(pending.result())
                pending = nxt
            if pending is not None:
                out = model.detections(pending.result())
        else:
            for _ in range(a.steps):
                out = model(example)
        barrier()
        dt = time.perf_counter() - t0
    import ctypes

    r_us, c_us, ns = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int32(0)
    _lib.check(L.pnx_profile_end(ctypes.byref(r_us), ctypes.byref(c_us), ctypes.byref(ns)), "pnx_profile_end")

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    frames = a.batch * world * a.steps
    nx, ny = model.reader._geom.gx, model.reader._geom.gy
    pfn_us = float(L.pnx_profile_last_pfn_us())
    canvas_bytes = a.batch * nx * ny * 64 * 2
    frame_alg_bytes = 24 * pts.shape[0] + canvas_bytes           # 24*N + nx*ny*64*e per frame, x frames per launch
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    model.reader.forward_dense(pts, a.batch, counts=counts)
    P, n_kept = (int(v) for v in counts.tolist())
    # the HBM-bound kernel of the reader is the canvas zero-fill (k_canvas_fill_nhwc): it writes every cell that holds no pillar;
    # the P occupied cells (128 B each) are written by the PFN kernel's epilogue.  Its own algorithmic bytes per launch:
    fill_bytes = canvas_bytes - P * 128
    achieved = fill_bytes / (c_us.value * 1e-6) / 1e9 if c_us.value > 0 else None
    pfn_flops = 2.0 * n_kept * (10 * 32 + 64 * 64)              # 8 832 FLOP per kept point
    pfn_tf = pfn_flops / (pfn_us * 1e-6) / 1e12 if pfn_us > 0 else None
    traffic = None
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tj):
        try:
            traffic = json.load(open(tj)).get(f"{a.config}_b{a.batch}_{a.dist}", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    res = {
        "metric": "frames/s PillarNeXt-B nuScenes 300k-pt cloud (inference, end-to-end)", "value": round(frames / dt, 2), "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{a.config}: PillarNeXt-B nuScenes inference, {cfg['n']} pts/frame, voxel {cfg['voxel_size'][0]} m, BEV {nx}x{ny}, "
                               f"6 tasks/10 classes, cloud={a.dist}, random-init weights", "frames_per_gpu_per_step": a.batch,
                   "global_batch": a.batch * world, "parallelism": f"frame-sharded replicas x{world}", "reader_dtype": "fp32 MFMA PFN -> bf16 canvas",
                   "pillars_per_launch": P, "kept_points_per_launch": n_kept},
        "roofline": {"bound": "hbm", "kernel": "k_canvas_fill_nhwc<bf16> (writes every pillar-free cell of the dense BEV canvas; occupied cells "
                                               "come from the PFN epilogue)", "achieved": round(achieved, 1) if achieved else None,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": traffic,
                     "algorithmic_bytes_per_launch": fill_bytes, "kernel_us": round(c_us.value, 2), "samples": ns.value,
                     "reader_all_kernels_us": round(r_us.value, 2), "reader_algorithmic_bytes_per_launch": frame_alg_bytes,
                     "frac_all_reader_kernels": round(frame_alg_bytes / (r_us.value * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if r_us.value > 0 else None},
        "roofline_pfn": {"bound": "mfma", "kernel": "k_pfn_mfma<5,64> (+ k_pfn_big): fp32 v_mfma_f32_32x32x2_f32", "achieved": round(pfn_tf, 2) if pfn_tf else None,
                         "peak": 157.3, "unit": "TFLOP/s", "frac": round(pfn_tf / 157.3, 4) if pfn_tf else None, "kernel_us": round(pfn_us, 2),
                         "algorithmic_flops_per_launch": pfn_flops},
    }
    if rank == 0:
        if world == 1 and a.cpu_frames > 0:
            res["cpu_baseline"] = cpu_baseline(cfg, a.config, a.dist, a.cpu_frames)
        else:
            res["cpu_baseline"] = None
        res["detections_last_step"] = int(sum(len(v["scores"]) for v in out.values()))
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
