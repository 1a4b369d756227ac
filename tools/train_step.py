#!/usr/bin/env python3
"""Data-parallel training step of PillarNeXt-B on synthetic frames + labels (one process per GPU, torchrun / RCCL).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_step.py --batch 4

Mirrors tools/train.py:26-67 + trainer/trainer/trainer.py:94-108 of the reference: SyncBatchNorm conversion, DDP wrap
(bucketed gradient all-reduce overlapped with backward), AdamW(0.9,0.99,wd .01, configs/optimizer/adamW.yaml), OneCycleLR(max_lr .002,
div_factor 10, pct_start .4, configs/scheduler/onecycle.yaml) stepped every iteration, clip 35.  --amp runs the dense graph under bf16
autocast in channels_last -- narrower arithmetic than the reference, which trains in fp32 throughout (no autocast anywhere in its tree);
--nhwc is the fp32 step in channels_last.  The masked BatchNorm + ReLU + mask of every sparse block is one recomputing autograd node
(models.masked_bn_act) either way; under --amp the backbone's 3x3 layers run on the masked HIP kernels (models.masked_conv).
--yaml configs/pillarnext_b_waymo.yaml builds the Waymo detector (2 tasks, iou head -> IouLoss on the fused loss kernel,
configs/experiments/waymo_det_pp18_aspp_iou_car_sp_f1.yaml) at the geometry of --config (C4 / C5)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import dist_utils, synth  # noqa: E402
from pillarnext_amd.models import NUSC_TASKS, build_pillarnext_b  # noqa: E402


def synthetic_labels(tasks, B, H, W, M, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    ex = {k: [] for k in ("hm", "ind", "mask", "cat", "anno_box", "gt_boxes")}
    for names in tasks:
        ex["hm"].append(torch.rand((B, len(names), H, W), device=dev, generator=g) * 0.2)
        ex["ind"].append(torch.randint(0, H * W, (B, M), device=dev, generator=g))
        m = torch.zeros((B, M), dtype=torch.uint8, device=dev)
        m[:, : M // 8] = 1
        ex["mask"].append(m)
        ex["cat"].append(torch.randint(0, len(names), (B, M), device=dev, generator=g))
        ex["anno_box"].append(torch.randn((B, M, 10), device=dev, generator=g) * 0.3)
        ex["gt_boxes"].append(torch.rand((B, M, 7), device=dev, generator=g) + torch.tensor([0, 0, -1, 1.5, 0.6, 1.2, 0], device=dev))
    return ex


def main():
    # MIOpen's find step: the naive reference solvers take 0.5 s per call on these shapes (step 0: 250 s instead of 11 s) and never win
    for k in ("FWD", "BWD", "WRW"):
        os.environ.setdefault(f"MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_{k}", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--check", action="store_true", help="assert finite loss / gradients and print the peak device memory")
    ap.add_argument("--amp", action="store_true", help="bf16 autocast + channels_last for the dense backbone / neck / head")
    ap.add_argument("--nhwc", action="store_true", help="channels_last without autocast (fp32): the fused masked-BatchNorm kernels need NHWC maps")
    ap.add_argument("--total-steps", type=int, default=1000, help="length of the OneCycle schedule the steps are taken from")
    ap.add_argument("--find", action="store_true", help="let MIOpen time its solvers per conv problem (cudnn.benchmark): ~2 minutes in step 0, a step about a tenth faster")
    ap.add_argument("--yaml", default="", help="build the detector from this YAML (configs/pillarnext_b_waymo.yaml) instead of the nuScenes PillarNeXt-B")
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = bool(a.find)
    rank, world, local = dist_utils.init()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = synth.CONFIGS[a.config]
    torch.manual_seed(0)
    tasks = NUSC_TASKS
    if a.yaml:
        from pillarnext_amd import config as C

        y = C.load(a.yaml)
        y["model"]["reader"]["voxel_size"], y["model"]["reader"]["pc_range"] = list(cfg["voxel_size"]), list(cfg["pc_range"])
        for blk in ("head", "post_processing"):
            y["model"][blk]["voxel_size"], y["model"][blk]["pc_range"] = list(cfg["voxel_size"]), list(cfg["pc_range"])
        model = C.instantiate(y["model"]).to(dev).train()
        tasks = [list(t) for t in y["model"]["head"]["tasks"]]
        assert model.head.with_iou, "the Waymo YAML carries the iou head"
    else:
        model = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"]).to(dev).train()
    if a.amp or a.nhwc:
        model = model.to(memory_format=torch.channels_last)
    model = dist_utils.wrap_ddp(model, device_ids=[local])
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.01)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=0.002, total_steps=max(a.total_steps, a.steps + 1), div_factor=10.0, pct_start=0.4)
    pts = torch.from_numpy(synth.make_batch(a.config, a.batch, "sweep", frame0=rank * a.batch)).to(dev)
    net = model.module if hasattr(model, "module") else model
    ny, nx = (int(v) for v in net.reader.grid_size)
    ex = synthetic_labels(tasks, a.batch, ny // 4, nx // 4, 500, dev, 100 + rank)
    ex.update(points=pts, batch_size=a.batch)
    for it in range(a.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.amp):
            loss, _ = model(ex)
        opt.zero_grad()
        loss.backward()                                   # DDP: bucketed all-reduce over RCCL overlapped with backward
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 35)
        if a.check:
            assert bool(torch.isfinite(loss)) and bool(torch.isfinite(gn)) and float(gn) > 0, (float(loss), float(gn))
        opt.step()
        sched.step()
        torch.cuda.synchronize()
        dt = dist_utils.max_over_ranks(time.perf_counter() - t0, dev)
        if rank == 0:
            print(f"step {it}: loss {loss.item():.4f}  {dt*1e3:.1f} ms  ({a.batch * world / dt:.1f} frames/s over {world} GPU)")
    if rank == 0 and a.check:
        print(f"peak device memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB allocated, {torch.cuda.max_memory_reserved() / 2**30:.2f} GiB reserved "
              f"({a.config}, {a.batch} frames per GPU, {'bf16 autocast' if a.amp else 'fp32'} training)")


if __name__ == "__main__":
    main()
