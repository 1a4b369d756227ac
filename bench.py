#!/usr/bin/env python3
"""bench.py -- PillarNeXt-B inference frames/s on synthetic nuScenes-shaped clouds (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step = one batch of `--batch` frames through the whole path with inputs already resident in HBM:
  HIP reader (voxelize + PFN + dense bf16 canvas)  ->  dense masked ResNet-18 + ASPP + CenterHead (PyTorch-ROCm + HIP conv kernels,
  bf16, channels_last)  ->  decode + batched rotated NMS (HIP; on the model's own side stream, launched behind the NEXT batch's reader, so it runs
  beside that batch's convolutions: PNX_DECODE_STREAM=0 keeps one stream).  FOUR different frame batches rotate through the loop.
`--gpus N` without WORLD_SIZE in the environment re-launches itself under torch.distributed.run with N ranks (one per GPU, RCCL);
under a launcher (RANK/LOCAL_RANK/WORLD_SIZE set) it is one rank.  Frames are sharded by rank, there is no collective on the path
("replicas only"); time = max over ranks between barriers.  Rank 0 prints ONE JSON line.  Extra objects:
  roofline        the READER (all of its kernels): algorithmic bytes (24*N + nx*ny*64*2) * frames per launch over the time
                  from the reader's first to its last kernel, HIP events recorded on the reader's own stream inside libpnx_hip.so
  roofline_fill   its dominant interval (span grouping + PFN kernel with the canvas zero-fill kernel beside it on a second stream, HBM-write
                  bound) with the canvas bytes the two write
  roofline_pfn    the same interval read as MFMA work (8 832 FLOP per kept point)
  host_enqueue_ms_per_step   launch-thread time per step (the backbone, the head and the decoder are one C call each: pnx_enqueue)
  value_train / train / roofline_train   the training step of BASELINE configs[2] (C2 x 4 frames per GPU: forward + CenterHead losses +
                  backward + clip + AdamW + OneCycle, bf16 autocast, DDP + SyncBN when N > 1), frames/s and dense-equivalent MFMA FLOP rate
  value_with_h2d_merge   the same loop fed the way the reference's loader feeds it (collate.py:15-22, trainer.py:111, nusc.py:101-121):
                  every step the frames' RAW sweeps are copied into pinned memory, uploaded on a side stream (double-buffered) and merged
                  on the device (pnx_merge_sweeps: per-sweep transform, time lag, batch index) before the reader sees them
  value_train_fp32 / train_fp32   the same step at the reference's precision (fp32 channels_last; 3x3 layers as three bf16 products accumulated in fp32 on the HIP kernels, the rest on MIOpen fp32, immediate mode); N = 1 only;
                  PNX_BENCH_NO_TRAIN_FP32=1 skips it (MIOpen compiles its fp32 kernels for ~3.5 minutes on a fresh box)
  value_uniform / roofline_uniform   frames/s on the worst-case uniform cloud (~1.2 points per pillar) and the reader's in-loop roofline on it
  ranks           per-rank ms/step (min / max) and the world size the backend reports; N > 1 runs the short form (value, roofline, value_train)
  value_c4/_c5    frames/s of the Waymo detector (configs/pillarnext_b_waymo.yaml) at BASELINE configs[3] / [4]: 180 k points bf16, 540 k fp16
  sections_us     reader / backbone / neck / head / decode+NMS per step (torch.cuda events, separate short pass)
  nms_us          stand-alone batched rotated NMS on SURVEY 8d's box sets
  cpu_baseline    the CPU oracle ("port") on a bounded sample of the same frames, hot path only: all cores (threads over frames)
                  as `value`, one thread and an own PyTorch-CPU statement of the op sequence beside it; `c1` = the same on BASELINE
                  configs[0] (50 k points, 0.2 m, 512 x 512, the reference's CPU-runnable case)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ROTATE = 4             # distinct frame batches in the timed loop


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PNX_BENCH_BATCH", "12")),
                    help="frames per GPU per step (round 3, one MI355X, same box: 8 -> 569, 12 -> 595, 16 -> 598 frames/s; the reader is at its best at 12)")
    ap.add_argument("--config", default="C2")
    ap.add_argument("--dist", default="sweep", choices=["uniform", "sweep"],
                    help="sweep = ring-structured 10-sweep cloud (BASELINE configs[1]); uniform = worst case, ~1.2 points per pillar")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames of the bounded one-thread CPU-baseline sample (0 = skip all CPU legs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dry-run", action="store_true",
                    help="launch/rendezvous/timing/JSON plumbing only, no GPU work (CPU test of the --gpus path, backend gloo)")
    ap.add_argument("--include-h2d", action="store_true", help="also time the loop with the COLLATED frames uploaded from pinned host memory each step (no merge)")
    ap.add_argument("--no-extras", action="store_true", help="skip value_uniform / sections / NMS / CPU legs (profiling runs)")
    ap.add_argument("--no-back-to-back", action="store_true", help="skip the 15 back-to-back reader calls behind the timed loop (PMC passes: every reader call is then an in-loop call)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="ranks beyond the visible GPUs share devices (rank r -> cuda:(r mod device_count)): runs the N-rank path with real kernels on a 1-GPU box (use --backend gloo)")
    ap.add_argument("--leg", default="", choices=["", "train_fp32", "train_bf16"],
                    help="internal: run ONE leg in this process and print its dict as a JSON line (the fp32 training leg runs in a child process with a clean MIOpen environment)")
    ap.add_argument("--all-legs", action="store_true", help="with --gpus > 1: also run the uniform / H2D-merge / Waymo legs (default: value, roofline, value_train only, so that an 8-rank run stays short)")
    return ap.parse_args()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn(a):
    """`python bench.py --gpus N` on its own: become N ranks (tools/test.py:26-31 relies on torchrun for the same)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------ CPU baselines (rank 0, N = 1)
def cpu_baselines(cfg, config, dist_name, frames_1t):
    """Bounded samples (~10-30 s of CPU work in total) of voxelize + PFN + scatter to the fp32 canvas on the host:
    the plain-C oracle on one thread and on all cores (threads over frames; ctypes releases the GIL), and an own PyTorch-CPU
    statement of the same op sequence (oracle/torch_cpu_reader.py) with 1 and all threads."""
    from concurrent.futures import ThreadPoolExecutor

    import torch

    from oracle import oracle as O
    from oracle import torch_cpu_reader as T
    from pillarnext_amd import synth

    layers = synth.pfn_params()
    ncore = os.cpu_count() or 1
    pool = [synth.make_batch(config, 1, dist_name, frame0=f) for f in range(min(frames_1t, 8))]

    def one(i):
        O.reader_forward(pool[i % len(pool)], cfg["pc_range"], cfg["voxel_size"], [64, 64], layers, B=1, want_canvas=True)

    O.lib()
    t0 = time.perf_counter()
    for i in range(frames_1t):
        one(i)
    t1 = time.perf_counter() - t0
    # all cores: the canvas alone is 0.53 GB fp32 per frame in flight, so the thread count is capped at 32
    nthr = min(ncore, 32)
    n_all = nthr * 2
    with ThreadPoolExecutor(nthr) as ex:
        list(ex.map(one, range(nthr)))  # warm
        t0 = time.perf_counter()
        list(ex.map(one, range(n_all)))
        tall = time.perf_counter() - t0
    res = {"value": round(n_all / tall, 3), "unit": "frames/s (voxelize+PFN+scatter only, fp32 canvas)", "cores": nthr, "kind": "port",
           "sample": f"{n_all} frames of {config}/{dist_name} over {nthr} threads (one frame per call), oracle/pnx_oracle.c orc_reader_forward; "
                     f"host {ncore} logical cores, {cpu_model()}",
           "value_1thread": round(frames_1t / t1, 3), "sample_1thread": f"{frames_1t} frames, 1 thread"}
    # own PyTorch-CPU statement of the op sequence (sort-unique + scatter, like the reference's)
    tc = {}
    for thr, nfr in ((1, 3), (min(ncore, 32), 8)):
        torch.set_num_threads(thr)
        T.reader_forward(pool[0], cfg["pc_range"], cfg["voxel_size"], layers)  # warm
        t0 = time.perf_counter()
        for i in range(nfr):
            T.reader_forward(pool[i % len(pool)], cfg["pc_range"], cfg["voxel_size"], layers)
        tc[f"threads_{thr}"] = {"frames_per_s": round(nfr / (time.perf_counter() - t0), 3), "frames": nfr}
    res["torch_cpu"] = tc
    # BASELINE configs[0]: the reference's own CPU-runnable case (C1: 50 k points, 0.2 m pillars, 512 x 512)
    c1 = synth.CONFIGS["C1"]
    pool1 = [synth.make_batch("C1", 1, dist_name, frame0=f) for f in range(8)]

    def one1(i):
        O.reader_forward(pool1[i % 8], c1["pc_range"], c1["voxel_size"], [64, 64], layers, B=1, want_canvas=True)

    one1(0)
    t0 = time.perf_counter()
    for i in range(48):
        one1(i)
    t1c = time.perf_counter() - t0
    with ThreadPoolExecutor(nthr) as ex:
        list(ex.map(one1, range(nthr)))
        t0 = time.perf_counter()
        list(ex.map(one1, range(nthr * 8)))
        tallc = time.perf_counter() - t0
    res["c1"] = {"value": round(nthr * 8 / tallc, 2), "cores": nthr, "value_1thread": round(48 / t1c, 2),
                 "sample": f"C1/{dist_name}: 48 frames on 1 thread, {nthr * 8} frames over {nthr} threads, same C port"}
    # the reference's OWN rotated IoU / NMS on the host (det3d/core/iou3d_nms/src/iou3d_cpu.cpp compiled in place by oracle/Makefile into oracle/_ref,
    # which travels with the snapshot): the "reference" kind of baseline for the post-processing half of the path
    if O.have_ref():
        import numpy as np

        bx, _ = synth.clustered_boxes(1000, 7, spread=12.0)
        t0 = time.perf_counter()
        iou = O.ref_boxes_iou_bev(bx, bx)
        t_iou = time.perf_counter() - t0
        t0 = time.perf_counter()
        keep = O.ref_nms_rotated(bx, 0.2)
        t_nms = time.perf_counter() - t0
        res["nms_reference"] = {"kind": "reference", "cores": 1, "iou_1000x1000_ms": round(t_iou * 1e3, 1), "nms_1000_thr0.2_ms": round(t_nms * 1e3, 1),
                                "kept": int(len(keep)), "pairs_over_thr": int((np.asarray(iou) > 0.2).sum()),
                                "sample": "1000 clustered boxes (synth.clustered_boxes seed 7): IoU-BEV matrix, then mask + greedy NMS (iou3d_nms.cpp:113-159 host loop) -- nms_us.n1000x10 is ten such lists on the GPU"}
    return res


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


# ------------------------------------------------------------------------------------------------ GPU legs
def serving_loop(model, examples, steps, upload=None):
    """`steps` frame batches through the fused graph; batch i+1 is enqueued before the host blocks on (and unpacks) the detections of
    batch i, and every batch's detections are on the host, as dicts, before this returns."""
    pending, out = None, None
    host = 0.0
    for i in range(steps):
        ex = examples[i % len(examples)]
        if upload is not None:
            ex = upload(i)
        t0 = time.perf_counter()
        nxt = model.forward_async(ex)
        host += time.perf_counter() - t0
        if pending is not None:
            out = model.detections(pending.result())
        pending = nxt
    if pending is not None:
        out = model.detections(pending.result())
    serving_loop.host_enqueue_ms = 1e3 * host / max(steps, 1)   # launch-thread time per step: what the host must sustain for the GPU to stay busy
    return out


def waymo_model(config, dtype, dev):
    """configs/pillarnext_b_waymo.yaml (the reference's waymo_det_pp18_aspp_iou_car_sp.yaml model block) at the geometry of BASELINE's
    synthetic Waymo configs, random-init weights, as the fused inference graph."""
    import torch

    from pillarnext_amd import config as C
    from pillarnext_amd import synth
    from pillarnext_amd.models import FusedPillarNeXt

    cfg = C.load(os.path.join(ROOT, "configs", "pillarnext_b_waymo.yaml"))
    g = synth.CONFIGS[config]
    for blk in ("reader", "head", "post_processing"):
        cfg["model"][blk]["voxel_size"] = list(g["voxel_size"])
        cfg["model"][blk]["pc_range"] = list(g["pc_range"])
    torch.manual_seed(0)
    det = C.instantiate(cfg["model"]).to(dev).eval()
    return FusedPillarNeXt(det, dtype=dtype).to(dev).eval()


def nms_bench(dev):
    """SURVEY 8d NMS inputs: n=1000 x 10 classes (thr 0.2, post 83) and n=4096 x 3 classes (thr 0.7/0.2/0.25, post 500)."""
    import torch

    from pillarnext_amd import ops, synth

    res = {}
    for name, n, nseg, thr, post in (("n1000x10_thr0.2_post83", 1000, 10, [0.2] * 10, 83), ("n4096x3_thr0.7_0.2_0.25_post500", 4096, 3, [0.7, 0.2, 0.25], 500)):
        boxes = torch.from_numpy(__import__("numpy").concatenate([synth.clustered_boxes(n, 7 + s)[0] for s in range(nseg)])).to(dev)
        off = (torch.arange(nseg + 1, dtype=torch.int32, device=dev) * n).contiguous()
        th = torch.tensor(thr, dtype=torch.float32, device=dev)
        for _ in range(3):
            ops.nms_batched(boxes, off, th, n, post_max=post)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            keep, cnt = ops.nms_batched(boxes, off, th, n, post_max=post)
        e1.record()
        torch.cuda.synchronize()
        res[name] = {"us": round(e0.elapsed_time(e1) * 1e3 / 20, 1), "kept": [int(v) for v in cnt[:nseg].tolist()]}
    return res


def sections(model, examples, batch):
    """Per-section GPU time of one step (torch.cuda events between the sections; separate pass on ONE stream, median of 5 steps)."""
    import torch

    acc = {}
    for i in range(7):
        marks = []
        packed = []
        model.forward_preds(examples[i % len(examples)]["points"], batch, marks=marks, packed_out=packed)
        pend = model.launch_decode(packed, None)
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(("decode+nms", e))
        pend.result()
        torch.cuda.synchronize()
        if i < 2:
            continue
        for (_, a), (n1, b) in zip(marks[:-1], marks[1:]):
            acc.setdefault(n1, []).append(a.elapsed_time(b) * 1e3)
    # median of five passes: this pass allocates a fresh canvas per call (the timed loop writes a persistent one), and an allocator stall lands in one section
    return {k: round(sorted(v)[len(v) // 2], 1) for k, v in acc.items()}



def backbone_roofline(sec, frames, dev, model, example):
    """MFMA roofline of the masked-dense backbone in REAL FLOPs -- the row segments the kernels compute, measured on this batch's masks --
    with the rate a tuned library GEMM reaches on THIS box next to the 2.5 PFLOP/s spec peak: the chip clocks to its power budget (MI355X guide, DVFS
    give-back), and on random bf16 operands hipBLASLt's 8192^3 GEMM runs at 1.15-1.25 PFLOP/s here (1.5-1.75 on all-zero operands:
    profiles/r06_power_ceiling.txt) -- that, not 2.5, is what a convolution kernel can be priced against in practice."""
    import torch

    from pillarnext_amd import ops

    n = 8192
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn((n, n), device=dev, generator=g).bfloat16()
    b = torch.randn((n, n), device=dev, generator=g).bfloat16()
    for _ in range(3):
        a @ b
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        a @ b
    e1.record()
    torch.cuda.synchronize()
    gemm_tf = 2.0 * n ** 3 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    del a, b
    # active 32-pixel row segments per stage: stage 0 runs on the 3x3 dilation of the occupancy (its entry layer is a SparseConv2d, sparse_resnet.py:53-54),
    # stages 1-3 on the stride-2 pooled masks (sparse_conv.py:16-29)
    ny, nx = (int(v) for v in model.reader.grid_size)
    occ = torch.empty((frames, ny, nx), dtype=torch.uint8, device=dev)
    model.reader.forward_dense(example["points"], frames, dtype=model.dtype, occupancy=occ)
    fr, m = [], occ
    for stage in range(4):
        m = ops.mask_pool3(m, 1 if stage == 0 else 2)
        fr.append(float(torch.nn.functional.max_pool1d(m.float(), 32, 32, ceil_mode=True).mean().item()))
    dense = [764e9, 688e9, 688e9, 191e9]            # dense-equivalent FLOPs per frame and stage at 1440^2
    us = [sec.get(f"backbone.stage{i}", 0.0) for i in range(4)]
    real = sum(d * f for d, f in zip(dense, fr)) * frames
    tot_us = sum(us)
    tf = real / (tot_us * 1e-6) / 1e12 if tot_us > 0 else None
    per_stage = [round(d * f * frames / (u * 1e-6) / 1e12, 1) if u > 0 else None for d, f, u in zip(dense, fr, us)]
    return {"bound": "mfma", "kernel": "backbone: 20 masked 3x3 convolutions (k_conv3x3_pc / _ldsx / _s2), sections_us backbone.*",
            "achieved": round(tf, 1) if tf else None, "peak": 2500.0, "unit": "TFLOP/s (real bf16 FLOPs: dense-equivalent FLOPs of a stage x the fraction of its 32-pixel row segments that hold an active site)",
            "frac": round(tf / 2500.0, 4) if tf else None, "kernel_us": round(tot_us, 1), "algorithmic_flops_per_launch": real,
            "active_row_segments_per_stage": [round(f, 3) for f in fr], "achieved_per_stage": per_stage,
            "dense_equivalent_tflops": round(sum(dense) * frames / (tot_us * 1e-6) / 1e12, 1) if tot_us > 0 else None,
            "library_gemm_tflops": round(gemm_tf, 1), "frac_of_library_gemm": round(tf / gemm_tf, 4) if tf else None,
            "library_gemm": "torch.matmul bf16 8192^3, random-normal operands, measured in this run: the practical MFMA ceiling under the power cap"}


def train_leg(dev, rank, world, frames=int(os.environ.get("PNX_BENCH_TRAIN_FRAMES", "4")), steps=int(os.environ.get("PNX_BENCH_TRAIN_STEPS", "5")),
              warmup=int(os.environ.get("PNX_BENCH_TRAIN_WARMUP", "3")), amp=True):
    """One data-parallel TRAINING step of PillarNeXt-B (reference: trainer/trainer/trainer.py:94-108 -- forward, loss, backward, clip 35,
    AdamW, OneCycle) on synthetic C2 frames + labels, `frames` per GPU, under bf16 autocast in channels_last (the reference trains in
    fp32; the bf16 step is what this repository optimises and is labelled as such).  DDP over the job's process group when world > 1
    (gradient all-reduce over RCCL overlapped with backward; SyncBatchNorm over the active sites).  Returns the dict behind
    value_train / train / roofline_train."""
    import torch
    import torch.distributed as dist

    from pillarnext_amd import dist_utils, synth
    from pillarnext_amd.models import NUSC_TASKS, build_pillarnext_b, convert_sync_batchnorm

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from train_step import synthetic_labels

    for k in ("FWD", "BWD", "WRW"):   # MIOpen's naive reference solvers only lengthen the find step (tools/train_step.py)
        os.environ.setdefault(f"MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_{k}", "0")
    cfg = synth.CONFIGS["C2"]
    torch.manual_seed(0)
    model = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"]).to(dev).train().to(memory_format=torch.channels_last)
    if world > 1:
        model = convert_sync_batchnorm(model)
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index])
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.01)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=0.002, total_steps=1000, div_factor=10.0, pct_start=0.4)
    pts = torch.from_numpy(synth.make_batch("C2", frames, "sweep", frame0=rank * frames)).to(dev)
    net = model.module if hasattr(model, "module") else model
    ny, nx = (int(v) for v in net.reader.grid_size)
    ex = synthetic_labels(NUSC_TASKS, frames, ny // 4, nx // 4, 500, dev, 100 + rank)
    ex.update(points=pts, batch_size=frames)

    def step():
        with torch.enable_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):   # the inference legs around this run under no_grad
            loss, _ = model(ex)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 35)
        opt.step()
        sched.step()
        return loss

    # Rounds 3-5: MIOpen's find pass over the forward / dgrad / wgrad shapes of a 1440^2 training graph (~50 problems) took ~2 minutes and bought a 2.2 x faster
    # step over MIOpen's immediate-mode choices.  Round 6 moved the backbone's, the head's and the neck's 3x3 layers onto the product's kernels: what MIOpen
    # still runs (dilated / 1x1 / 256 -> 64 layers, stride-2 dgrad) is as fast in immediate mode (84.4 vs 84.9 ms per step measured in one run), so immediate
    # mode is the default at every N and the leg takes seconds; PNX_BENCH_TRAIN_FIND=1 turns the find pass back on.
    # `train.miopen` / `ranks.miopen` say which mode a line was measured with.
    bench_mode = torch.backends.cudnn.benchmark
    # The SAME mode at every N, so that value_train(N) / (N x value_train(1)) compares like with like.
    # The fp32 leg: PNX_BENCH_TRAIN_F32_FIND (its backbone is on the product's kernels since round 6, what MIOpen still has to search is the neck / head).
    find = os.environ.get("PNX_BENCH_TRAIN_FIND", "0") == "1" if amp else os.environ.get("PNX_BENCH_TRAIN_F32_FIND", "0") == "1"
    torch.backends.cudnn.benchmark = find
    t_leg = time.perf_counter()
    for _ in range(warmup):
        step()
    torch.cuda.reset_peak_memory_stats(dev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finite = bool(torch.isfinite(loss))
    flops = 3 * 2.96e12 * frames                      # 2.96 TFLOP per frame forward (dense-equivalent, 1440^2, 6 tasks); x 3 for dgrad + wgrad
    tf = flops * steps / dt / 1e12
    if not amp:   # the reference's precision (trainer/trainer/trainer.py:94-108: fp32 throughout, no autocast in its tree), channels_last, torch / MIOpen convolutions
        torch.backends.cudnn.benchmark = bench_mode
        res = {"value_train_fp32": round(frames * world * steps / dt, 2),
               "train_fp32": {"frames_per_gpu_per_step": frames, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2),
                              "dtype": "fp32, channels_last -- the reference's training precision; backbone 3x3 layers: three bf16 products of the operands' bf16 halves "
                                       "accumulated in fp32 on the masked HIP kernels (pnx_conv3x3_x3, relative error 4e-6; PNX_TRAIN_F32_HIP=0: MIOpen fp32), "
                                       "the head's 64 -> 64 and output convolutions and the neck's BasicBlock likewise; stride-2 dgrad, dilated / 1x1 / 256 -> 64 layers on MIOpen fp32; run in a child process with MIOpen's default solver set",
                              "loss_finite": finite, "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                              "miopen": "find (cudnn.benchmark)" if find else "immediate mode", "leg_seconds": round(time.perf_counter() - t_leg, 1)}}
        del model, opt, ex
        torch.cuda.empty_cache()
        return res
    res = {"value_train": round(frames * world * steps / dt, 2),
           "train": {"frames_per_gpu_per_step": frames, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2),
                     "dtype": "bf16 autocast, channels_last (the reference trains fp32: value_train_fp32 / train_fp32 beside this line time that)",
                     "step": "forward + CenterHead losses + backward + clip 35 + AdamW + OneCycle, DDP + SyncBN when n_gpus > 1",
                     "loss_finite": finite, "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                     "miopen": "find (cudnn.benchmark)" if find else "immediate mode", "leg_seconds": round(time.perf_counter() - t_leg, 1)},
           "roofline_train": {"bound": "mfma", "kernel": "whole training step (backbone 3x3 layers, the head's 64 -> 64 and output convolutions, the neck's BasicBlock: forward, stride-1 dgrad and weight gradients on the HIP kernels; stride-2 dgrad, dilated / 1x1 / 256 -> 64 layers and dense BatchNorm on MIOpen)",
                              "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s (dense-equivalent bf16 FLOPs, 3 x forward)",
                              "frac": round(tf / 2500.0, 4), "algorithmic_flops_per_step": flops}}
    torch.backends.cudnn.benchmark = bench_mode
    del model, opt, ex
    torch.cuda.empty_cache()
    return res


def train_child_leg(dev, leg):
    """A single-GPU training leg ("train_bf16": bf16 autocast; "train_fp32": the reference's precision, torch / MIOpen convolutions) in a
    CHILD process.  Why a child: this process pins MIOpen's solver search for the inference legs (MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0,
    main()); MIOpen caches such switches per process, and without the asm NHWC solvers (a) the fp32 forward convolutions fall back to solvers that transpose
    through workspaces -- 551 ms and 75 GiB per step instead of 375 ms and 50 GiB (rounds 4-5 reported the former; tools/train_step.py --nhwc always showed
    the latter) -- and (b) the bf16 leg's find pass takes 127 s instead of 40 s for the same 92 ms step (profiles/r06_train_fp32_env.txt)."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if not k.startswith("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM") and k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    vis, idx = os.environ.get("HIP_VISIBLE_DEVICES"), dev.index or 0   # the child sees this rank's GPU as its device 0
    env["HIP_VISIBLE_DEVICES"] = vis.split(",")[idx] if vis else str(idx)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", leg], env=env, capture_output=True, text=True, timeout=1500)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {("train" if leg == "train_bf16" else leg): {"error": (p.stderr or p.stdout)[-400:]}}
    return json.loads(lines[-1])


def main():
    a = parse()
    if a.leg:
        import torch

        torch.cuda.set_device(0)
        d0 = torch.device("cuda", 0)
        print(json.dumps(train_leg(d0, 0, 1, steps=3, warmup=2, amp=False) if a.leg == "train_fp32" else train_leg(d0, 0, 1)))
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn(a))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MIOPEN_USER_DB_PATH" not in os.environ and not a.dry_run:
            # every rank times MIOpen's solvers for the same few conv shapes during warm-up: give each its own find-db so that N
            # processes do not serialise on one sqlite file lock
            db = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pnx_miopen_rank{rank}")
            os.makedirs(db, exist_ok=True)
            os.environ["MIOPEN_USER_DB_PATH"] = db
        dist.init_process_group("gloo" if a.dry_run else a.backend, init_method="env://")

    if a.dry_run:
        # plumbing only: spawn, rendezvous, barrier-bracketed timing, max over ranks, one JSON line from rank 0
        def barrier():
            if world > 1:
                dist.barrier()
        for _ in range(a.warmup):
            time.sleep(0.001)
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            time.sleep(0.001)
        barrier()
        dt = time.perf_counter() - t0
        rank_ms = [round(dt / max(a.steps, 1) * 1e3, 3)]
        if world > 1:
            t = torch.zeros(world, dtype=torch.float64)
            t[rank] = dt
            dist.all_reduce(t, op=dist.ReduceOp.SUM)   # every rank's own time, as the real run reports it
            rank_ms = [round(float(v) / max(a.steps, 1) * 1e3, 3) for v in t.tolist()]
            dt = float(t.max().item())
        if rank == 0:
            print(json.dumps({"metric": "dry run (no GPU work)", "ranks": {"world_size_seen_by_backend": dist.get_world_size() if world > 1 else 1,
                                                                            "backend": "gloo" if world > 1 else None, "ms_per_step_min": min(rank_ms),
                                                                            "ms_per_step_max": max(rank_ms)}, "value": round(a.batch * world * a.steps / dt, 2), "unit": "frames/s", "n_gpus": world,
                              "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none", "dry_run": True,
                              "config": {"workload": "launcher plumbing only", "global_batch": a.batch * world,
                                         "parallelism": f"frame-sharded replicas x{world}"}}))
        if world > 1:
            dist.destroy_process_group()
        return

    import ctypes

    if a.share_gpu:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    # The dense 256-channel layers at 180 x 180 (neck pre_conv, ASPP) are MIOpen's.  Its find step times CK's grouped convolution (287 us) and the
    # asm implicit-GEMM solver (477 us) and picked the slower one in about one run of four (and in every run under rocprofv3): +0.95 ms per
    # step.  With that solver out of the search the choice is CK in every run (562-564 frames/s, three runs).  The naive reference solver
    # (0.46 s per call, 40 calls) only lengthens the find step.
    os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC", "0")
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")
    if not os.environ.get("PNX_NO_MIOPEN_BENCH"):
        torch.backends.cudnn.benchmark = True  # MIOpen times its applicable solvers once per conv shape (during warm-up)
    dev = torch.device("cuda", local)

    from pillarnext_amd import _lib, synth
    from pillarnext_amd.models import FusedPillarNeXt, build_pillarnext_b

    cfg = synth.CONFIGS[a.config]
    torch.manual_seed(0)
    model = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"]).to(dev).eval()
    model = FusedPillarNeXt(model).to(dev).eval()  # same network: eval-BN folded, HIP epilogues, merged head branches

    def make_examples(dist_name):
        # frames are sharded across ranks: rank r owns frames [r*ROTATE*B, (r+1)*ROTATE*B) (replicas only, no collective on the path)
        exs = []
        for k in range(ROTATE):
            pts = torch.from_numpy(synth.make_batch(a.config, a.batch, dist_name, frame0=(rank * ROTATE + k) * a.batch)).to(dev)
            exs.append({"points": pts, "token": [f"r{rank}b{k}f{i}" for i in range(a.batch)], "batch_size": a.batch})
        return exs

    examples = make_examples(a.dist)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(exs, steps, upload=None):
        barrier()
        t0 = time.perf_counter()
        out = serving_loop(model, exs, steps, upload)
        barrier()
        dt = time.perf_counter() - t0
        timed.rank_ms = [round(dt / steps * 1e3, 3)]
        if world > 1:
            t = torch.zeros(world, dtype=torch.float64, device=dev)
            t[rank] = dt
            dist.all_reduce(t, op=dist.ReduceOp.SUM)          # every rank's own time: the line reports min / max over ranks, value uses the max
            timed.rank_ms = [round(float(v) / steps * 1e3, 3) for v in t.tolist()]
            dt = float(t.max().item())
        return dt, out

    L = _lib.lib()
    with torch.no_grad():
        for i in range(max(a.warmup, ROTATE)):  # at least one untimed pass per batch: library handles, MIOpen find mode, workspaces
            model(examples[i % ROTATE])
        _lib.check(L.pnx_profile_begin(max(a.steps, 1)), "pnx_profile_begin")
        dt, out = timed(examples, a.steps)
        rank_ms = list(timed.rank_ms)
        host_ms = serving_loop.host_enqueue_ms
        r_us, c_us, ns = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int32(0)
        _lib.check(L.pnx_profile_end(ctypes.byref(r_us), ctypes.byref(c_us), ctypes.byref(ns)), "pnx_profile_end")
        pfn_us, vox_us = float(L.pnx_profile_last_pfn_us()), float(L.pnx_profile_last_voxelize_us())
        # the same reader calls back to back, nothing else on the GPU between them: the convolutions of the detector leave the chip in a
        # power / cache state in which the reader's kernels run 12-20 % slower (profiles/r04_reader_between.txt); `roofline` is the in-loop figure
        ny_, nx_ = (int(v) for v in model.reader.grid_size)
        rb_us, cb_us, nb_ = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int32(0)
        if not a.no_back_to_back:
            # into the SAME canvas / occupancy buffers the timed loop writes (the launch plan's persistent ones): a freshly allocated 3.2 GB canvas
            # made this figure bimodal from run to run (617 or ~700 us, with or without the decoder stream; where the allocator places it)
            bbp = model._backbone_plan(a.batch, dev) if model._plan_ok() else None
            cv = bbp["canvas"] if bbp is not None else torch.empty((a.batch, 64, ny_, nx_), dtype=model.dtype, device=dev, memory_format=torch.channels_last)
            oc = bbp["occ"] if bbp is not None else torch.empty((a.batch, ny_, nx_), dtype=torch.uint8, device=dev)
            for i in range(3):
                model.reader.forward_dense(examples[i % ROTATE]["points"], a.batch, dtype=model.dtype, out=cv, occupancy=oc)
            torch.cuda.synchronize()
            _lib.check(L.pnx_profile_begin(12), "pnx_profile_begin")
            for i in range(12):
                model.reader.forward_dense(examples[i % ROTATE]["points"], a.batch, dtype=model.dtype, out=cv, occupancy=oc)
            torch.cuda.synchronize()
            _lib.check(L.pnx_profile_end(ctypes.byref(rb_us), ctypes.byref(cb_us), ctypes.byref(nb_)), "pnx_profile_end")
            del cv, oc, bbp

        extras = {}
        short = world > 1 and not a.all_legs   # an N-rank run: value, roofline and value_train only (DESIGN.md section 7 gives its expected wall time)
        if not a.no_extras and not short:
            other = "uniform" if a.dist == "sweep" else "sweep"
            ex2 = make_examples(other)
            for i in range(ROTATE):
                model(ex2[i])
            st2 = max(a.steps // 2, 4)
            _lib.check(L.pnx_profile_begin(st2), "pnx_profile_begin")
            dt2, _ = timed(ex2, st2)
            r2_us, c2_us, n2 = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int32(0)
            _lib.check(L.pnx_profile_end(ctypes.byref(r2_us), ctypes.byref(c2_us), ctypes.byref(n2)), "pnx_profile_end")
            extras[f"value_{other}"] = round(a.batch * world * st2 / dt2, 2)
            cnt2 = torch.zeros(2, dtype=torch.int32, device=dev)
            model.reader.forward_dense(ex2[0]["points"], a.batch, counts=cnt2)
            b2 = 24 * ex2[0]["points"].shape[0] + a.batch * nx_ * ny_ * 64 * 2
            # SURVEY 8(d) asks for the reader's roofline on BOTH distributions: same byte formula, same in-loop HIP-event interval
            extras[f"roofline_{other}"] = {"bound": "hbm", "kernel": f"reader, all of its kernels, cloud={other}, in the detector's loop",
                                           "achieved": round(b2 / r2_us.value / 1e3, 1) if r2_us.value > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(b2 / r2_us.value / 1e3 / HBM_PEAK_GBS, 4) if r2_us.value > 0 else None,
                                           "algorithmic_bytes_per_launch": b2, "kernel_us": round(r2_us.value, 2), "samples": n2.value,
                                           "pillars_per_launch": int(cnt2[0]), "kept_points_per_launch": int(cnt2[1])}
            # the loader's side of the contract in the loop: raw sweeps -> pinned memory -> H2D on a side stream -> device merge
            from pillarnext_amd.io import PointUploader, SweepMerger

            raws = [synth.raw_sweeps(e["points"].cpu().numpy()) for e in examples]
            upr = PointUploader(max(len(r) for r, _ in raws), 4, dev)
            merger = SweepMerger()
            descs = [merger.descriptors(sg, dev) for _, sg in raws]
            mbuf = [torch.empty((max(len(r) for r, _ in raws), 6), dtype=torch.float32, device=dev) for _ in range(2)]
            mcnt = torch.zeros(1, dtype=torch.int32, device=dev)

            def upload_merge(i):
                k = i % ROTATE
                raw_dev = upr.upload_rows(raws[k][0])
                pts, _ = merger(raw_dev, descs[k], n_copy=4, out=mbuf[i % 2], n_out=mcnt)
                return {"points": pts, "token": examples[k]["token"], "batch_size": a.batch}

            for i in range(ROTATE):
                model(upload_merge(i))
            dt4, out4 = timed(examples, a.steps, upload_merge)
            extras["value_with_h2d_merge"] = round(a.batch * world * a.steps / dt4, 2)
            extras["h2d_merge"] = {"raw_rows_per_step": int(len(raws[0][0])), "sweeps_per_frame": len(raws[0][1]) // a.batch,
                                   "bytes_per_step": int(raws[0][0].nbytes), "detections_last_step": int(sum(len(v["scores"]) for v in out4.values()))}
            if a.include_h2d:
                from pillarnext_amd.io import PointUploader

                up = PointUploader(max(e["points"].shape[0] for e in examples), 5, dev)
                host = [[e["points"][e["points"][:, 0] == b][:, 1:].cpu().numpy() for b in range(a.batch)] for e in examples]

                def upload(i):
                    pts, B = up.upload(host[i % ROTATE])
                    return {"points": pts, "token": examples[i % ROTATE]["token"], "batch_size": B}


                for i in range(ROTATE):
                    model(upload(i))
                dt3, _ = timed(examples, a.steps, upload)
                extras["value_with_h2d"] = round(a.batch * world * a.steps / dt3, 2)
            # BASELINE configs[3] / [4]: the Waymo detector (configs/pillarnext_b_waymo.yaml: 2 tasks, iou head, NMS pre 4096 / post 500)
            # at the synthetic Waymo geometry -- C4 180 k points bf16, C5 540 k points (3 sweeps) fp16 -- same serving loop, resident inputs
            if a.config == "C2" and not os.environ.get("PNX_BENCH_NO_WAYMO"):
                for wc, wdt, wb in (("C4", torch.bfloat16, a.batch), ("C5", torch.float16, max(a.batch // 2, 1))):
                    wm = waymo_model(wc, wdt, dev)
                    wex = []
                    for k in range(2):
                        wp = torch.from_numpy(synth.make_batch(wc, wb, "sweep", frame0=(rank * 2 + k) * wb)).to(dev)
                        wex.append({"points": wp, "token": [f"w{k}f{i}" for i in range(wb)], "batch_size": wb})
                    for i in range(3):
                        wm(wex[i % 2])
                    wsteps = max(a.steps // 3, 4)
                    barrier()
                    t0 = time.perf_counter()
                    serving_loop(wm, wex, wsteps)
                    barrier()
                    wdt_s = time.perf_counter() - t0
                    if world > 1:
                        t = torch.tensor([wdt_s], dtype=torch.float64, device=dev)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)
                        wdt_s = float(t.item())
                    extras[f"value_{wc.lower()}"] = round(wb * world * wsteps / wdt_s, 2)
                    extras.setdefault("waymo", {})[wc] = {"frames_per_gpu_per_step": wb, "steps": wsteps, "dtype": str(wdt).split(".")[-1],
                                                         "points_per_frame": synth.CONFIGS[wc]["n"], "grid": [1504, 1504],
                                                         "ms_per_step": round(wdt_s / wsteps * 1e3, 3)}
                    del wm, wex
                    torch.cuda.empty_cache()
            # the training step (BASELINE configs[2]: 4 frames per GPU), timed by this run: value_train / roofline_train
        if not a.no_extras:
            if a.config == "C2" and not os.environ.get("PNX_BENCH_NO_TRAIN"):
                # N = 1: both training legs in child processes with MIOpen's default solver set (train_child_leg); N > 1: in process, under DDP
                extras.update(train_child_leg(dev, "train_bf16") if world == 1 else train_leg(dev, rank, world))
                if not short and not os.environ.get("PNX_BENCH_NO_TRAIN_FP32"):
                    extras.update(train_child_leg(dev, "train_fp32"))
            if rank == 0 and not short:
                extras["sections_us"] = sections(model, examples, a.batch)
                extras["nms_us"] = nms_bench(dev)
                extras["roofline_backbone"] = backbone_roofline(extras["sections_us"], a.batch, dev, model, examples[0])
            for i in range(ROTATE):  # leave the persistent workspaces in the main distribution's state
                model(examples[i])

    frames = a.batch * world * a.steps
    nx, ny = model.reader._geom.gx, model.reader._geom.gy
    n_pts = examples[0]["points"].shape[0]
    canvas_bytes = a.batch * nx * ny * 64 * 2
    reader_bytes = 24 * n_pts + canvas_bytes                     # (24*N + nx*ny*64*e) per frame, x frames per launch
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    model.reader.forward_dense(examples[0]["points"], a.batch, counts=counts)
    P, n_kept = (int(v) for v in counts.tolist())
    reader_gbs = reader_bytes / (r_us.value * 1e-6) / 1e9 if r_us.value > 0 else None
    split = (ctypes.c_int32 * 3)()
    L.pnx_reader_fill_split(split)
    # the PFN launch writes the P pillar cells and its share of the pillar-free tiles; the rest of the zero-fill rides on the grouping kernels
    launch_bytes = int(P * 128 + (canvas_bytes - P * 128) * (100 - sum(split)) / 100)
    fill_gbs = launch_bytes / (c_us.value * 1e-6) / 1e9 if c_us.value > 0 else None
    pfn_flops = 2.0 * n_kept * (10 * 32 + 64 * 64)               # 8 832 FLOP per kept point
    pfn_tf = pfn_flops / (pfn_us * 1e-6) / 1e12 if pfn_us > 0 else None
    traffic, tsrc, tcond = None, None, None
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tj):
        try:
            ent = json.load(open(tj)).get(f"{a.config}_b{a.batch}_{a.dist}", {})
            traffic = ent.get("hbm_bytes_per_launch") if ent.get("pipeline") == "spans" else None   # entries of older pipelines do not describe this one
            tcond = ent.get("condition", "reader calls back to back (tools/reader_ab.py)") if traffic is not None else None
            tsrc = ("profiles/pmc_traffic.json (rocprofv3 --pmc passes of tools/profile_round.sh at this batch size, not measured in this run)" if traffic is not None
                    else f"no PMC entry for {a.config} / {a.batch} frames / {a.dist} of the current pipeline: run tools/profile_round.sh with PNX_BENCH_BATCH={a.batch}")
        except Exception:
            traffic = None
    res = {
        "metric": "frames/s PillarNeXt-B nuScenes 300k-pt cloud (inference, end-to-end)", "value": round(frames / dt, 2), "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "host_enqueue_ms_per_step": round(host_ms, 3),   # launch-thread time inside forward_async per step (launch plans: pnx_enqueue)
        "ranks": {"world_size_seen_by_backend": dist.get_world_size() if world > 1 else 1, "backend": a.backend if world > 1 else None,
                  "ms_per_step_min": min(rank_ms), "ms_per_step_max": max(rank_ms),
                  "miopen": {"inference": "find (cudnn.benchmark), one find-db per rank" if torch.backends.cudnn.benchmark else "immediate mode",
                             "train": "find (cudnn.benchmark), one find-db per rank" if os.environ.get("PNX_BENCH_TRAIN_FIND", "0") == "1" else "immediate mode"}},
        # which two numbers a scaling efficiency is computed from: the same leg, the same MIOpen mode, the same frames per GPU at N and at 1
        "efficiency_basis": {"inference": "value(N) / (N x value(1))", "train": "value_train(N) / (N x value_train(1))",
                             "frames_per_gpu_per_step": {"inference": a.batch, "train": int(os.environ.get("PNX_BENCH_TRAIN_FRAMES", "4"))},
                             "n1_reference": "BENCH_rNN.json of the same round (the driver's N = 1 run of this file); every N uses MIOpen find for both legs"},
        "config": {"workload": f"{a.config}: PillarNeXt-B nuScenes inference, {cfg['n']} pts/frame, voxel {cfg['voxel_size'][0]} m, BEV {nx}x{ny}, "
                               f"6 tasks/10 classes, cloud={a.dist} (1.5 % of the rows outside the range), random-init weights, {ROTATE} distinct frame batches rotating, "
                               f"inputs resident in HBM (value_with_h2d_merge: raw sweeps uploaded from pinned memory + merged on the device every step)",
                   "frames_per_gpu_per_step": a.batch, "global_batch": a.batch * world, "parallelism": f"frame-sharded replicas x{world}",
                   "reader_dtype": "fp32 layer 0 + fp16x3 (22-bit) layer 1 on MFMA -> bf16 canvas", "pillars_per_launch": P,
                   "kept_points_per_launch": n_kept},
        "roofline": {"bound": "hbm", "kernel": "reader, ALL of its kernels (clear, chunk sort, then slab totals + span carve + span grouping/PFN + tail with the canvas zero-fill kernel beside them on a second stream)",
                     "achieved": round(reader_gbs, 1) if reader_gbs else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(reader_gbs / HBM_PEAK_GBS, 4) if reader_gbs else None, "traffic": traffic, "traffic_source": tsrc, "traffic_condition": tcond,
                     "algorithmic_bytes_per_launch": reader_bytes, "kernel_us": round(r_us.value, 2), "samples": ns.value,
                     "voxelize_us": round(vox_us, 2),
                     "back_to_back": None if a.no_back_to_back else {"kernel_us": round(rb_us.value, 2), "frac": round(reader_bytes / rb_us.value / 1e3 / HBM_PEAK_GBS, 4) if rb_us.value > 0 else None,
                                      "note": "the same reader calls with nothing else on the GPU between them (12 calls); frac above is measured inside the detector's loop"}},
        "roofline_fill": {"bound": "hbm", "kernel": "k_span_pfn (span grouping + PFN: the pillar cells) and k_canvas_fill_bytes (every pillar-free tile) running concurrently, fork to join",
                          "achieved": round(fill_gbs, 1) if fill_gbs else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(fill_gbs / HBM_PEAK_GBS, 4) if fill_gbs else None, "algorithmic_bytes_per_launch": launch_bytes,
                          "kernel_us": round(c_us.value, 2), "zero_fill_percent_carried_by_grouping_kernels": [int(v) for v in split]},
        "roofline_pfn": {"bound": "mfma", "kernel": "k_span_pfn (same interval, the zero-fill beside it): fp32 v_mfma_f32_32x32x2_f32 layer 0 + 3 x v_mfma_f32_32x32x16_f16 layer 1",
                         "achieved": round(pfn_tf, 2) if pfn_tf else None, "peak": 157.3, "unit": "TFLOP/s (reference fp32 FLOPs)",
                         "frac": round(pfn_tf / 157.3, 4) if pfn_tf else None, "kernel_us": round(pfn_us, 2), "algorithmic_flops_per_launch": pfn_flops},
    }
    res.update(extras)
    if rank == 0:
        if world == 1 and a.cpu_frames > 0 and not a.no_extras:
            res["cpu_baseline"] = cpu_baselines(cfg, a.config, a.dist, a.cpu_frames)
        else:
            res["cpu_baseline"] = None
        res["detections_last_step"] = int(sum(len(v["scores"]) for v in out.values()))
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
