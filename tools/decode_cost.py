#!/usr/bin/env python3
"""What the decoder costs the detector's loop: the same 12-frame batches through the fused graph with the decoder (a) on its own stream (default), (b) inline on the
detector's stream, (c) not run at all (an upper bound: the head maps are produced and dropped).  usage (GPU box): python tools/decode_cost.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import synth
from pillarnext_amd.models import FusedPillarNeXt, build_pillarnext_b

os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC", "0")
torch.backends.cudnn.benchmark = True
cfg = synth.CONFIGS["C2"]
torch.manual_seed(0)
model = FusedPillarNeXt(build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"]).cuda().eval()).cuda().eval()
B = 12
exs = [{"points": torch.from_numpy(synth.make_batch("C2", B, "sweep", frame0=4 * i * B)).cuda(), "token": [str(k) for k in range(B)], "batch_size": B} for i in range(4)]

def loop(mode, steps):
    pend = None
    for i in range(steps):
        ex = exs[i % 4]
        if mode == "none":
            with torch.no_grad():
                model.forward_preds(ex["points"], B, packed_out=[])
        else:
            nxt = model.forward_async(ex)
            if pend is not None:
                pend.result()
            pend = nxt
    if pend is not None:
        pend.result()
    torch.cuda.synchronize()

for mode in ("side", "inline", "none", "side", "inline", "none"):
    model.decode_on_side_stream = mode == "side"
    loop(mode, 6)
    t0 = time.perf_counter()
    loop(mode, 20)
    dt = time.perf_counter() - t0
    print(f"decoder {mode:6s}: {dt / 20 * 1e3:.2f} ms per 12-frame step = {B * 20 / dt:.1f} frames/s", flush=True)
