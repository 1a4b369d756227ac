#!/usr/bin/env python3
"""Section timing of the fused inference graph (events)."""
import argparse, os, sys
from collections import defaultdict
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import synth
from pillarnext_amd.models import FusedPillarNeXt, build_pillarnext_b

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--dist", default="uniform")
a = ap.parse_args()
cfg = synth.CONFIGS["C2"]
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
model = FusedPillarNeXt(build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"]).cuda().eval()).cuda().eval()
pts = torch.from_numpy(synth.make_batch("C2", a.batch, a.dist)).cuda()
ex = {"points": pts, "token": [str(i) for i in range(a.batch)], "batch_size": a.batch}
acc = defaultdict(float)
for it in range(a.iters + 2):
    marks = []
    packed = []
    model.forward_preds(pts, a.batch, marks, packed_out=packed)
    out = model.launch_decode(packed, ex["token"]).result()
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("predict", e))
    torch.cuda.synchronize()
    if it >= 2:
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            acc[n1] += e0.elapsed_time(e1)
tot = sum(acc.values()) / a.iters
print(f"fused graph, batch {a.batch}: {tot:.2f} ms per step = {tot/a.batch:.2f} ms/frame")
for k, v in acc.items():
    print(f"  {k:20s} {v/a.iters:8.3f} ms")
