#!/usr/bin/env python3
"""Per-layer GPU time of the fused inference graph (events around every conv module of FusedPillarNeXt; the ASPP branches are
plain F.conv2d calls and show up as the remainder of the 'mapping+neck' section)."""
import argparse, os, sys
from collections import defaultdict
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import synth
from pillarnext_amd.models import FusedPillarNeXt, build_pillarnext_b

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--dist", default="sweep")
a = ap.parse_args()
cfg = synth.CONFIGS["C2"]
torch.manual_seed(0)
model = FusedPillarNeXt(build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"]).cuda().eval()).cuda().eval()
pts = torch.from_numpy(synth.make_batch("C2", a.batch, a.dist)).cuda()
recs, info = defaultdict(list), {}
for name, mod in model.named_modules():
    if type(mod).__name__ in ("_FusedConv", "_HipConv3x3", "_HipSepHeadOut", "_HipDeconv2x2"):
        def pre(m, inp, name=name):
            m._t0 = torch.cuda.Event(enable_timing=True); m._t0.record()
        def post(m, inp, out, name=name):
            e = torch.cuda.Event(enable_timing=True); e.record()
            recs[name].append((m._t0, e))
            w = getattr(m, "weight", None)
            info[name] = f"{type(m).__name__} in{tuple(inp[0].shape)} -> {tuple(out.shape)}" + (f" w{tuple(w.shape)}" if w is not None else "")
        mod.register_forward_pre_hook(pre); mod.register_forward_hook(post)
for it in range(a.iters + 2):
    if it == 2:
        for k in recs: recs[k].clear()
    model.forward_preds(pts, a.batch, None, packed_out=[])
torch.cuda.synchronize()
tot = 0.0
for k, v in recs.items():
    us = sum(e0.elapsed_time(e1) for e0, e1 in v) / a.iters * 1e3
    tot += us
    print(f"{us:9.1f} us  {k:22s} {info[k]}")
print(f"{tot:9.1f} us  sum of the hooked layers (batch {a.batch}, {a.dist})")
