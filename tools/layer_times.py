#!/usr/bin/env python3
"""Per-module GPU time of the dense part of PillarNeXt-B (events around every conv / norm / leaf module)."""
import argparse
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import synth  # noqa: E402
from pillarnext_amd.models import build_pillarnext_b  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    torch.manual_seed(0)
    model = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"]).cuda().eval()
    for m in (model.backbone, model.neck, model.head):
        m.to(memory_format=torch.channels_last, dtype=torch.bfloat16)
    pts = torch.from_numpy(synth.make_batch(a.config, a.batch, "uniform")).cuda()
    ex = {"points": pts, "token": [str(i) for i in range(a.batch)], "batch_size": a.batch}
    recs = defaultdict(list)
    info = {}

    def pre(name):
        def f(mod, inp):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            mod._t0 = e
        return f

    def post(name):
        def f(mod, inp, out):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            recs[name].append((mod._t0, e))
            x = inp[0]
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                info[name] = f"{type(mod).__name__} in{tuple(x.shape)} w{tuple(mod.weight.shape)} s{mod.stride} d{mod.dilation}"
            else:
                info[name] = f"{type(mod).__name__} in{tuple(x.shape) if hasattr(x, 'shape') else ''}"
        return f

    for name, mod in model.named_modules():
        if len(list(mod.children())) == 0 and not name.startswith("reader"):
            mod.register_forward_pre_hook(pre(name))
            mod.register_forward_hook(post(name))
    with torch.no_grad():
        for _ in range(2):
            model(ex)
        torch.cuda.synchronize()
        for k in recs:
            recs[k].clear()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(a.iters):
            model(ex)
        t1.record()
        torch.cuda.synchronize()
    print(f"total {t0.elapsed_time(t1)/a.iters:.2f} ms per step (batch {a.batch})")
    rows = []
    for k, v in recs.items():
        ms = sum(s.elapsed_time(e) for s, e in v) / a.iters
        rows.append((ms, k))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"sum of leaf modules {tot:.2f} ms")
    for ms, k in rows[:60]:
        print(f"{ms:9.3f} ms  {k:45s} {info.get(k,'')}")


if __name__ == "__main__" and "--sections" not in sys.argv:
    main()


def sections():
    """Coarse sections: reader / backbone / neck / head / predict."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--sections", action="store_true")
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    torch.manual_seed(0)
    model = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"]).cuda().eval()
    for m in (model.backbone, model.neck, model.head):
        m.to(memory_format=torch.channels_last, dtype=torch.bfloat16)
    pts = torch.from_numpy(synth.make_batch(a.config, a.batch, "uniform")).cuda()
    ex = {"points": pts, "token": [str(i) for i in range(a.batch)], "batch_size": a.batch}
    ny, nx = (int(v) for v in model.reader.grid_size)
    acc = defaultdict(float)

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    import torch.nn.functional as F
    with torch.no_grad():
        for it in range(a.iters + 2):
            marks = [("start", ev())]
            occ = torch.empty((a.batch, ny, nx), dtype=torch.uint8, device="cuda")
            canvas = model.reader.forward_dense(pts, a.batch, dtype=torch.bfloat16, occupancy=occ)
            marks.append(("reader", ev()))
            mask = occ.unsqueeze(1).to(torch.bfloat16)
            marks.append(("mask_cast", ev()))
            x = canvas
            for bi, blk in enumerate(model.backbone.blocks):
                x, mask = blk(x, mask)
                marks.append((f"backbone.stage{bi}", ev()))
            x = F.relu(model.backbone.mapping[1](model.backbone.mapping[0](x), mask)) * mask
            marks.append(("backbone.mapping", ev()))
            x = model.neck.pre_conv(x)
            marks.append(("neck.pre_conv", ev()))
            w = model.neck.weight.to(x.dtype)
            outs = [x, model.neck.conv1x1(x)]
            marks.append(("neck.conv1x1", ev()))
            for d in (1, 6, 12, 18):
                outs.append(F.conv2d(x, w, stride=1, bias=None, padding=d, dilation=d))
                marks.append((f"neck.dil{d}", ev()))
            x = model.neck.post_conv(torch.cat(outs, dim=1))
            marks.append(("neck.cat+post", ev()))
            preds = model.head(x)
            marks.append(("head", ev()))
            out = model.head.predict(ex, preds, model.post_processing)
            marks.append(("predict", ev()))
            torch.cuda.synchronize()
            if it >= 2:
                for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                    acc[n1] += e0.elapsed_time(e1)
    tot = sum(acc.values()) / a.iters
    print(f"batch {a.batch}: {tot:.2f} ms per step")
    for k, v in acc.items():
        print(f"  {k:22s} {v / a.iters:8.3f} ms")


if "--sections" in sys.argv:
    sections()
