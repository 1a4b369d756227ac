#!/usr/bin/env python3
"""Quick parity of the producer / consumer convolution kernel (PNX_CONV_PC) against fp32 torch on random masks, small shapes first (a hang shows early)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PNX_CONV_PC", "7")
from pillarnext_amd import ops

def one(cin, cout, B, H, W, density, res, tiles, ws):
    g = torch.Generator(device="cuda").manual_seed(cin + cout + H)
    x = torch.randn((B, cin, H, W), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cout, cin, 3, 3), device="cuda", generator=g) / 24).to(torch.bfloat16)
    bias = torch.randn((cout,), device="cuda", generator=g)
    mask = None
    if density < 1:
        mask = (torch.rand((B, H, W), device="cuda", generator=g) < density).to(torch.uint8)
        if density < 0.2:  # whole rows / tiles empty
            mask[:, ::3] = 0
            mask[:, :, W // 2:] = 0
        x = (x * mask.unsqueeze(1)).contiguous(memory_format=torch.channels_last)
    r = (x[:, :cout] * 1.0).contiguous(memory_format=torch.channels_last) if res else None
    wf = ops.conv3x3_pack_weights(w)
    out = ops.conv3x3_workspace(B, cout, H, W, "cuda") if (ws and mask is not None) else None
    tl = ops.conv_tile_list(mask, [out[1]], ops.conv_tile_rows(cin, cout, 1)) if (tiles and out is not None) else None
    for rep in range(2):   # the second call runs on a dirty workspace
        y = ops.conv3x3_masked(x, wf, bias, cout, 1, mask, r, True, out=out, tiles=tl)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), bias, 1, 1)
    if r is not None:
        ref = ref + r.float()
    ref = torch.relu(ref)
    if mask is not None:
        ref = ref * mask.unsqueeze(1)
    err = (y.float() - ref).abs().max().item()
    tol = 0.02 * max(1.0, ref.abs().max().item())
    print(f"{cin}->{cout} B{B} {H}x{W} density {density} res {res} tiles {tiles} ws {ws}: max err {err:.4f} (tol {tol:.3f})", flush=True)
    return err <= tol

ok = True
for cin, cout in ((64, 64), (128, 128), (256, 256), (64, 384)):
    for (B, H, W) in ((1, 16, 32), (1, 40, 70), (2, 97, 131)):
        for density, res, tiles, ws in ((1.0, False, False, False), (0.5, True, False, False), (0.1, True, True, True), (0.1, False, False, True)):
            if res and cout != cin:
                res = False
            ok &= one(cin, cout, B, H, W, density, res, tiles, ws)


def moving(cin, cout, B, H, W):
    """A workspace that goes stale: frame 0 is active in the left half, frame 1 in the right half -- the tile list of frame 1 then holds the left half's tiles with
    NO active row (zero-fill only) in front of the active ones, more than 3 tiles per workgroup, so the ticketed schedule and the null-tile path of the ring
    are both exercised (a lost first ticket showed here)."""
    g = torch.Generator(device="cuda").manual_seed(99 + cin)
    w = (torch.randn((cout, cin, 3, 3), device="cuda", generator=g) / 24).to(torch.bfloat16)
    wf = ops.conv3x3_pack_weights(w)
    bias = torch.randn((cout,), device="cuda", generator=g)
    out = ops.conv3x3_workspace(B, cout, H, W, "cuda")
    good = True
    for frame in range(3):
        mask = (torch.rand((B, H, W), device="cuda", generator=g) < 0.15).to(torch.uint8)
        if frame % 2 == 0:
            mask[:, :, W // 2:] = 0
        else:
            mask[:, :, :W // 2] = 0
        x = (torch.randn((B, cin, H, W), device="cuda", generator=g) * mask.unsqueeze(1)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        tl = ops.conv_tile_list(mask, [out[1]], ops.conv_tile_rows(cin, cout, 1))
        y = ops.conv3x3_masked(x, wf, bias, cout, 1, mask, None, True, out=out, tiles=tl)
        ref = torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), bias, 1, 1)) * mask.unsqueeze(1)
        err = (y.float() - ref).abs().max().item()
        stale = int(tl[1].item())
        print(f"moving mask {cin}->{cout} B{B} {H}x{W} frame {frame}: listed tiles {stale}, max err {err:.4f}", flush=True)
        good &= err <= 0.02 * max(1.0, ref.abs().max().item())
    return good


ok &= moving(64, 64, 4, 384, 384)
ok &= moving(128, 128, 2, 384, 384)
ok &= moving(256, 256, 2, 384, 384)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
