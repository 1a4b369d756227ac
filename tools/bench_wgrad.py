#!/usr/bin/env python3
"""Weight gradient of the backbone's stride-1 3x3 layers on the C2 sweep cloud's real active sets (4 frames): pnx_conv3x3_wgrad_bf16
(csrc/conv_wgrad.hip) against MIOpen's dense wrw (torch.nn.grad.conv2d_weight on the same bf16 channels_last tensors)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import ops, synth  # noqa: E402
from pillarnext_amd.reader import PillarFeatureNet  # noqa: E402


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    with_miopen = not (len(sys.argv) > 2 and sys.argv[2] == "nomio")      # MIOpen's find pass over the four wrw problems takes ~2.5 minutes
    cfg = synth.CONFIGS["C2"]
    net = PillarFeatureNet(5, [64, 64], list(cfg["voxel_size"]), list(cfg["pc_range"])).cuda().eval()
    pts = torch.from_numpy(synth.make_batch("C2", B, "sweep")).cuda()
    ny, nx = (int(v) for v in net.grid_size)
    occ = torch.empty((B, ny, nx), dtype=torch.uint8, device="cuda")
    net.forward_dense(pts, B, occupancy=occ)
    m = F.max_pool2d(occ.float()[:, None], 3, 1, 1)          # stage 0: the 3 x 3 dilation of the pillars (SparseConv2d entry layer)
    torch.backends.cudnn.benchmark = True
    g = torch.Generator(device="cuda").manual_seed(0)
    for c, stride_in in ((64, 1), (128, 2), (256, 2), (256, 2)):
        if stride_in == 2:
            m = F.max_pool2d(m, 3, 2, 1)
        H, W = m.shape[2:]
        x = (torch.randn((B, c, H, W), device="cuda", generator=g) * m).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = (torch.randn((B, c, H, W), device="cuda", generator=g) * m).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        mu8 = (m[:, 0] != 0).to(torch.uint8).contiguous()
        seg = float((F.max_pool2d(m, (1, 16), (1, 16), ceil_mode=True) > 0).float().mean())
        t_hip = timed(lambda: ops.conv3x3_wgrad(x, dy, mu8))
        t_mio = timed(lambda: torch.nn.grad.conv2d_weight(x, (c, c, 3, 3), dy, stride=1, padding=1)) if with_miopen else float("nan")
        fl = 2.0 * c * c * 9 * B * H * W
        print(f"{c:3d} -> {c:3d} at {H}x{W} x {B}: active {float(m.mean()) * 100:4.1f} % of the cells, {seg * 100:4.1f} % of the 16-pixel pieces | "
              f"HIP {t_hip:7.1f} us ({fl / t_hip / 1e6:6.1f} dense-equivalent TFLOP/s)  MIOpen wrw {t_mio:7.1f} us ({fl / t_mio / 1e6:6.1f})", flush=True)


if __name__ == "__main__":
    main()
