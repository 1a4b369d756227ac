#!/usr/bin/env python3
"""Numerical feasibility study (CPU, numpy + the C oracle; no GPU): would the PFN's second layer (64 -> 64, 93 % of the PFN
FLOPs) stay inside the 1e-4 feature tolerance if it ran on bf16 MFMAs with hi/lo splits of activations and weights
(x ~ x_hi + x_lo, three products, fp32 accumulation) instead of fp32 MFMAs?  Prints the error of the final pillar features
against the fp32 oracle for 1-term (plain bf16), 3-term and 4-term variants."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402  (a study tool, not product code)
from pillarnext_amd import synth  # noqa: E402


def bf16(x):
    """round-to-nearest-even fp32 -> bf16 -> fp32"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def fold(L, eps=1e-3):
    a = (L["gamma"] / np.sqrt(L["var"].astype(np.float64) + eps)).astype(np.float32)
    return (L["W"] * a[:, None]).astype(np.float32), (L["beta"] - L["mean"] * a).astype(np.float32)


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
    cfg = synth.CONFIGS[cfgname]
    layers = synth.pfn_params()
    pts = synth.make_batch(cfgname, 1, "sweep", n=n)
    vox = O.voxelize(pts, cfg["pc_range"], cfg["voxel_size"])
    feat = O.decorate(pts, vox, cfg["pc_range"], cfg["voxel_size"])          # (N', 10) fp32
    inv, P = vox["inv"], vox["P"]
    W0, s0 = fold(layers[0])
    W1, s1 = fold(layers[1])
    h0 = np.maximum(feat @ W0.T + s0, 0).astype(np.float32)                   # layer 0 stays on fp32 MFMA
    g0 = np.zeros((P, 32), np.float32)
    np.maximum.at(g0, inv, h0)
    u = np.concatenate([h0, g0[inv]], 1)                                      # (N', 64)
    ref = u.astype(np.float64) @ W1.T.astype(np.float64)                      # exact pre-activation

    def pillar_out(pre):
        y = np.maximum(pre + s1, 0)
        g = np.zeros((P, 64), np.float64)
        np.maximum.at(g, inv, y)
        return g

    gref = pillar_out(ref)
    fp32 = u @ W1.T
    uh, wh = bf16(u), bf16(W1)
    ul, wl = bf16(u - uh), bf16(W1 - wh)
    variants = {
        "fp32 (reference kernel)": fp32.astype(np.float64),
        "bf16 x1 (hi*hi)": (uh @ wh.T).astype(np.float64),
        "bf16 x3 (hi*hi + hi*lo + lo*hi)": (uh @ wh.T + uh @ wl.T + ul @ wh.T).astype(np.float64),
        "bf16 x4 (+ lo*lo)": (uh @ wh.T + uh @ wl.T + ul @ wh.T + ul @ wl.T).astype(np.float64),
    }
    print(f"{cfgname}: N'={len(inv)} P={P}; |u| max {np.abs(u).max():.2f}, feat_max max {gref.max():.2f}; tolerance of the parity tests: rtol=atol=1e-4")
    for name, pre in variants.items():
        g = pillar_out(pre)
        err = np.abs(g - gref)
        viol = err > 1e-4 + 1e-4 * np.abs(gref)
        print(f"  {name:36s} max abs err {err.max():.3e}   max err/(1e-4+1e-4|ref|) {np.max(err / (1e-4 + 1e-4 * np.abs(gref))):.3f}   violations {int(viol.sum())}")


if __name__ == "__main__":
    main()
