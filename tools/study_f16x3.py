#!/usr/bin/env python3
"""Numerical study behind the fp16x3 form of PFN layer 1 (pillarnext_amd/csrc/pfn_v3.hip), numpy only.

u (non-negative, pre-scaled by 2^SU) and W1' (pre-scaled by 2^SW) are each split into fp16 hi + fp16 lo (22 significant bits);
hi*hi + hi*lo + lo*hi is accumulated in fp32 by v_mfma_f32_32x32x16_f16.  Here the same three products are formed in fp64 (the MFMA's
fp32 accumulation adds the usual ~1e-7 relative on top) and compared with the exact product and with an fp32 matmul."""
import numpy as np

SU, SW = 6, 8


def rtz16(x):
    h = x.astype(np.float16)
    bad = np.abs(h.astype(np.float32)) > np.abs(x)
    hb = h.view(np.uint16).copy()
    hb[bad] -= 1
    return hb.view(np.float16)


def main():
    rng = np.random.default_rng(0)
    f = lambda a: a.astype(np.float64)  # noqa: E731
    for wscale, uscale in [(0.125, 3.0), (1.0, 30.0), (0.01, 0.05), (5.0, 200.0)]:
        W = (rng.uniform(-1, 1, (64, 64)) * wscale).astype(np.float32)
        U = np.maximum(rng.normal(0, 1, (64, 4096)) * uscale, 0).astype(np.float32)
        Ws, Us = (W * 2 ** SW).astype(np.float32), (U * 2 ** SU).astype(np.float32)
        Wh = Ws.astype(np.float16)
        Wl = (Ws - Wh.astype(np.float32)).astype(np.float16)
        Uh = rtz16(Us)
        Ul = rtz16(Us - Uh.astype(np.float32))
        approx = (f(Wh) @ f(Uh) + f(Wh) @ f(Ul) + f(Wl) @ f(Uh)) * 2.0 ** -(SW + SU)
        exact = f(W) @ f(U)
        fp32 = f(W @ U)
        tol = 1e-4 + 1e-4 * np.abs(exact)
        print(f"|W|~{wscale:<6} u~{uscale:<6} max|y| {np.abs(exact).max():10.2f}  fp16x3 max err {np.abs(approx - exact).max():.2e} "
              f"({np.max(np.abs(approx - exact) / tol):.3f} of the test tolerance)  fp32 matmul max err {np.abs(fp32 - exact).max():.2e}  "
              f"scaled u max {Us.max():.0f} (fp16 max 65504; >= 60000 falls back to fp32 MFMA)")


if __name__ == "__main__":
    main()
