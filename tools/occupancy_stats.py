#!/usr/bin/env python3
"""Active cells / row segments / pixel groups per backbone stage on the synthetic C2 sweep cloud (CPU, numpy): what the masked convolution
kernels of csrc/conv3x3.hip see.  Run from the repo root: PYTHONPATH=. python tools/occupancy_stats.py [sweep|uniform]"""
import sys
import numpy as np
from pillarnext_amd import synth
DIST = sys.argv[1] if len(sys.argv) > 1 else "sweep"   # sweep | uniform (bench.py's worst case: value_uniform)
print(f"# cloud = {DIST}")
cfg = synth.CONFIGS["C2"]
pts = synth.make_batch("C2", 1, DIST)
r = cfg["pc_range"]; vs = cfg["voxel_size"]
x = np.floor((pts[:,1]-r[0])/vs[0]).astype(int); y = np.floor((pts[:,2]-r[1])/vs[1]).astype(int)
ok = (x>=0)&(x<1440)&(y>=0)&(y<1440)&(pts[:,3]>=r[2])&(pts[:,3]<r[5])
m = np.zeros((1440,1440),bool); m[y[ok],x[ok]] = True
def stats(m, th, name):
    H,W = m.shape
    Hp, Wp = -(-H//th)*th, -(-W//32)*32
    mp = np.zeros((Hp,Wp),bool); mp[:H,:W]=m
    t = mp.reshape(Hp//th, th, Wp//32, 32)
    seg = t.any(3)           # (ty, th, tx)
    P = t.sum((1,3))         # per tile
    act_tiles = (P>0).sum()
    nseg = seg.sum()
    ntile = np.ceil(P/32).sum()
    print(f"{name}: px {m.mean():.3f} active; tiles {act_tiles}/{P.size} = {act_tiles/P.size:.2f}; active segs {nseg} ({nseg/seg.size:.3f}); gather N-tiles {int(ntile)}; ratio {nseg/ntile:.2f}; mean P/active tile {P[P>0].mean():.1f}")
stats(m, 16, "stage0 1440^2 th16"); stats(m, 8, "stage0 th8")
def pool(m):  # maxpool3 s2 p1
    H,W=m.shape; p=np.zeros((H+2,W+2),bool); p[1:-1,1:-1]=m
    o=np.zeros((H//2,W//2),bool)
    for dy in range(3):
        for dx in range(3):
            o |= p[dy:dy+H:2, dx:dx+W:2][:H//2,:W//2]
    return o
m1=pool(m); stats(m1, 8, "stage1 720^2 th8"); stats(m1,16,"stage1 th16")
m2=pool(m1); stats(m2, 8, "stage2 360^2 th8")
m3=pool(m2); stats(m3, 8, "stage3 180^2 th8")
def dil(m):
    H,W=m.shape; p=np.zeros((H+2,W+2),bool); p[1:-1,1:-1]=m
    o=np.zeros((H,W),bool)
    for dy in range(3):
        for dx in range(3):
            o |= p[dy:dy+H, dx:dx+W]
    return o
md = dil(m)
stats(md, 16, "stage0 DILATED th16")
stats(md.T.copy(), 16, "stage0 DILATED transposed th16")
m1d = pool(md); stats(m1d, 8, "stage1 (from dilated) th8")
m2d = pool(m1d); stats(m2d, 8, "stage2 (from dilated)")
