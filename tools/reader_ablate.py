#!/usr/bin/env python3
"""Timing ablations of the binned reader (results are WRONG under these knobs; timing only).  PNX_PFN_DBG / PNX_SORT_DBG."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import _lib, synth
from pillarnext_amd.reader import PillarFeatureNet

cfg = synth.CONFIGS["C2"]
B = 8
net = PillarFeatureNet(5, [64, 64], list(cfg["voxel_size"]), list(cfg["pc_range"])).cuda().eval()
batches = [torch.from_numpy(synth.make_batch("C2", B, sys.argv[1] if len(sys.argv) > 1 else "sweep", frame0=k * B)).cuda() for k in range(4)]
ny, nx = (int(v) for v in net.grid_size)
out = torch.empty((B, 64, ny, nx), dtype=torch.bfloat16, device="cuda", memory_format=torch.channels_last)
occ = torch.empty((B, ny, nx), dtype=torch.uint8, device="cuda")
L = _lib.lib()
os.environ["PNX_READER_FUSE"] = "0"
for name, env in [("baseline", {}), ("pfn: no stores", {"PNX_PFN_DBG": "1"}), ("pfn: no scans", {"PNX_PFN_DBG": "2"}), ("pfn: no layer-1 MFMA", {"PNX_PFN_DBG": "4"}),
                  ("pfn: no stores/scans", {"PNX_PFN_DBG": "3"}), ("pfn: nothing but layer 0", {"PNX_PFN_DBG": "7"}),
                  ("sort: no fp64 sums", {"PNX_SORT_DBG": "1"})]:
    for k in ("PNX_PFN_DBG", "PNX_SORT_DBG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for i in range(3):
        net.forward_dense(batches[i % 4], B, out=out, occupancy=occ)
    torch.cuda.synchronize()
    L.pnx_profile_begin(20)
    for i in range(20):
        net.forward_dense(batches[i % 4], B, out=out, occupancy=occ)
    torch.cuda.synchronize()
    r, c, ns = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int32(0)
    L.pnx_profile_end(ctypes.byref(r), ctypes.byref(c), ctypes.byref(ns))
    print(f"{name:28s} reader {r.value:8.1f}  voxelize {L.pnx_profile_last_voxelize_us():7.1f}  pfn {L.pnx_profile_last_pfn_us():7.1f}  fill {c.value:7.1f}", flush=True)
