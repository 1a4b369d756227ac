#!/usr/bin/env python3
"""Times the REFERENCE's own reader on the CPU of the build container -- BASELINE.md section 3.1's protocol: the unmodified
det3d/models/readers/pillar_encoder.py (PillarFeatureNet.forward :174-182 = PillarNet.forward :78-125 + 2 x PFNLayer + scatter_max) with
the torch_scatter restatement of oracle/gen_golden.py, eval mode, one sample per call, 5 warm-up + 20 timed iterations, median;
torch.set_num_threads in {1, nproc}.  Needs /root/reference: runs ONLY in the build container, never on the GPU box
(bench.py's cpu_baseline leg there times the C port, oracle/pnx_oracle.c, on the same clouds).

    python tools/time_reference_reader.py [--configs C1,C2] [--dists sweep,uniform] [--iters 20]   ->  one JSON line per row
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden as G  # noqa: E402  (import machinery only: torch_scatter / numba restatements)
from pillarnext_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C1,C2")
    ap.add_argument("--dists", default="sweep,uniform")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    assert os.path.isdir(G.REF), "the reference tree is not mounted: build container only"
    G.install_torch_scatter()
    G.install_numba_identity()
    sys.path.insert(0, G.REF)
    from det3d.models.readers.pillar_encoder import PillarFeatureNet  # the reference, unmodified

    ncore = os.cpu_count() or 1
    for config in a.configs.split(","):
        cfg = synth.CONFIGS[config]
        net = PillarFeatureNet(5, [64, 64], list(cfg["voxel_size"]), list(cfg["pc_range"])).eval()
        for dist in a.dists.split(","):
            pts = torch.from_numpy(synth.make_batch(config, 1, dist))
            for thr in (1, ncore):
                torch.set_num_threads(thr)
                ts_full, ts_vox = [], []
                with torch.no_grad():
                    for i in range(a.warmup + a.iters):
                        t0 = time.perf_counter()
                        net.voxelization(pts)
                        t1 = time.perf_counter()
                        fm, coords, grid = net(pts)
                        t2 = time.perf_counter()
                        if i >= a.warmup:
                            ts_vox.append(t1 - t0)
                            ts_full.append(t2 - t1)
                print(json.dumps({"config": config, "dist": dist, "points": int(pts.shape[0]), "pillars": int(fm.shape[0]), "threads": thr,
                                  "voxelize_ms_median": round(statistics.median(ts_vox) * 1e3, 1),
                                  "reader_ms_median": round(statistics.median(ts_full) * 1e3, 1),
                                  "frames_per_s": round(1.0 / statistics.median(ts_full), 3), "iters": a.iters, "warmup": a.warmup,
                                  "host_cores": ncore}), flush=True)


if __name__ == "__main__":
    main()
