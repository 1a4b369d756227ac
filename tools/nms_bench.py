#!/usr/bin/env python3
"""Stand-alone batched rotated NMS on SURVEY 8(d)'s box sets (bench.py's nms_us leg by itself): one JSON line.  Profile it with tools/prof_cmd.sh."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.nms_bench(torch.device("cuda", 0))))
