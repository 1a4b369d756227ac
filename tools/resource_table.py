#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in the library (hipcc -Rpass-analysis=kernel-resource-usage) as one markdown table.
usage: python tools/resource_table.py > profiles/rNN_kernel_resources.md      (no GPU needed; same flags as pillarnext_amd/build.py)"""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
print("| source | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | occupancy waves/SIMD | LDS B/block |")
print("|---|---|---|---|---|---|---|---|")
for f in sorted(glob.glob("pillarnext_amd/csrc/*.hip")):
    extra = []
    if os.path.basename(f) in ("pfn_v3.hip", "pfn_spans.hip"):
        extra = ["-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
    elif os.path.basename(f) == "conv3x3.hip":
        extra = ["-fno-honor-nans"]
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude", "-Ipillarnext_amd/csrc"] + extra
                       + ["-c", f, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}

    def flush():
        if cur.get("name"):
            n = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            n = n.replace("(anonymous namespace)::", "")
            n = re.sub(r"^void ", "", n).split("(")[0][:72]
            print(f"| {os.path.basename(f)} | `{n}` | {cur.get('VGPRs')} | {cur.get('AGPRs')} | {cur.get('TotalSGPRs')} | {cur.get('ScratchSize [bytes/lane]')} | "
                  f"{cur.get('Occupancy [waves/SIMD]')} | {cur.get('LDS Size [bytes/block]')} |")

    for line in p.stderr.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", line)
        if m:
            flush()
            cur.clear()
            cur["name"] = m.group(1)
            continue
        m = re.search(r"remark: +([A-Za-z \[\]/]+): (\d+)", line)
        if m:
            cur[m.group(1).strip()] = m.group(2)
    flush()
