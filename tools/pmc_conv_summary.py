#!/usr/bin/env python3
"""Per-kernel ratios from the counter_collection.csv files of tools/pmc_conv.sh (SQ counters summed over the dispatches of a kernel)."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]
print("| run | kernel | dispatches | parked (WAIT_ANY) | issue stall (WAIT_INST_ANY) | of which LDS issue | issuing (ACTIVE_INST_ANY) | MFMA busy cycles / (4 x wave quad-cycles) | LDS bank-conflict / LDS active |")
print("|---|---|---|---|---|---|---|---|---|")
for d in sorted(glob.glob(os.path.join(out, "*/"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(float))
    nd = defaultdict(set)
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        if "k_conv3x3" not in k:
            continue
        k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        nd[k].add(row["Dispatch_Id"])
    for k, c in acc.items():
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0) or 1.0
        print(f"| {os.path.basename(d.rstrip('/'))} | `{k}` | {len(nd[k])} | {100 * c.get('SQ_WAIT_ANY', 0) / wc:.1f} % | {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:.1f} % | "
              f"{100 * c.get('SQ_WAIT_INST_LDS', 0) / wc:.1f} % | {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.1f} % | "
              f"{100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * wc):.1f} % | {100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / lds:.1f} % |")
print("\nSQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles (guide: 32 per 32x32x16 bf16 MFMA);")
print("with W resident waves per SIMD the MFMA pipe's own utilisation is W x the column above (2 waves per SIMD in these kernels).")
