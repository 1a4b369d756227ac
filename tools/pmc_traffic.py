#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output) of a reader run into profiles/pmc_traffic.json.

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for wide
coalesced reads on gfx950; WRITE_SIZE is taken as reported.  Two kinds of entries:
  <key>            the WHOLE reader: every dispatch of every reader kernel (and the workspace memset) summed, divided by the number of
                   reader calls (= dispatches of k_chunk_sort) -- what bench.py's `roofline.traffic` quotes
  <key>:<kernel>   one kernel, average per dispatch
usage: pmc_traffic.py fetch.csv write.csv <key> <out.json> [condition]     condition = how the profiled reader calls ran (recorded in the entry and
quoted by bench.py as roofline.traffic_condition)"""
import csv
import json
import sys

READER = ["k_clear2", "k_chunk_sort", "k_slab_totals", "k_span_carve", "k_span_pfn", "k_pfn3", "k_canvas_fill", "k_keys", "k_pack_scan", "k_scan_blocks",
          "k_scan_local", "k_bin_count", "k_bin_scatter", "k_bin_sort", "k_rank", "k_fill", "k_pfn_big", "fillBufferAligned"]


def rows(path, counter):
    return [(r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]


def main():
    fetch_csv, write_csv, key, out = sys.argv[1:5]
    cond = sys.argv[5] if len(sys.argv) > 5 else "reader calls back to back (tools/reader_ab.py)"
    f, w = rows(fetch_csv, "FETCH_SIZE"), rows(write_csv, "WRITE_SIZE")
    try:
        d = json.load(open(out))
    except Exception:
        d = {}
    for k in [k for k in d if k.startswith(key + ":")]:   # per-kernel entries of an earlier pipeline
        del d[k]
    calls = max(sum(1 for n, _ in f if "k_chunk_sort" in n) or sum(1 for n, _ in f if "k_keys" in n), 1)
    names = READER
    if any("k_chunk_sort" in n for n, _ in f):   # the span pipeline clears with its own kernel: generic names (memset, k_fill, scans) belong to other ops of the process
        names = ["k_clear2", "k_chunk_sort", "k_slab_totals", "k_span_carve", "k_span_pfn", "k_pfn3", "k_canvas_fill"]
    tot_f = sum(v for n, v in f if any(k in n for k in names))
    tot_w = sum(v for n, v in w if any(k in n for k in names))
    d[key] = {"kernel": "all reader kernels", "pipeline": "spans" if any("k_chunk_sort" in n for n, _ in f) else "bins", "reader_calls": calls, "condition": cond, "fetch_size_kib_raw_per_call": tot_f / calls, "write_size_kib_per_call": tot_w / calls,
              "hbm_bytes_per_launch": int((2 * tot_f + tot_w) / calls * 1024), "note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as reported"}
    for k in names:
        fv = [v for n, v in f if k in n]
        wv = [v for n, v in w if k in n]
        if fv or wv:
            fa, wa = sum(fv) / max(len(fv), 1), sum(wv) / max(len(wv), 1)
            d[f"{key}:{k}"] = {"dispatches": [len(fv), len(wv)], "fetch_size_kib_raw": fa, "write_size_kib": wa, "hbm_bytes_per_dispatch": int((2 * fa + wa) * 1024)}
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d[key]))


if __name__ == "__main__":
    main()
