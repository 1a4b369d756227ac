#!/usr/bin/env python3
"""Where the spread of k_span_pfn's durations comes from ("min / max 412 / 593 us over 24 calls is a wide spread nobody explained").
bench.py and tools/reader_ab.py rotate FOUR different frame batches through the reader; this groups the dispatches of a rocprofv3 --kernel-trace
(rocpd SQLite) by their position in that rotation and prints, per slot, the duration of k_span_pfn and of the zero-fill kernel that runs beside it,
plus how long the two overlapped.  usage: python tools/span_spread.py <dir or .db> [rotation = 4]"""
import glob
import sqlite3
import sys


def main():
    path = sys.argv[1]
    rot = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    db = path if path.endswith(".db") else glob.glob(path + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    span = list(cur.execute("select start, end from kernels where name like '%k_span_pfn%' order by start"))
    fill = list(cur.execute("select start, end from kernels where name like '%k_canvas_fill_bytes%' order by start"))
    n = min(len(span), len(fill))
    span, fill = span[-(n // rot) * rot:], fill[-(n // rot) * rot:]     # whole rotations, the most recent ones (behind the warm-up)
    print(f"{len(span)} reader calls, rotation of {rot} frame batches\n")
    print("| slot | calls | k_span_pfn us (min / mean / max) | k_canvas_fill_bytes us (min / mean / max) | overlap us (mean) |\n|---|---|---|---|---|")
    for s in range(rot):
        sp = [(e - b) / 1e3 for b, e in span[s::rot]]
        fl = [(e - b) / 1e3 for b, e in fill[s::rot]]
        ov = [max(0, min(e1, e2) - max(b1, b2)) / 1e3 for (b1, e1), (b2, e2) in zip(span[s::rot], fill[s::rot])]
        print(f"| {s} | {len(sp)} | {min(sp):.0f} / {sum(sp) / len(sp):.0f} / {max(sp):.0f} | {min(fl):.0f} / {sum(fl) / len(fl):.0f} / {max(fl):.0f} | {sum(ov) / len(ov):.0f} |")
    allsp = [(e - b) / 1e3 for b, e in span]
    print(f"\nall calls: k_span_pfn min {min(allsp):.0f}, max {max(allsp):.0f} us; spread inside a slot vs between slots is what the table separates")


if __name__ == "__main__":
    main()
