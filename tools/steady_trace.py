#!/usr/bin/env python3
"""Per-step kernel breakdown of the steady state of a rocprofv3 --kernel-trace of bench.py (rocpd SQLite).

Steps are delimited by a kernel that runs ONCE PER DETECTOR STEP -- k_gather_kept, the last kernel of the decoder (argv[4] names another one, e.g. the
optimizer's kernel for a training trace) -- not by the reader's k_chunk_sort: bench.py runs reader-only calls behind its timed loop (roofline.back_to_back)
and a reader delimiter would average THAT window (round 4's committed r04_bench_steady_trace.md did: "0.68 ms/step, 7 dispatches").  The last `n`
complete steps are averaged, which skips the MIOpen find-mode searches of the warm-up; traces without the delimiter fall back to k_chunk_sort / k_keys."""
import glob
import sqlite3
import sys


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    db = path if path.endswith(".db") else glob.glob(path + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    delim = sys.argv[4] if len(sys.argv) > 4 else "k_gather_kept"
    starts = []
    for name in (delim, "k_chunk_sort", "k_keys"):
        starts = [r[0] for r in cur.execute("select start from kernels where name like ? order by start", (f"%{name}%",))]
        if len(starts) > nsteps:
            delim = name
            break
    # the LAST delimiter belongs to the final batch, whose decoder is flushed at once instead of waiting behind a next reader (models._DeferredDecode):
    # that interval is short, so the window ends one delimiter earlier
    if len(starts) > nsteps + 2:
        starts = starts[:-1]
    t0, t1 = starts[-nsteps - 1], starts[-1]
    rows = list(cur.execute("select name, count(*), avg(end-start)/1000.0, sum(end-start)/1000.0 from kernels where start >= ? and start < ? "
                            "group by name order by 4 desc", (t0, t1)))
    tot = sum(r[3] for r in rows)
    print(f"steps delimited by `{delim}`; steady state over {nsteps} steps: {(t1 - t0) / 1e6 / nsteps:.2f} ms/step wall, {tot / 1000 / nsteps:.2f} ms/step kernel-busy, "
          f"{sum(r[1] for r in rows) / nsteps:.0f} dispatches/step\n")
    print("| us/step | calls/step | avg us | kernel |\n|---|---|---|---|")
    for r in rows[:top]:
        print(f"| {r[3] / nsteps:.1f} | {r[1] / nsteps:.1f} | {r[2]:.1f} | `{r[0][:120]}` |")


if __name__ == "__main__":
    main()
