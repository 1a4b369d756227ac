#!/usr/bin/env python3
"""Per-step kernel breakdown of the steady state of a rocprofv3 --kernel-trace of bench.py (rocpd SQLite).

Steps are delimited by the reader's grouping kernel (k_chunk_sort; k_keys for the older pipelines); the last `n` complete steps are averaged, which skips the
MIOpen find-mode searches of the warm-up."""
import glob
import sqlite3
import sys


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    db = path if path.endswith(".db") else glob.glob(path + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    starts = [r[0] for r in cur.execute("select start from kernels where name like '%k_chunk_sort%' order by start")]
    if len(starts) <= nsteps:
        starts = [r[0] for r in cur.execute("select start from kernels where name like '%k_keys%' order by start")]
    t0, t1 = starts[-nsteps - 1], starts[-1]
    rows = list(cur.execute("select name, count(*), avg(end-start)/1000.0, sum(end-start)/1000.0 from kernels where start >= ? and start < ? "
                            "group by name order by 4 desc", (t0, t1)))
    tot = sum(r[3] for r in rows)
    print(f"steady state over {nsteps} steps: {(t1 - t0) / 1e6 / nsteps:.2f} ms/step wall, {tot / 1000 / nsteps:.2f} ms/step kernel-busy, "
          f"{sum(r[1] for r in rows) / nsteps:.0f} dispatches/step\n")
    print("| us/step | calls/step | avg us | kernel |\n|---|---|---|---|")
    for r in rows[:top]:
        print(f"| {r[3] / nsteps:.1f} | {r[1] / nsteps:.1f} | {r[2]:.1f} | `{r[0][:120]}` |")


if __name__ == "__main__":
    main()
