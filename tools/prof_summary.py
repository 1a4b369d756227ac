#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel table (markdown)."""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    only = sys.argv[3] if len(sys.argv) > 3 else None   # regex: list only the kernels whose name matches (percentages stay of the total)
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = list(cur.execute("select name, count(*), avg(end-start)/1000.0, sum(end-start)/1000.0, min(end-start)/1000.0, max(end-start)/1000.0 "
                            "from kernels group by name order by 4 desc"))
    total = sum(r[3] for r in rows)
    if only:
        import re

        rows = [r for r in rows if re.search(only, r[0])]
    print(f"| kernel | calls | avg us | min us | max us | total us | % |\n|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        print(f"| `{r[0][:110]}` | {r[1]} | {r[2]:.2f} | {r[4]:.2f} | {r[5]:.2f} | {r[3]:.1f} | {100*r[3]/total:.1f} |")
    print(f"\ntotal kernel time {total/1000:.2f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main()
