#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration, share) of a rocprofv3 rocpd sqlite database,
in the layout of `rocprofv3 --stats`' kernel_stats table.  Usage: rocpd_summary.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for n, c, t, a, mn, mx in rows:
        lines.append(f'"{n}",{c},{t},{a:.1f},{100.0 * t / total:.2f},{mn},{mx}')
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
