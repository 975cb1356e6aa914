#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd database, restricted to dispatches whose
grid is at least --min-grid work-items.  --full-size keeps, per kernel, only the dispatches that ran at least
0.88 x the kernel's median duration: the persistent GEMMs launch the same grid for every problem size, and the
bench's dominant launches (9 of 10 per forward) are the median; the last layer's masked-rows-only launches are
0.3x / 0.78x of it.  Usage: pmc_query.py results.db [--min-grid N] [--like pattern] [--full-size]"""
import argparse
import sqlite3
import statistics


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--min-grid", type=int, default=0)
    ap.add_argument("--like", default="%")
    ap.add_argument("--full-size", action="store_true")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    rows = db.execute(
        "select kernel_name, counter_name, value, end-start from counters_collection "
        "where grid_size >= ? and kernel_name like ?", (a.min_grid, a.like)).fetchall()
    per = {}
    for k, cn, val, dur in rows:
        per.setdefault(k, {}).setdefault(cn, []).append((dur, val))
    for k, d in per.items():
        print(k[:100])
        first = True
        for cn in sorted(d):
            v = d[cn]
            if a.full_size:
                med = statistics.median(x[0] for x in v)
                v = [x for x in v if x[0] >= 0.88 * med]
            if first:
                print("   dispatches %d of %d  avg duration %.1f us" % (len(v), len(d[cn]), sum(x[0] for x in v) / len(v) / 1e3))
                first = False
            print("   %-36s %16.0f" % (cn, sum(x[1] for x in v) / len(v)))


if __name__ == "__main__":
    main()
