#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd database, restricted to dispatches whose
grid is at least --min-grid work-items.  Usage: pmc_query.py results.db [--min-grid N] [--like pattern]"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--min-grid", type=int, default=0)
    ap.add_argument("--like", default="%")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    rows = db.execute(
        "select kernel_name, counter_name, avg(value), count(*), avg(end-start) from counters_collection "
        "where grid_size >= ? and kernel_name like ? group by kernel_name, counter_name", (a.min_grid, a.like)).fetchall()
    agg = {}
    for k, cn, val, n, dur in rows:
        agg.setdefault(k, {"_n": n, "_dur_us": dur / 1e3})[cn] = val
    for k, d in agg.items():
        print(k[:100])
        print("   dispatches %d  avg duration %.1f us" % (d.pop("_n"), d.pop("_dur_us")))
        for cn in sorted(d):
            print("   %-36s %16.0f" % (cn, d[cn]))


if __name__ == "__main__":
    main()
