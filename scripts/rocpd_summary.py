#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd sqlite output (kernel-trace [+ --pmc]) as a text table: per kernel calls, total,
average/min/max duration and, when present, per-dispatch average of each PMC counter.
usage: rocpd_summary.py <results.db> [more.db ...]"""
import sqlite3
import sys


def main(paths):
    for p in paths:
        con = sqlite3.connect(p)
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        print(f"# {p}")
        rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                           "from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print(f"{'kernel':<90} {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
        for n, c, s, a, mn, mx in rows:
            print(f"{n[:90]:<90} {c:>7} {s/1e3:>12.1f} {a/1e3:>10.3f} {mn/1e3:>10.3f} {mx/1e3:>10.3f} {100*s/tot:>6.2f}")
        try:
            pc = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                             "group by kernel_name, counter_name order by 1, 2").fetchall()
        except sqlite3.Error as e:
            pc = []
        if pc:
            print(f"{'kernel':<90} {'counter':<20} {'dispatches':>10} {'avg_value':>16}")
            for n, cn, c, v in pc:
                print(f"{n[:90]:<90} {cn:<20} {c:>10} {v:>16.2f}")
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
