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


def k2_json(fetch_db, write_db, out_path, match="dq"):
    """HBM bytes per launch of the K2 kernel from two PMC passes: FETCH_SIZE (KB, reads 1/2 of a wide coalesced
    stream on gfx950 -> x2, MI355X_MICROARCH.md HBM section) + WRITE_SIZE (KB, 1:1)."""
    import json

    def avg(db, counter):
        con = sqlite3.connect(db)
        rows = con.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
                           "where counter_name = ? group by kernel_name", (counter,)).fetchall()
        rows = [r for r in rows if match in r[0]]
        rows.sort(key=lambda r: -r[1])
        return rows[0] if rows else None
    f, w = avg(fetch_db, "FETCH_SIZE"), avg(write_db, "WRITE_SIZE")
    out = {"kernel": f[0][:120], "launches": f[1], "FETCH_SIZE_KB_avg": f[2], "WRITE_SIZE_KB_avg": w[2],
           "fetch_correction": 2.0, "hbm_bytes_per_launch": int((2.0 * f[2] + w[2]) * 1024),
           "avg_kernel_ns_under_pmc": f[3]}
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--k2-json":
        k2_json(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        main(sys.argv[1:])
