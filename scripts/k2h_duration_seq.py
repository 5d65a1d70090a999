#!/usr/bin/env python3
"""Per-dispatch durations of the headline kernel out of a rocprofv3 kernel trace (rocpd sqlite), in dispatch order: which launches hit
the minimum (profiles/r04F: min 3.72 us, average 5.19) and why not all?  Prints the histogram, the durations by position in the run
(first launches after an idle GPU vs steady state), by ring copy (i mod ring: an address effect would show as a period), the gap between
consecutive kernels (end -> next start) and the autocorrelation of the sequence.
usage: k2h_duration_seq.py <trace_results.db> [ring=96] [match=dq_h_kernel]"""
import sqlite3
import sys

import numpy as np


def main():
    db = sys.argv[1]
    ring = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    match = sys.argv[3] if len(sys.argv) > 3 else "dq_h_kernel"
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    rows = [(s, e) for n, s, e in rows if match in n]
    if not rows:
        print("no dispatches of", match)
        return
    st = np.array([r[0] for r in rows], dtype=np.float64)
    en = np.array([r[1] for r in rows], dtype=np.float64)
    dur = (en - st) / 1e3
    gap = (st[1:] - en[:-1]) / 1e3
    n = len(dur)
    print(f"# {db}: {n} dispatches of {match}")
    print(f"duration us: mean {dur.mean():.3f}  median {np.median(dur):.3f}  min {dur.min():.3f}  p10 {np.percentile(dur, 10):.3f}  p90 {np.percentile(dur, 90):.3f}  max {dur.max():.3f}")
    print(f"gap end->next start us: median {np.median(gap):.3f}  p10 {np.percentile(gap, 10):.3f}  p90 {np.percentile(gap, 90):.3f}  (negative = overlap)")
    period = (st[1:] - st[:-1]) / 1e3
    print(f"start->start period us: median {np.median(period):.3f}  mean of the middle 80 % {np.mean(np.sort(period)[n // 10: -n // 10]):.3f}")
    edges = np.arange(np.floor(dur.min() * 4) / 4, min(dur.max(), dur.min() + 4) + 0.25, 0.25)
    hist, _ = np.histogram(dur, bins=edges)
    print("histogram (0.25 us bins):")
    for a, c in zip(edges[:-1], hist):
        print(f"  {a:5.2f}-{a + 0.25:5.2f}  {c:5d}  {'#' * int(60 * c / max(hist.max(), 1))}")
    print("by position in the run (mean duration of consecutive blocks of n/10 dispatches):")
    for k in range(10):
        blk = dur[k * n // 10:(k + 1) * n // 10]
        print(f"  {k * n // 10:5d}..{(k + 1) * n // 10 - 1:5d}: mean {blk.mean():.3f}  min {blk.min():.3f}")
    print("first 12 dispatches:", " ".join(f"{v:.2f}" for v in dur[:12]))
    fast = dur < dur.min() + 0.3
    print(f"dispatches within 0.3 us of the minimum: {int(fast.sum())} of {n}; their gaps BEFORE them (us): "
          f"median {np.median(gap[fast[1:]]) if fast[1:].any() else float('nan'):.3f} vs all {np.median(gap):.3f}")
    # duration against the idle time in front of the dispatch
    order = np.argsort(gap)
    q = len(gap) // 4
    for name, idx in (("shortest gaps", order[:q]), ("longest gaps", order[-q:])):
        print(f"  quarter with the {name} in front (median gap {np.median(gap[idx]):.3f} us): mean duration {dur[1:][idx].mean():.3f}")
    byring = [dur[i::ring].mean() for i in range(min(ring, n))]
    print(f"by ring copy (i mod {ring}): min {min(byring):.3f} max {max(byring):.3f} spread {max(byring) - min(byring):.3f}  (std over copies {np.std(byring):.3f}; "
          f"std expected from noise alone {dur.std() / np.sqrt(max(n // ring, 1)):.3f})")
    d0 = dur - dur.mean()
    ac = [float((d0[:-k] * d0[k:]).mean() / d0.var()) for k in (1, 2, 3, 4, 8)]
    print("autocorrelation at lags 1, 2, 3, 4, 8:", " ".join(f"{v:+.2f}" for v in ac))


if __name__ == "__main__":
    main()
