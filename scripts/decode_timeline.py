#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of the graph-replayed decode engine -> what every launch of a block costs the token: its traced duration and
the GAP to the start of the next launch (the dependent kernel boundary in situ), per position in the block's launch sequence.

The trace holds every dispatch of the process (model set-up, warm-up, the timed tokens).  The timed tokens are the tail of the trace: the
last `--tokens` graph replays are cut into steps at the head launch, and within a step the block's launch sequence is its shortest period.
Medians over (tokens x blocks) per position.

    python scripts/decode_timeline.py <trace_results.db> [--tokens 48] [--head head_kernel]"""
import argparse
import sqlite3

import numpy as np


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:n.index("(")] if "(" in n else n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--tokens", type=int, default=48)
    ap.add_argument("--head", default="", help="substring of the kernel that ends a step (default: the longest-running frequent kernel)")
    ap.add_argument("--offset", type=int, default=-1, help="launches of a step in front of the first block (default: half of what the blocks leave over)")
    ap.add_argument("--dump", default="", help="write (kernel name ids, start, end) of the analysed steps to this .npz for later re-analysis")
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    names = [short(r[0]) for r in rows]
    st = np.array([r[1] for r in rows], dtype=np.int64)
    en = np.array([r[2] for r in rows], dtype=np.int64)
    head = a.head
    if not head:
        cand = {}
        for n, s, e in zip(names, st, en):
            cand.setdefault(n, []).append(e - s)
        freq = {n: (len(v), float(np.median(v))) for n, v in cand.items() if len(v) >= a.tokens}
        head = max(freq, key=lambda n: freq[n][1])
    ends = [i for i, n in enumerate(names) if head in n]
    ends = ends[-a.tokens:]
    steps = [(ends[i - 1] + 1, ends[i] + 1) for i in range(1, len(ends))]
    lens = [b - c for c, b in steps]
    L = int(np.median(lens))
    steps = [(c, b) for c, b in steps if b - c == L]
    seq = names[steps[-1][0]:steps[-1][1]]
    per, off, run = 0, 0, 0
    for p_ in range(2, 80):                                             # the block's launch sequence: the shortest period that holds over most of the step
        best, cur, start, bstart = 0, 0, 0, 0
        for i in range(L - p_):
            if seq[i] == seq[i + p_]:
                if cur == 0:
                    start = i
                cur += 1
                if cur > best:
                    best, bstart = cur, start
            else:
                cur = 0
        if best >= 0.6 * L:
            per, off, run = p_, bstart, best + p_
            break
    if not per:
        raise SystemExit(f"no periodic block structure found in a step of {L} launches: " + ", ".join(seq[:40]))
    if a.offset >= 0:
        off = a.offset
    nblk = run // per if a.offset < 0 else (L - off) // per
    dur = np.zeros((len(steps), nblk - 1, per))
    gap = np.zeros_like(dur)
    for si, (c, _) in enumerate(steps):
        for b in range(nblk - 1):
            for k in range(per):
                i = c + off + b * per + k
                dur[si, b, k] = en[i] - st[i]
                gap[si, b, k] = st[i + 1] - en[i]
    if a.dump:
        uniq = sorted(set(names))
        ix = {n: i for i, n in enumerate(uniq)}
        lo_i, hi_i = steps[0][0], steps[-1][1]
        np.savez_compressed(a.dump, names=np.array(uniq), ids=np.array([ix[n] for n in names[lo_i:hi_i]], dtype=np.int32), start=st[lo_i:hi_i], end=en[lo_i:hi_i],
                            steps=np.array(steps) - lo_i)
    tok = np.array([st[b - 1] + (en[b - 1] - st[b - 1]) - st[c] for c, b in steps])
    print(f"# {a.db}: {len(steps)} steps of {L} launches (step = up to '{head}'), {per} launches per block x {nblk} blocks; "
          f"median step {np.median(tok) / 1e3:.1f} us from the first launch's start to the head's end")
    print(f"{'#':>2} {'kernel':<70} {'duration us':>12} {'gap to next us':>15} {'period us':>10}")
    tot = 0.0
    for k in range(per):
        d, g = np.median(dur[:, :, k]) / 1e3, np.median(gap[:, :, k]) / 1e3
        tot += d + g
        print(f"{k:2d} {seq[off + k][:70]:<70} {d:12.2f} {g:15.2f} {d + g:10.2f}")
    print(f"   {'one block':<70} {'':>12} {'':>15} {tot:10.2f}")
    rest = names[steps[-1][0]:steps[-1][0] + off] + names[steps[-1][0] + off + nblk * per:steps[-1][1]]
    print("# outside the blocks: " + ", ".join(rest))


if __name__ == "__main__":
    main()
