#!/usr/bin/env python3
"""What is between two launches of a decode step?  (round 6)

In the rocprofv3 timeline of the graph-replayed engine every launch's "duration" is its PERIOD (begin / end stamps tile the stream), and the
in-situ stamps of scripts/decode_stamps.py see one workgroup.  Here the probe build's QA_LOG makes thread 0 of EVERY workgroup of the
instrumented kernels (blk_stage_kernel, dq_h / dq_hg_kernel, decode_attn_kernel) append {s_memrealtime, start | end, workgroup} to a device
log while the engine's hipGraph replays -- no host in the loop.  Per launch: workgroups, dispatch skew (first -> last start), the median
workgroup's own span, the tail (first -> last end), and the TRUE gap = first start of the next launch - last end of this one.

    python scripts/decode_wglog.py [--blocked] [--arch opt]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "quip_amd", "csrc", "libquip_amd_probe.so")
os.environ["QUIP_AMD_LIB"] = PROBE
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="opt", choices=["opt", "llama"])
    ap.add_argument("--blocked", action="store_true")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--bits", type=int, default=2)
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--tokens", type=int, default=2, help="tokens logged (the table holds 1024 launches)")
    a = ap.parse_args()
    from quip_amd import _lib, decode
    import decode_engine_bench as B
    model, nbytes, arch = B.build(a)
    dev = torch.device("cuda:0")
    eng = decode.DecodeEngine(model, bs=a.bs, max_len=160, mode="auto")
    ids = torch.randint(0, 30000, (a.bs, 128), device=dev)
    for i in range(96):                                                 # fill the cache, capture and warm the graph
        eng.forward(ids[:, i])
    torch.cuda.synchronize()
    buf = torch.zeros(258 + 1024 * 1024 * 2, dtype=torch.int64, device=dev)
    _lib.call("quipamd_probe_set", buf.data_ptr())
    buf[257] = 1
    torch.cuda.synchronize()
    for i in range(a.tokens):
        eng.forward(ids[:, 96 + i])
    torch.cuda.synchronize()
    nl = int(buf[256].item())
    tab = buf[258:].view(1024, 1024, 2).cpu().numpy().astype(np.int64) * 10        # ns
    _lib.call("quipamd_probe_set", None)
    launches = []
    for k in range(max(0, nl - 1000), nl):
        rec = tab[k % 1024]
        live = rec[:, 0] > 0
        if live.any():
            launches.append({"s": rec[live, 0], "e": rec[live, 1]})
    print(f"# {arch}; engine mode {eng.mode}; {nl} instrumented launches in {a.tokens} tokens (kernels without QA_LOG -- the fused Kronecker "
          f"launches, the head -- show up as long gaps)")
    print(f"{'#':>4} {'workgroups':>10} {'start skew':>11} {'median wg span':>15} {'end tail':>9} {'first start -> last end':>24} {'gap to next launch':>19}   (us)")
    per = None
    rows = []
    for i, l in enumerate(launches):
        s, e = np.array(l["s"]), np.array(l["e"])
        nxt = min(launches[i + 1]["s"]) if i + 1 < len(launches) else None
        rows.append((len(s), (s.max() - s.min()) / 1e3, (np.median(e) - np.median(s)) / 1e3, (e.max() - e.min()) / 1e3, (e.max() - s.min()) / 1e3,
                     (nxt - e.max()) / 1e3 if nxt is not None else float("nan")))
    # one token in the middle: print its launches, then medians per distinct workgroup count
    lo = len(rows) // a.tokens if a.tokens > 1 else 0
    for i in range(lo, min(lo + 40, len(rows))):
        r = rows[i]
        print(f"{i:4d} {r[0]:10d} {r[1]:11.2f} {r[2]:15.2f} {r[3]:9.2f} {r[4]:24.2f} {r[5]:19.2f}")
    arr = np.array(rows[lo:-1])
    print("# medians over the launches from the second token on, gaps < 3 us only (instrumented kernel follows instrumented kernel):")
    adj = arr[arr[:, 5] < 3.0]
    if len(adj):
        print(f"#   start skew {np.median(arr[:, 1]):.2f}  median workgroup span {np.median(arr[:, 2]):.2f}  end tail {np.median(arr[:, 3]):.2f}  "
              f"launch body {np.median(arr[:, 4]):.2f}  true gap {np.median(adj[:, 5]):.2f} (p10 {np.percentile(adj[:, 5], 10):.2f}, p90 {np.percentile(adj[:, 5], 90):.2f}; "
              f"{len(adj)} adjacent pairs)")


if __name__ == "__main__":
    main()
