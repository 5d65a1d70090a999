#!/usr/bin/env python3
"""A/B of decode-engine switches on ONE model in ONE process, alternating (round 6): quip_amd.decode.OPERAND_PREFETCH off / on.
One JSON line per (switch, repetition): tok/s at batch 1 through the graph-replayed engine (scripts/decode_engine_bench.py's measure)."""
import argparse
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import decode_engine_bench as B  # noqa: E402
from quip_amd import decode  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="opt")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--blocked", action="store_true")
    a0 = ap.parse_args()
    a = types.SimpleNamespace(arch=a0.arch, layers=0, bits=2, blocked=a0.blocked, prompt=64, tokens=64, mode="auto", bs=a0.bs, blk_fused_n=-1)
    model = B.build(a)
    for rep in range(a0.reps):
        for pf in ((False,) if a0.blocked else (False, True)):
            decode.OPERAND_PREFETCH = pf
            r = B.measure(a, *model)
            print(json.dumps({"arch": a0.arch, "blocked": a0.blocked, "bs": a0.bs, "operand_prefetch": pf, "rep": rep, "tok_per_s": round(r["tok_per_s"], 1),
                              "ms_per_step": round(r["ms_per_step_median"], 4), "engine_mode": r["engine_mode"]}), flush=True)


if __name__ == "__main__":
    main()
