#!/usr/bin/env python3
"""A/B of the fused attention launch's two forms at 8 / 16 sequences (round 6): 12 waves per workgroup (one per CU: two rounds at 512 workgroups)
against 4 waves (two per CU: one round).  One model, one process, alternating; one JSON line per (form, sequences, repetition)."""
import argparse
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import decode_engine_bench as B  # noqa: E402
from quip_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="opt")
    ap.add_argument("--reps", type=int, default=2)
    a0 = ap.parse_args()
    a = types.SimpleNamespace(arch=a0.arch, layers=0, bits=2, blocked=False, prompt=64, tokens=64, mode="auto", bs=1, blk_fused_n=-1)
    model = B.build(a)
    for rep in range(a0.reps):
        for form in ((0, 0), (0, 257)):                                  # one head per workgroup | three from 257 pairs on
            ops.decode_attention_config(*form)
            for bs in (8, 16):
                a.bs = bs
                r = B.measure(a, *model)
                print(json.dumps({"arch": a0.arch, "attention_forms (one_group_from, three_heads_from)": list(form), "bs": bs, "rep": rep, "tok_per_s": round(r["tok_per_s"], 1),
                                  "ms_per_step": round(r["ms_per_step_median"], 4), "engine_mode": r["engine_mode"]}), flush=True)
    ops.decode_attention_config()


if __name__ == "__main__":
    main()
