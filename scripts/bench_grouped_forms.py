#!/usr/bin/env python3
"""A/B of the kernel behind quipamd_dequant_gemm_grouped at d = 4096 inside the real decode engine (round 6): Llama-2-7B at 8 / 16 sequences
per step with the grouped h kernel (4 row tiles per workgroup: round 5's default) against the grouped weight-stream kernel dq_sg_kernel in its
three workgroup forms.  One model, one process, alternating; one JSON line per (form, bs)."""
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
    a = types.SimpleNamespace(arch="llama", layers=0, bits=2, blocked=False, prompt=32, tokens=32, mode="auto", bs=16, blk_fused_n=-1)
    model = B.build(a)
    for rep in range(2):
        for form in (4, 74, 72, 81):
            for bs in (16, 8):
                a.bs = bs
                ops.dequant_gemm_grouped_config(form)
                try:
                    r = B.measure(a, *model)
                finally:
                    ops.dequant_gemm_grouped_config(0)
                print(json.dumps({"form": form, "bs": bs, "rep": rep, "ms_per_step": round(r["ms_per_step_median"], 4), "tok_per_s": round(r["tok_per_s"], 1),
                                  "engine_mode": r["engine_mode"]}), flush=True)


if __name__ == "__main__":
    main()
