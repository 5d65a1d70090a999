#!/usr/bin/env python3
"""The weight-stream family of K2 (dq_s_kernel, bs <= 16) under its forced configurations, cold weights (a ring of packed copies larger than
the 256 MiB Infinity Cache), one hipGraph of `steps` launches per configuration: us per launch, GB/s of packed bytes, fraction of 8 TB/s.
Every configuration is checked against the default one first.
    python scripts/bench_k2_s_cfgs.py [--shapes 28672x7168,32768x8192] [--steps 60]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops  # noqa: E402

FAM_S = 3
CFGS = [None, (FAM_S, 7, 2), (FAM_S, 8, 1), (FAM_S, 4, 2), (FAM_S, 2, 4), (FAM_S, 1, 8)]     # (round 4 also timed 24 / 22; round 6 ring depth 4 / 5 and two stages per step: no gain, not in the library)     # (round 4 also timed 24 / 22: two row tiles per compute wave -- no gain, not in the library)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="28672x7168,32768x8192,16384x8192,8192x8192")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for sh in a.shapes.split(","):
        m, d = (int(v) for v in sh.split("x"))
        g = torch.Generator().manual_seed(m + d)
        codes = torch.randint(0, 4, (m, d), generator=g, dtype=torch.uint8).to(dev)
        q = ops.pack(codes, 2, ops.LAYOUT_STREAM)
        del codes
        wbytes = m * d // 4
        ring = [q] + [q.clone() for _ in range(max(2, min(64, (420 << 20) // wbytes + 1)) - 1)]
        adt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
        x = torch.randn(a.bs, d, generator=g).to(adt).to(dev)
        sc = torch.tensor([0.05], device=dev)
        y = torch.empty(a.bs, m, dtype=adt, device=dev)
        ref = ops.dequant_gemm(x, q, 2, "b", sc, None, None, out_dtype=torch.float32).clone()
        for cfg in CFGS:
            try:
                got = ops.dequant_gemm(x, q, 2, "b", sc, None, None, out_dtype=torch.float32, cfg=cfg)
                err = float((got - ref).abs().max() / ref.abs().max())
                side = torch.cuda.Stream()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    for i in range(3):
                        ops.dequant_gemm(x, ring[i % len(ring)], 2, "b", sc, None, None, out=y, cfg=cfg)
                    side.synchronize()
                    with torch.cuda.graph(graph, stream=side):
                        for i in range(a.steps):
                            ops.dequant_gemm(x, ring[(i + 3) % len(ring)], 2, "b", sc, None, None, out=y, cfg=cfg)
                graph.replay()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    graph.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / a.steps)
                byt = wbytes + a.bs * d * 2 + a.bs * m * 2
                print(json.dumps({"m": m, "d": d, "bs": a.bs, "cfg": cfg, "us": round(best, 3), "GBs": round(byt / best / 1e3, 1),
                                  "hbm_frac": round(byt / best / 1e3 / 8000, 4), "max_rel_diff_vs_default": err}), flush=True)
            except Exception as e:
                print(json.dumps({"m": m, "d": d, "cfg": cfg, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}), flush=True)
        del ring, q
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
