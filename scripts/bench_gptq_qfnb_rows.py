#!/usr/bin/env python3
"""csrc/gptq_qfnb.hip: microseconds per column against the rows-per-workgroup choice (one all-gather of G = m / R granules per column;
fewer, fatter workgroups make the gather cheaper and the per-column LDS work longer).  A/B on one box, codes compared with the default."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops  # noqa: E402

dev = "cuda:0"
for (m, d) in [(2048, 2048), (4096, 4096), (8192, 2048), (2048, 8192), (11008, 4096), (1024, 1024), (16384, 2048)]:
    torch.manual_seed(m + d)
    X = torch.randn(d + 256, d, device=dev)
    H = X.T @ X / (d + 256) + 0.01 * torch.eye(d, device=dev)
    W = 0.02 * torch.randn(m, d, device=dev)
    FT = ops.gptq_feedback(H)
    ref, cs_ref = ops.gptq_round_qfnb(W.clone(), FT, 2)
    step = 2.0 * cs_ref[None, :] / 3                                # grid step of every column (the criterion of tests/test_gpu_gptq_qfnb.py)
    row = {"shape": f"{m}x{d}"}
    for R in (0, 1, 64):                                              # 0 default (one XCD up to 4096 rows), 1 pipelined across the XCDs, 64 rounds 3-5
        ops.gptq_qfnb_debug(0, 0, R)
        try:
            try:
                q, _ = ops.gptq_round_qfnb(W.clone(), FT, 2)
            except Exception as ex:
                row[f"R{R}"] = f"{type(ex).__name__}"[:40]
                continue
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                ops.gptq_round_qfnb(W.clone(), FT, 2)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 2
            row[f"R{R}"] = {"ms": round(dt * 1e3, 2), "us_per_column": round(dt * 1e6 / d, 2), "flipped_vs_default": float(((q - ref).abs() > 0.25 * step).float().mean())}
        finally:
            ops.gptq_qfnb_debug(0, 0, 0)
    print(json.dumps(row), flush=True)
