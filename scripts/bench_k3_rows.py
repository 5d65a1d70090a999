#!/usr/bin/env python3
"""Kronecker operator on many rows: row-walking single-launch kernels vs the general two-stage kernel."""
import time, torch, sys, os, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops, method
dev = torch.device("cuda:0")
np.random.seed(0); torch.manual_seed(0)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for n, rows in [(2048, 2048), (8192, 2048), (8192, 8192), (4096, 4096)]:
    op = ops.OrthoOp(method.gen_rand_ortho_butterfly_noblock(n), dev)
    x = torch.randn(rows, n, device=dev)
    y_small = op.apply_rows(x)
    old = op.SMALL_ROWS; small_ok = op.small_ok
    op.small_ok = False
    y_gen = op.apply_rows(x); t_gen = t(lambda: op.apply_rows(x))
    op.small_ok = small_ok
    t_small = t(lambda: op.apply_rows(x))
    xh = x.half()
    t_half = t(lambda: op.apply_rows(xh, out_dtype=torch.bfloat16))
    print(json.dumps({"n": n, "rows": rows, "general_two_stage_ms": round(t_gen, 3), "row_walking_fp32_ms": round(t_small, 3),
                      "row_walking_split_f16_ms": round(t_half, 3), "max_abs_diff": float((y_small - y_gen).abs().max()),
                      "GBs_fp32": round(2 * rows * n * 4 / t_small / 1e6, 1)}), flush=True)
