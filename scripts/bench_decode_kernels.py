#!/usr/bin/env python3
"""GPU time per launch (hipGraph of 200 dependent launches) of the kernels a packed decode step is made of, batch 1:
small-batch K3 (V side with LayerNorm, U side with bias+residual), K2 GEMV-shaped launches, and rocBLAS fp16 GEMV."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops, method
from quip_amd.quant import QuantLinear

dev = torch.device("cuda:0")


def graph_time(fn, n=200):
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


np.random.seed(0); torch.manual_seed(0)
rows = 1
out = {}
for n in (2048, 8192):
    op = ops.OrthoOp(method.gen_rand_ortho_butterfly_noblock(n), dev)
    x = torch.randn(rows, n, device=dev).half(); xo = torch.empty(rows, n, device=dev, dtype=torch.bfloat16)
    ln = torch.nn.LayerNorm(n).half().to(dev)
    cs = (0.5 + torch.rand(n)).to(dev)
    out[f"K3 small V-side n={n} (LN + colscale, fp16->bf16)"] = graph_time(
        lambda: ops.ortho_small_ops([op.small_op(x, xo, colscale=cs, ln=(ln.weight, ln.bias, ln.eps))], rows))
    y = torch.randn(rows, n, device=dev); yo = torch.empty(rows, n, device=dev, dtype=torch.float16)
    b = torch.randn(n, device=dev); res = torch.randn(rows, n, device=dev).half()
    out[f"K3 small U-side n={n} (transpose, bias + residual, fp32->fp16)"] = graph_time(
        lambda: ops.ortho_small_ops([op.small_op(y, yo, transpose=True, bias=b, residual=res)], rows))
for (m, d) in [(2048, 2048), (8192, 2048), (2048, 8192)]:
    codes = torch.randint(0, 4, (m, d), dtype=torch.uint8, device=dev)
    qs = ops.pack(codes, 2, ops.LAYOUT_STREAM)
    sc = torch.tensor([0.05], device=dev)
    xb = torch.randn(rows, d, device=dev).to(torch.bfloat16)
    yb = torch.empty(rows, m, device=dev)
    out[f"K2 {m}x{d} bs=1 (fp32 y)"] = graph_time(lambda: ops.dequant_gemm(xb, qs, 2, 'b', sc, None, None, out=yb))
    W = torch.randn(m, d, device=dev).half(); xh = torch.randn(rows, d, device=dev).half()
    out[f"rocBLAS fp16 F.linear {m}x{d} bs=1"] = graph_time(lambda: torch.nn.functional.linear(xh, W))
xh = torch.randn(rows, 2048, device=dev).half(); ln = torch.nn.LayerNorm(2048).half().to(dev)
out["torch LayerNorm 2048 fp16"] = graph_time(lambda: ln(xh))
out["torch add 2048 fp16"] = graph_time(lambda: xh + xh)
print(json.dumps({k: round(v, 2) for k, v in out.items()}, indent=1))
