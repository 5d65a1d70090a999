#!/usr/bin/env python3
"""Packed incoherence-processed Linear at prefill batch sizes (rows = tokens) next to dense fp16 F.linear: the V-side and
U-side Kronecker operators run on the row-walking single-launch K3 kernels, the product on K2's batched kernels."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import method, ops  # noqa: E402
from quip_amd.quant import QuantLinear  # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    dev = torch.device("cuda:0")
    np.random.seed(0)
    torch.manual_seed(0)
    for (m, d) in [(2048, 2048), (8192, 2048), (2048, 8192)]:
        W = (0.02 * torch.randn(m, d)).half().to(dev)
        s = ops.qfnb_scale(W)
        _, codes = ops.quantize(W, 'b', s, None, 3, want_codes=True)
        ql = QuantLinear(d, m, bits=2, qfn='b').to(dev)
        ql.pack(codes, s, None, bias=torch.randn(m).to(dev), scaleWH=(0.5 + torch.rand(d)).to(dev),
                U=method.gen_rand_ortho_butterfly_noblock(m), V=method.gen_rand_ortho_butterfly_noblock(d))
        lin = torch.nn.Linear(d, m).half().to(dev)
        for rows in (128, 2048):
            x = torch.randn(rows, d, device=dev).half()
            with torch.no_grad():
                t_p = timeit(lambda: ql(x))
                t_d = timeit(lambda: lin(x))
                xt = ql.V.apply_rows(x, colscale=ql.inv_scaleWH, out_dtype=torch.bfloat16)
                t_v = timeit(lambda: ql.V.apply_rows(x, colscale=ql.inv_scaleWH, out_dtype=torch.bfloat16))
                y = ops.dequant_gemm(xt, ql.qweight, 2, 'b', ql.scales, None, None, out_dtype=torch.float32, m=m)
                t_g = timeit(lambda: ops.dequant_gemm(xt, ql.qweight, 2, 'b', ql.scales, None, None, out_dtype=torch.float32, m=m))
                t_u = timeit(lambda: ql.U.apply_rows(y, transpose=True, out_dtype=torch.float16, bias=ql.bias))
            print(json.dumps({"m": m, "d": d, "rows": rows, "packed_forward_us": round(t_p * 1e6, 1), "dense_fp16_us": round(t_d * 1e6, 1),
                              "V_side_us": round(t_v * 1e6, 1), "K2_us": round(t_g * 1e6, 1), "U_side_us": round(t_u * 1e6, 1),
                              "K2_TFLOPs": round(2.0 * rows * m * d / t_g / 1e12, 1)}), flush=True)


if __name__ == "__main__":
    main()
