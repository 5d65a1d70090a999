#!/usr/bin/env python3
"""quipamd_decode_bigp_v_gemm (Llama's down_proj in a decode step: V_down over 688 x 16 + split-K 2-bit GEMM 4096 x 11008) by row count,
with the K-slices meeting through fp32 atomics and through the fixed-order scratch (quant.DETERMINISTIC_SPLITK): us per launch in a
hipGraph of 50 launches.   python scripts/bench_bigp_tail.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from quip_amd import ops  # noqa: E402

DEV = "cuda:0"


def main():
    import test_gpu_decode_bigp as T
    ffn, h = 11008, 4096
    down, _ = T._layer(ffn, h, 7, bias=False, bits=2)
    V = down.V
    for rows in (1, 2, 4, 5, 8, 16):
        g = torch.randn(rows, ffn, device=DEV).half()
        u = torch.randn(rows, ffn, device=DEV).half()
        y = torch.zeros(rows, h, device=DEV)
        partials = torch.empty(V.p // 16, rows, h, device=DEV)
        res = {"rows": rows}
        for name, kw in (("atomics", {}), ("fixed_order", dict(partials=partials)), ("two_launch", dict(xt=torch.empty(rows, ffn, dtype=torch.float16, device=DEV)))):
            for nrt in (0, 2, 1):
                def one():
                    ops.decode_bigp_v_gemm(V, g, u, down.decode_qweight(), down.scales, y, nrt, bits=2, **kw)
                one()
                torch.cuda.synchronize()
                side = torch.cuda.Stream()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(graph, stream=side):
                        for _ in range(50):
                            one()
                graph.replay()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    graph.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / 50)
                res[f"{name}_nrt{nrt}_us"] = round(best, 2)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
