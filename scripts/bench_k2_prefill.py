#!/usr/bin/env python3
"""K2 at prefill-sized batches: the round-2 MB kernel (family 4; the round-3 prefill kernel it beat is scripts/dqgemm_pf_lab.hip, no longer in the library) and the dense
fp16 / bf16 rocBLAS GEMM of the same shape, cold-ish operands (weights cycled over several copies), HIP events around graph-captured
launches.  Fractions against the 2.5 PFLOP/s dense bf16 MFMA peak (MI355X_MICROARCH.md)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops  # noqa: E402

PEAK = 2500.0


def time_graph(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (m, d, bs) in [(4096, 4096, 2048), (4096, 4096, 256), (28672, 7168, 256), (7168, 28672, 256), (8192, 8192, 1024), (11008, 4096, 2048), (4096, 4096, 8192)]:
        ncopy = max(1, min(8, (256 << 20) // (m * d // 4)))
        qs = [ops.pack(torch.randint(0, 4, (m, d), dtype=torch.uint8, device=dev), 2, ops.LAYOUT_STREAM) for _ in range(ncopy)]
        sc = torch.tensor([0.05], device=dev)
        row = {"m": m, "d": d, "bs": bs}
        for dt in (torch.bfloat16,):
            x = torch.randn(bs, d, device=dev).to(dt)
            y = torch.empty(bs, m, dtype=dt, device=dev)
            for name, cfg in (("mb", (4, 44)), ("auto", None)):
                it = [0]

                def f():
                    ops.dequant_gemm(x, qs[it[0] % ncopy], 2, 'b', sc, None, None, out=y, m=m, cfg=cfg)
                    it[0] += 1
                try:
                    t = time_graph(f)
                    tf = 2.0 * bs * m * d / t / 1e12
                    row[name] = {"us": round(t * 1e6, 1), "TFLOPs": round(tf, 1), "mfma_frac": round(tf / PEAK, 3)}
                except Exception as ex:
                    row[name] = {"error": str(ex)[:80]}
            Wd = torch.randn(m, d, device=dev).to(dt)
            t = time_graph(lambda: torch.nn.functional.linear(x, Wd))
            row["dense_rocblas_" + str(dt).split(".")[-1]] = {"us": round(t * 1e6, 1), "TFLOPs": round(2.0 * bs * m * d / t / 1e12, 1)}
            del Wd
        print(json.dumps(row), flush=True)
        del qs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
