#!/usr/bin/env python3
"""K2 microbenchmark sweep: us/launch for the fused dequant-GEMM over shapes, batch sizes and forced workgroup
shapes (quipamd_tune_dequant_gemm), cold (weight ring > 256 MiB Infinity Cache) and warm (same weight).
Every timed configuration is first parity-checked against an fp64 dense product of the dequantised weights.
Usage: python scripts/bench_k2.py [--quick] > gpurun_out/k2_sweep.jsonl"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops, _lib  # noqa: E402

vp = ctypes.c_void_p


def make_layer(m, d, bits, dev, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    codes = torch.randint(0, 2 ** bits, (m, d), generator=g, dtype=torch.uint8).to(dev)
    scale = torch.tensor([0.048], device=dev)
    return codes, scale, ops.pack(codes, bits, ops.LAYOUT_STREAM)


def time_graph(launch, weights, steps):
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    nw = len(weights)
    with torch.cuda.stream(side):
        for i in range(3):
            launch(weights[i % nw], vp(side.cuda_stream))
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            st = vp(torch.cuda.current_stream().cuda_stream)
            for i in range(steps):
                launch(weights[i % nw], st)
    graph.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--shapes", default="")
    ap.add_argument("--default-only", action="store_true", help="only the heuristic configuration")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    shapes = [(4096, 4096)] if args.quick else [(2048, 2048), (8192, 2048), (2048, 8192), (4096, 4096), (11008, 4096),
                                                 (4096, 11008), (8192, 8192), (28672, 7168), (7168, 28672)]
    if args.shapes:
        shapes = [tuple(int(v) for v in sh.split("x")) for sh in args.shapes.split(",")]
    for (m, d) in shapes:
        for bits in ([2] if (args.quick or args.default_only) else [2, 4]):
            if d % (512 // bits):
                continue
            codes, scale, qs = make_layer(m, d, bits, dev)
            What = ops.codes_to_weight(codes, "b", scale, None, 2 ** bits - 1, out_dtype=torch.float32).double()
            wbytes = m * d * bits // 8
            nring = max(2, min(96, (400 << 20) // wbytes + 1))
            ring = [qs] + [qs.clone() for _ in range(nring - 1)]
            for bs in ([1, 16] if args.quick else [1, 16, 64, 256]):
                x = torch.randn(bs, d, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).to(dev)
                y = torch.empty(bs, m, dtype=torch.bfloat16, device=dev)
                ref = x.double() @ What.T
                nb = (bs + 15) // 16
                if nb == 1:
                    cfgs = [(0, 0, 0), (1, 0, 16), (2, 0, 16), (4, 0, 16), (1, 0, 8), (2, 0, 8), (4, 0, 8), (1, 1, 16), (2, 1, 16),
                            (1, 0, 8, 200), (2, 0, 16, 200), (4, 0, 16, 200), (2, 0, 8, 200), (4, 0, 8, 200), (1, 0, 4, 200),
                            (8, 0, 16), (8, 0, 16, 200), (8, 0, 8), (8, 0, 8, 200), (4, 1, 16), (4, 1, 8), (2, 1, 8), (4, 1, 4)]
                elif nb == 2:
                    cfgs = [(0, 0, 0), (1, 2, 8), (2, 2, 8), (1, 1, 16), (2, 1, 16), (1, 0, 0, 900), (2, 0, 0, 900), (4, 0, 0, 900)]
                else:
                    cfgs = [(0, 0, 0), (2, 4, 4), (2, 1, 16), (4, 1, 16), (1, 0, 0, 900), (2, 0, 0, 900), (4, 0, 0, 900)]
                if args.default_only:
                    cfgs = [(0, 0, 0)]
                for cfg in cfgs:
                    rt, bt, nw = cfg[:3]
                    sp = cfg[3] if len(cfg) > 3 else 0
                    if rt and (m // 16) % (rt * (2 if sp == 900 else 1)):
                        continue
                    lib.quipamd_tune_dequant_gemm(rt, bt, nw, sp)

                    def launch(qw, st):
                        rc = lib.quipamd_dequant_gemm(vp(x.data_ptr()), 2, vp(qw.data_ptr()), bits, 1, 1,
                                                      vp(scale.data_ptr()), vp(0), vp(0), vp(y.data_ptr()), 2, 0, bs, m, d, st)
                        if rc:
                            raise RuntimeError(lib.quipamd_last_error().decode())
                    try:
                        y.zero_()
                        launch(qs, vp(torch.cuda.current_stream().cuda_stream))
                        torch.cuda.synchronize()
                    except RuntimeError as ex:
                        print(json.dumps({"m": m, "d": d, "bits": bits, "bs": bs, "cfg": list(cfg), "error": str(ex)}), flush=True)
                        continue
                    rel = float((y.double() - ref).norm() / ref.norm())
                    t_cold = time_graph(launch, ring, args.steps)
                    t_warm = time_graph(launch, [qs], args.steps)
                    byts = wbytes + 2 * bs * d + 2 * bs * m
                    print(json.dumps({"m": m, "d": d, "bits": bits, "bs": bs, "cfg": list(cfg), "rel": round(rel, 6),
                                      "us_cold": round(t_cold, 3), "us_warm": round(t_warm, 3),
                                      "GBs_cold": round(byts / t_cold / 1e3, 1), "TF_cold": round(2 * bs * m * d / t_cold / 1e6, 2),
                                      "TF_warm": round(2 * bs * m * d / t_warm / 1e6, 2)}), flush=True)
            del ring
            torch.cuda.empty_cache()
    lib.quipamd_tune_dequant_gemm(0, 0, 0, 0)


if __name__ == "__main__":
    main()
