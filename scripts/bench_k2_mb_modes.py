#!/usr/bin/env python3
"""How the bs > 16 legs of bench.py's `k2_shapes` depend on the measurement: K launches timed after W warm-up launches, eager behind a
spin kernel (bench.py's form for K <= 256) or as one hipGraph, cold weights (ring of copies) -- the same kernel, the same box.
Usage: bench_k2_mb_modes.py m d bs"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import _lib, ops  # noqa: E402

m, d, bs = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
lib = _lib.load()
fn, vp = lib.quipamd_dequant_gemm, ctypes.c_void_p
g = torch.Generator().manual_seed(1)
codes = torch.randint(0, 4, (m, d), generator=g, dtype=torch.uint8).to(dev)
q = ops.pack(codes, 2, ops.LAYOUT_STREAM)
del codes
wb = m * d // 4
nr = max(2, min(96, (400 << 20) // wb + 1))
ring = [q] + [q.clone() for _ in range(nr - 1)]
x = torch.randn(bs, d, generator=g).to(torch.bfloat16).to(dev)
y = torch.empty(bs, m, dtype=torch.bfloat16, device=dev)
sc = torch.tensor([0.05], device=dev)
side = torch.cuda.Stream()


def launch(qw, st):
    rc = fn(vp(x.data_ptr()), 2, vp(qw.data_ptr()), 2, 1, 1, vp(sc.data_ptr()), vp(0), vp(0), vp(y.data_ptr()), 2, 0, bs, m, d, st)
    assert rc == 0, lib.quipamd_last_error()


def eager(steps, warmup, spin=True):
    with torch.cuda.stream(side):
        st = vp(side.cuda_stream)
        for i in range(warmup):
            launch(ring[(nr - 1 - i) % nr], st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if spin:
            torch.cuda._sleep(int(min(steps, 400) * 30000 + 300000))
        for i in range(min(warmup, 8)):
            launch(ring[(nr - 1 - i) % nr], st)
        e0.record(side)
        for i in range(steps):
            launch(ring[i % nr], st)
        e1.record(side)
        side.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


def graph(steps, warmup):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        st = vp(side.cuda_stream)
        launch(ring[0], st)
        side.synchronize()
        with torch.cuda.graph(gr, stream=side):
            cst = vp(torch.cuda.current_stream().cuda_stream)
            for i in range(steps):
                launch(ring[i % nr], cst)
        for i in range(warmup):
            launch(ring[(nr - 1 - i) % nr], st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        gr.replay()
        e1.record(side)
        side.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


flops = 2.0 * bs * m * d
for name, f in (("eager K=50 W=20 (bench.py)", lambda: eager(50, 20)), ("eager K=50 W=20 no spin", lambda: eager(50, 20, False)),
                ("eager K=50 W=300", lambda: eager(50, 300)), ("eager K=250 W=300", lambda: eager(250, 300)),
                ("graph K=300 W=0", lambda: graph(300, 0)), ("graph K=300 W=300", lambda: graph(300, 300)),
                ("eager K=50 W=20 again", lambda: eager(50, 20))):
    us = [f() for _ in range(3)]
    print(json.dumps({"shape": [m, d, bs], "mode": name, "us_per_launch": [round(u, 2) for u in us], "TFLOPs_best": round(flops / min(us) / 1e6, 1)}), flush=True)
