#!/usr/bin/env python3
"""Throughput of the headline K2 launch when several INDEPENDENT batches are in flight (fork/join branches of one hipGraph,
one output buffer per branch).  bench.py keeps the serialized number as `value`; this is the secondary figure in DESIGN.md."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from quip_amd import ops, _lib
vp = ctypes.c_void_p
lib = _lib.load()
M = D = 4096; BS = 16
torch.manual_seed(0)
codes = torch.randint(0, 4, (M, D), dtype=torch.uint8).cuda()
qs = ops.pack(codes, 2, ops.LAYOUT_STREAM)
ring = [qs] + [qs.clone() for _ in range(95)]
x = torch.randn(BS, D).to(torch.bfloat16).cuda()
scale = torch.tensor([0.05]).cuda()
def launch(qw, y, st):
    rc = lib.quipamd_dequant_gemm(vp(x.data_ptr()), 2, vp(qw.data_ptr()), 2, 1, 1, vp(scale.data_ptr()), vp(0), vp(0), vp(y.data_ptr()), 2, 0,
                                  ctypes.c_int64(BS), ctypes.c_int64(M), ctypes.c_int64(D), st)
    assert rc == 0
STEPS = 2000
for nstream in (1, 2, 4, 8):
    ys = [torch.empty(BS, M, dtype=torch.bfloat16, device='cuda') for _ in range(nstream)]
    main = torch.cuda.Stream()
    subs = [torch.cuda.Stream() for _ in range(nstream)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        for s_ in range(nstream): launch(ring[s_], ys[s_], vp(main.cuda_stream))
        main.synchronize()
        with torch.cuda.graph(g, stream=main):
            cur = torch.cuda.current_stream()
            for s_, sub in enumerate(subs):
                sub.wait_stream(cur)                       # fork
                with torch.cuda.stream(sub):
                    st = vp(sub.cuda_stream)
                    for i in range(s_, STEPS, nstream):
                        launch(ring[i % 96], ys[s_], st)
            for sub in subs:
                cur.wait_stream(sub)                       # join
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / STEPS)
    print(nstream, "concurrent branches: %.3f us per launch (cold ring), %.0f GB/s" % (best, 4456448 / best / 1e3))
