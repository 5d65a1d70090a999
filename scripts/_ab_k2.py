import ctypes, sys, torch
sys.path.insert(0, '.')
from quip_amd import ops, _lib
vp = ctypes.c_void_p
cur, old = _lib.load(), ctypes.CDLL("build/libk2_r1c.so")
M = D = 4096; BS = 16
torch.manual_seed(0)
codes = torch.randint(0, 4, (M, D), dtype=torch.uint8).cuda()
qs = ops.pack(codes, 2, ops.LAYOUT_STREAM)
ring = [qs] + [qs.clone() for _ in range(95)]
x = torch.randn(BS, D).to(torch.bfloat16).cuda(); y = torch.empty(BS, M, dtype=torch.bfloat16, device='cuda')
yf = torch.zeros(BS, M, device='cuda')
scale = torch.tensor([0.05]).cuda()
def mk(lib, acc):
    def launch(qw, st):
        rc = lib.quipamd_dequant_gemm(vp(x.data_ptr()), 2, vp(qw.data_ptr()), 2, 1, 1, vp(scale.data_ptr()), vp(0), vp(0),
                                      vp((yf if acc else y).data_ptr()), 0 if acc else 2, 1 if acc else 0,
                                      ctypes.c_int64(BS), ctypes.c_int64(M), ctypes.c_int64(D), st)
        assert rc == 0
    return launch
def graph_of(launch, weights, steps=1000):
    side = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for i in range(3): launch(weights[i % len(weights)], vp(side.cuda_stream))
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            st = vp(torch.cuda.current_stream().cuda_stream)
            for i in range(steps): launch(weights[i % len(weights)], st)
    return g
variants = {}
for name, lib in (("cur", cur), ("r1c", old)):
    for acc in (False, True):
        for cold in (True, False):
            variants[(name, "acc" if acc else "bf16", "cold" if cold else "warm")] = graph_of(mk(lib, acc), ring if cold else [qs])
for rnd in range(4):
    for k, g in variants.items():
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        if rnd: print(rnd, k, "%.3f us" % (e0.elapsed_time(e1)))
