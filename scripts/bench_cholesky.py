#!/usr/bin/env python3
"""K8 (ops.cholesky_lt) next to torch.linalg.cholesky + unit_lower_t at d = 2048 ... 11008 (numbers in DESIGN.md, K8)."""
import time, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops
dev = "cuda:0"
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for d in (2048, 4096, 8192, 11008, 16384):
    X = torch.randn(d + 256, d, device=dev); H = X.T @ X / d + 0.01 * torch.eye(d, device=dev)
    print(d, "K8 cholesky_lt %.3f ms (no check %.3f)" % (t(lambda: ops.cholesky_lt(H)), t(lambda: ops.cholesky_lt(H, check=False))),
          "torch cholesky+unit_lower_t %.3f ms" % t(lambda: ops.unit_lower_t(torch.linalg.cholesky(H))), flush=True)
