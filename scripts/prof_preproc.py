#!/usr/bin/env python3
"""Where QuantMethod.preproc + Balance.fasterquant spend their time at the OPT-1.3B shapes (one GPU): operator
sampling (host RNG + Householder accumulation), projection of W and H (K3), Cholesky, LDLQ (K4), postproc."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import bal, method, ops, quant  # noqa: E402


def sync_time(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def main():
    import scipy.stats
    dev = torch.device("cuda:0")
    np.random.seed(0)
    torch.manual_seed(0)
    for (m, d) in [(2048, 2048), (8192, 2048), (2048, 8192)]:
        out = {"shape": f"{m}x{d}"}
        for n in sorted({m, d}):
            p, q = method.butterfly_factors(n)
            t, _ = sync_time(lambda: method.gen_rand_ortho_butterfly(n), reps=2)
            t0 = time.perf_counter()
            scipy.stats.special_ortho_group.rvs(p, size=n // p)
            scipy.stats.special_ortho_group.rvs(q, size=n // q)
            out[f"gen_n{n}_ms"] = round(t * 1e3, 2)
            out[f"scipy_n{n}_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        X = torch.randn(d + 256, d, device=dev)
        H = (X.T @ X / (d + 256)).double()
        layer = torch.nn.Linear(d, m, bias=False).to(dev).half()
        W0 = (0.02 * torch.randn(m, d)).to(dev).half()

        def run():
            layer.weight.data = W0.clone()
            b = bal.Balance(layer)
            b.configure('ldlq', 2, 0, False)
            b.quantizer = quant.Quantizer()
            b.quantizer.configure(2, perchannel=True, sym=False, qfn='b', mse=False)
            b.H = H.clone()
            b.nsamples = 1
            b.post_batch()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b.preproc(preproc_gptqH=True, percdamp=0.01, preproc_rescale=True, preproc_proj=True, preproc_proj_extra=0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            b.fasterquant(lazy_batch=False)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            b.free()
            return t1 - t0, t2 - t1
        run()
        ts = [run() for _ in range(3)]
        out["preproc_ms"] = round(min(t[0] for t in ts) * 1e3, 2)
        out["fasterquant_ms"] = round(min(t[1] for t in ts) * 1e3, 2)
        Hf = H.float() + 0.01 * H.float().diag().mean() * torch.eye(d, device=dev)
        t, C = sync_time(lambda: torch.linalg.cholesky(Hf))
        out["cholesky_ms"] = round(t * 1e3, 2)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
