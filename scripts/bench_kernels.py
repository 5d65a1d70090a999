#!/usr/bin/env python3
"""Per-kernel timings of the quantisation-side hot path (K1 pack, K5 grid map, K3 structured projection, K4 LDLQ)
and of the whole Balance path (preproc + fasterquant) on one Linear, against each kernel's algorithmic bytes / flops
(SURVEY.md 8(d)).  One JSON object per line.  Usage: python scripts/bench_kernels.py [--shapes 4096x4096,...]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops  # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return float(np.median(ts))


def emit(**kw):
    print(json.dumps(kw), flush=True)


def correlated_H(d, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(d + 256, d, generator=g).to(dev)
    return (X.T @ X) / (d + 256)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096x4096,2048x2048,8192x2048,2048x8192")
    ap.add_argument("--bits", type=int, default=2)
    ap.add_argument("--no-balance", action="store_true")
    ap.add_argument("--only-k7", action="store_true", help="Hessian accumulation only")
    ap.add_argument("--tokens", type=int, default=2048, help="tokens per add_batch call (opt.py: seqlen 2048)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    bits, maxq = args.bits, 2 ** args.bits - 1
    for shp in args.shapes.split(","):
        m, d = (int(v) for v in shp.split("x"))
        torch.manual_seed(0)
        # ---- K7: one add_batch call (tokens x d fp16 activations) next to the reference's op, a dense fp64 GEMM
        xh = torch.randn(args.tokens, d, device=dev).half()
        Hacc = torch.zeros(d, d, dtype=torch.float64, device=dev)
        t = timeit(lambda: ops.hessian_accum(Hacc, xh), reps=5, warm=1)

        def ref_add():
            x64 = xh.t().to(torch.float64)
            Hacc.addmm_(x64, x64.t())
        t_ref = timeit(ref_add, reps=3, warm=1)
        t_fast = timeit(lambda: ops.hessian_accum(Hacc, xh, fast=True), reps=5, warm=1)
        t_fin = timeit(lambda: ops.hessian_finish(Hacc, 1.0), reps=3, warm=1)
        emit(kernel="K7 hessian_accum f16", tokens=args.tokens, d=d, ms=t * 1e3,
             dense_equiv_fp64_TFLOPs=2 * args.tokens * d * d / t / 1e12, torch_fp64_addmm_ms=t_ref * 1e3,
             speedup=t_ref / t, finish_ms=t_fin * 1e3, fast_mode_ms=t_fast * 1e3)
        del Hacc, xh
        if args.only_k7:
            continue
        W = (0.02 * torch.randn(m, d)).to(dev)
        W16 = W.half()
        # ---- K5
        t = timeit(lambda: ops.qfnb_scale(W16))
        emit(kernel="K5 qfnb_scale f16", m=m, d=d, ms=t * 1e3, GBs=m * d * 2 / t / 1e9)
        s = ops.qfnb_scale(W16)
        t = timeit(lambda: ops.gridmap(W16, 'b', s, None, maxq))
        emit(kernel="K5 gridmap f16->f32", m=m, d=d, ms=t * 1e3, GBs=m * d * 6 / t / 1e9)
        t = timeit(lambda: ops.quantize(W16, 'b', s, None, maxq, want_codes=True))
        emit(kernel="K5 quantize f16 (+codes)", m=m, d=d, ms=t * 1e3, GBs=m * d * 5 / t / 1e9)
        _, codes = ops.quantize(W16, 'b', s, None, maxq, want_codes=True)
        t = timeit(lambda: ops.codes_to_weight(codes, 'b', s, None, maxq))
        emit(kernel="K5 codes_to_weight ->f16", m=m, d=d, ms=t * 1e3, GBs=m * d * 3 / t / 1e9)
        # ---- K1
        for lay, name in [(ops.LAYOUT_CANONICAL, "canonical"), (ops.LAYOUT_STREAM, "stream")]:
            if lay == ops.LAYOUT_STREAM and d % (512 // bits):
                continue
            t = timeit(lambda: ops.pack(codes, bits, lay))
            emit(kernel=f"K1 pack {name}", m=m, d=d, ms=t * 1e3, GBs=m * d * (1 + bits / 8) / t / 1e9)
            pk = ops.pack(codes, bits, lay)
            t = timeit(lambda: ops.unpack(pk, bits, lay, m, d))
            emit(kernel=f"K1 unpack {name}", m=m, d=d, ms=t * 1e3, GBs=m * d * (1 + bits / 8) / t / 1e9)
        # ---- K3 (factors from the reference-style generator; scipy sampling is host-side setup, not timed)
        from quip_amd import method
        np.random.seed(0)
        torch.manual_seed(0)
        for gname, gen in [("blocked", method.gen_rand_ortho_butterfly), ("kron", method.gen_rand_ortho_butterfly_noblock)]:
            Bpp = gen(d)
            op = ops.OrthoOp(Bpp, dev)
            p, q = op.p, op.q
            for dt, nm in [(torch.float32, "f32"), (torch.bfloat16, "bf16")]:
                X = W.to(dt)
                t = timeit(lambda: op.apply_rows(X))
                emit(kernel=f"K3 ortho rows {gname} {nm}", m=m, d=d, p=p, q=q, ms=t * 1e3,
                     GBs=2 * m * d * X.element_size() / t / 1e9, GFLOPs=2 * m * d * (p + q) / t / 1e9)
            xb = torch.randn(16, d, device=dev, dtype=torch.bfloat16)
            t = timeit(lambda: op.apply_rows(xb), reps=20)
            emit(kernel=f"K3 ortho rows {gname} bf16 bs16 (activation side)", d=d, p=p, q=q, us=t * 1e6)
        # ---- K4
        H = correlated_H(d, dev)
        H = H + 0.01 * H.diag().mean() * torch.eye(d, device=dev)
        t_chol = timeit(lambda: torch.linalg.cholesky(H), reps=3, warm=1)
        C = torch.linalg.cholesky(H)
        t_ult = timeit(lambda: ops.unit_lower_t(C))
        t_k8 = timeit(lambda: ops.cholesky_lt(H, check=False), reps=3, warm=1)
        LT = ops.cholesky_lt(H)
        wg = ops.gridmap(W16, 'b', s, None, maxq)
        t = timeit(lambda: ops.ldlq_round(wg, LT, bits), reps=3, warm=1)
        emit(kernel="K4 ldlq_round", m=m, d=d, ms=t * 1e3, far_field_TFLOPs=m * d * d / t / 1e12,
             us_per_column=t / d * 1e6, k8_cholesky_lt_ms=t_k8 * 1e3, rocsolver_cholesky_ms=t_chol * 1e3,
             unit_lower_t_ms=t_ult * 1e3)
        # ---- whole Balance path on the GPU: preproc (rescale + blocked projection + damping) and fasterquant
        if not args.no_balance:
            from quip_amd import bal, quant
            layer = torch.nn.Linear(d, m, bias=False).to(dev).half()
            layer.weight.data = W16.clone()
            times = {}
            for lazy in (False,):
                layer.weight.data = W16.clone()
                b = bal.Balance(layer)
                b.configure('ldlq', bits, 0, False)
                b.quantizer = quant.Quantizer()
                b.quantizer.configure(bits, perchannel=True, sym=False, qfn='b', mse=False)
                b.H = correlated_H(d, dev).double()
                b.nsamples = 1
                b.post_batch()
                np.random.seed(0)
                torch.manual_seed(0)
                t0 = time.perf_counter()
                b.preproc(preproc_gptqH=True, percdamp=0.01, preproc_rescale=True, preproc_proj=True, preproc_proj_extra=0)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                b.fasterquant(lazy_batch=lazy)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                times = {"preproc_s": t1 - t0, "fasterquant_s": t2 - t1, "method_time_attr_s": b.time, "error": b.error}
                b.free()
            emit(kernel="Balance preproc+fasterquant (GPU wall, incl. host-side scipy factor sampling)", m=m, d=d, bits=bits, **times)


if __name__ == "__main__":
    main()
