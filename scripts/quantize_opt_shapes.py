#!/usr/bin/env python3
"""Whole-model LDLQ quantisation time at OPT-1.3B Linear shapes (BASELINE.json config "B"), synthetic W / H, one GPU:
for each of the 24 x {4 x 2048x2048, 8192x2048, 2048x8192} Linears run the reference call sequence
post_batch -> preproc(gptqH, rescale, proj) -> Balance.fasterquant (LDLQ w2 qfn b) -> free, all on the MI355X.
Prints per-shape and total wall time next to the reference's CPU figures recorded in BASELINE.md section 2.
--blocks N limits the run to N transformer blocks (per-block time is constant; total is extrapolated)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import bal, quant  # noqa: E402

CPU_REF_S = {"2048x2048": (1.5, 1.0), "8192x2048": (10.4, 4.5), "2048x8192": (52.1, 29.3)}   # (round_ldl, lazy) BASELINE.md


def llama(args, dev):
    """Llama-2-7B Linear shapes (llama.py:87-156 quantises q,k,v,o, gate, up, down per block), same call sequence."""
    from quip_amd import method
    shapes = [(4096, 4096)] * 4 + [(11008, 4096)] * 2 + [(4096, 11008)]
    Hs = {}
    for d in (4096, 11008):
        g = torch.Generator().manual_seed(d)
        X = torch.randn(d + 256, d, generator=g).to(dev)
        Hs[d] = (X.T @ X / (d + 256)).double()
        del X
    np.random.seed(0)
    torch.manual_seed(0)
    per = {}
    for blk in range(args.blocks):
        for (m, d) in shapes:
            layer = torch.nn.Linear(d, m, bias=False).to(dev).half()
            layer.weight.data = (0.02 * torch.randn(m, d)).to(dev).half()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b = bal.Balance(layer)
            b.configure('ldlq', args.bits, 0, False)
            b.quantizer = quant.Quantizer()
            b.quantizer.configure(args.bits, perchannel=True, sym=False, qfn='b', mse=False)
            b.H = Hs[d].clone()
            b.nsamples = 1
            b.post_batch()
            b.preproc(preproc_gptqH=True, percdamp=0.01, preproc_rescale=True, preproc_proj=True, preproc_proj_extra=args.proj_extra)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            b.fasterquant(lazy_batch=False)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            b.free()
            if blk > 0 or args.blocks == 1:
                e = per.setdefault(f"{m}x{d}", {"n": 0, "preproc_s": 0.0, "fasterquant_s": 0.0})
                e["n"] += 1
                e["preproc_s"] += t1 - t0
                e["fasterquant_s"] += t2 - t1
    counts = {"4096x4096": 4, "11008x4096": 2, "4096x11008": 1}
    out = {"model": "llama-2-7b shapes, synthetic W / H", "blocks_run": args.blocks, "per_layer": {}}
    tot = 0.0
    for k, e in per.items():
        n = max(e["n"], 1)
        out["per_layer"][k] = {"preproc_s": round(e["preproc_s"] / n, 4), "fasterquant_s": round(e["fasterquant_s"] / n, 4)}
        tot += 32 * counts[k] * (e["preproc_s"] + e["fasterquant_s"]) / n
    out["total_s_extrapolated_32_blocks"] = round(tot, 2)
    hess = {}
    if not args.no_hessian:
        for d in (4096, 11008):
            layer = torch.nn.Linear(d, 16, bias=False).to(dev).half()
            x = torch.randn(1, args.seqlen, d, device=dev).half()
            qm = method.QuantMethod(layer)
            qm.add_batch(x, None)
            torch.cuda.synchronize()
            qm = method.QuantMethod(layer)
            t0 = time.perf_counter()
            for _ in range(args.nsamples):
                qm.add_batch(x, None)
            qm.post_batch()
            torch.cuda.synchronize()
            hess[str(d)] = round(time.perf_counter() - t0, 4)
            del qm
        hb = 6 * hess["4096"] + hess["11008"]                    # q,k,v,o,gate,up see 4096 features; down sees 11008
        out["hessian"] = {"nsamples": args.nsamples, "seqlen": args.seqlen, "per_linear_k7_s": hess, "per_block_s": round(hb, 3),
                          "32_blocks_s": round(32 * hb, 2)}
        out["total_incl_hessian_s"] = round(tot + 32 * hb, 2)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=2)
    ap.add_argument("--bits", type=int, default=2)
    ap.add_argument("--nsamples", type=int, default=128, help="calibration samples per Linear for the Hessian leg (opt.py default)")
    ap.add_argument("--seqlen", type=int, default=2048)
    ap.add_argument("--no-hessian", action="store_true")
    ap.add_argument("--proj-extra", type=int, default=0, choices=[0, 1, 2],
                    help="preproc_proj_extra: 0 = blocked butterfly (what opt.py --incoh_processing effectively runs, it sets the "
                         "unused `proj_extra`), 1 = Kronecker (the paper's operator; single-launch row-walking K3)")
    ap.add_argument("--fast-hessian", action="store_true", help="K7's opt-in 16-bit-MFMA mode (method.HESSIAN_FAST)")
    ap.add_argument("--device-rng", action="store_true", help="opt-in method.DEVICE_RNG: operator sampling without the host RNG")
    ap.add_argument("--model", default="opt1p3b", choices=["opt1p3b", "llama7b"],
                    help="llama7b: BASELINE configs[3] shapes (32 blocks x {4 x 4096^2, 2 x 11008x4096, 4096x11008}); no CPU figures exist")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.fast_hessian:
        from quip_amd import method as _m
        _m.HESSIAN_FAST = True
    if args.device_rng:
        from quip_amd import method as _m
        _m.DEVICE_RNG = True
    if args.model == "llama7b":
        return llama(args, dev)
    shapes = [(2048, 2048)] * 4 + [(8192, 2048), (2048, 8192)]
    Hs = {}
    for d in (2048, 8192):
        g = torch.Generator().manual_seed(d)
        X = torch.randn(d + 256, d, generator=g).to(dev)
        Hs[d] = (X.T @ X / (d + 256)).double()
    np.random.seed(0)
    torch.manual_seed(0)
    per_shape = {}
    t_all0 = time.perf_counter()
    for blk in range(args.blocks):
        for (m, d) in shapes:
            layer = torch.nn.Linear(d, m, bias=False).to(dev).half()
            layer.weight.data = (0.02 * torch.randn(m, d)).to(dev).half()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b = bal.Balance(layer)
            b.configure('ldlq', args.bits, 0, False)
            b.quantizer = quant.Quantizer()
            b.quantizer.configure(args.bits, perchannel=True, sym=False, qfn='b', mse=False)
            b.H = Hs[d].clone()
            b.nsamples = 1
            b.post_batch()
            b.preproc(preproc_gptqH=True, percdamp=0.01, preproc_rescale=True, preproc_proj=True, preproc_proj_extra=args.proj_extra)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            b.fasterquant(lazy_batch=False)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            b.free()
            k = f"{m}x{d}"
            e = per_shape.setdefault(k, {"n": 0, "preproc_s": 0.0, "fasterquant_s": 0.0})
            if blk > 0 or args.blocks == 1:                  # first block pays one-off allocator / solver warm-up
                e["n"] += 1
                e["preproc_s"] += t1 - t0
                e["fasterquant_s"] += t2 - t1
    wall = time.perf_counter() - t_all0
    # Hessian accumulation as opt.py drives it: nsamples add_batch calls of [1, seqlen, d] fp16 per Linear, then
    # post_batch (method.py:98-123) -- K7 on the GPU; the reference's op (fp64 GEMM per call) timed next to it
    hess = {}
    if not args.no_hessian:
        from quip_amd import method
        for d in (2048, 8192):
            layer = torch.nn.Linear(d, 16, bias=False).to(dev).half()
            x = torch.randn(1, args.seqlen, d, device=dev).half()
            qm = method.QuantMethod(layer)
            qm.add_batch(x, None)
            torch.cuda.synchronize()
            qm = method.QuantMethod(layer)
            t0 = time.perf_counter()
            for _ in range(args.nsamples):
                qm.add_batch(x, None)
            qm.post_batch()
            torch.cuda.synchronize()
            t_k7 = time.perf_counter() - t0
            Href = torch.zeros(d, d, dtype=torch.float64, device=dev)
            reps = max(args.nsamples // 8, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                x64 = x[0].t().to(torch.float64)
                Href.add_(x64.matmul(x64.t()))
            torch.cuda.synchronize()
            t_ref = (time.perf_counter() - t0) * args.nsamples / reps
            hess[str(d)] = {"k7_s": round(t_k7, 4), "fp64_gemm_s": round(t_ref, 4)}
            del Href, qm
    out = {"blocks_run": args.blocks, "wall_s": wall, "per_layer": {}}
    tot = tot_cpu = tot_cpu_lazy = 0.0
    counts = {"2048x2048": 4, "8192x2048": 1, "2048x8192": 1}
    for k, e in per_shape.items():
        n = max(e["n"], 1)
        p, f = e["preproc_s"] / n, e["fasterquant_s"] / n
        out["per_layer"][k] = {"preproc_s": round(p, 4), "fasterquant_s": round(f, 4), "cpu_ref_fasterquant_s": CPU_REF_S[k][0],
                               "cpu_ref_lazy_s": CPU_REF_S[k][1], "speedup_fasterquant": round(CPU_REF_S[k][0] / f, 1)}
        tot += 24 * counts[k] * (p + f)
        tot_cpu += 24 * counts[k] * CPU_REF_S[k][0]
        tot_cpu_lazy += 24 * counts[k] * CPU_REF_S[k][1]
    out["opt1p3b_total_s_extrapolated_24_blocks"] = round(tot, 2)
    if hess:
        # per block: q, k, v, out, fc1 see d = 2048 inputs, fc2 sees d = 8192 (the reference accumulates q/k/v separately)
        hb = 5 * hess["2048"]["k7_s"] + hess["8192"]["k7_s"]
        hb_ref = 5 * hess["2048"]["fp64_gemm_s"] + hess["8192"]["fp64_gemm_s"]
        out["hessian"] = {"nsamples": args.nsamples, "seqlen": args.seqlen, "per_linear": hess,
                          "per_block_s": round(hb, 3), "per_block_fp64_gemm_s": round(hb_ref, 3),
                          "opt1p3b_24_blocks_s": round(24 * hb, 2), "opt1p3b_24_blocks_fp64_gemm_s": round(24 * hb_ref, 2)}
        out["opt1p3b_total_incl_hessian_s"] = round(tot + 24 * hb, 2)
    out["cpu_reference_ldlq_only_s"] = {"round_ldl": tot_cpu, "lazy_batch": tot_cpu_lazy, "source": "BASELINE.md section 2, 8 cores"}
    out["speedup_vs_cpu_round_ldl_incl_our_preproc"] = round(tot_cpu / tot, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
