#!/usr/bin/env python3
"""BASELINE configs[4] on ONE MI355X: one transformer block at the OPT-30B geometry (hidden 7168, ffn 28672, 56 heads) through the
reference's own opt_sequential (opt.py:29-190, the staged copy) on quip_amd -- LDLQ w2 + incoherence processing, the reference's default
calibration size (128 x 2048 tokens) -- with the phase split of scripts/run_full_model.py, followed by every kernel of the d = 28672
side timed alone (what the fc2 Linear, 7168 x 28672, costs: 2/3 of a block's LDLQ work):

    K7  one add_batch call, X [2048, 28672] fp16 -> H fp64                       (method.py:98-123)
    op  gen_rand_ortho_butterfly(28672): 64 Haar 448 x 448 + 448 Haar 64 x 64   (method.py:20-31; host RNG)
    K3  W <- U W V^T on 7168 x 28672 and H <- V H V^T on 28672^2                (method.py:173-176)
    K8  LDL factor of the 28672^2 Hessian                                        (vector_balance.py:171-173)
    K4  LDLQ sweep, 7168 rows x 28672 columns, w2                                (vector_balance.py:155-199)
    K5/K1  grid map + pack of the codes                                          (quant.py:10-15, zeroShot/models/quant.py:190-199)

    python scripts/opt30b_block.py --out profiles/r05_opt30b_block.json [--blocks 1] [--nsamples 128]"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def ev(fn, reps=2, warm=1):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def kernels_alone(dev):
    from quip_amd import ops, method as M
    m, d, bits = 7168, 28672, 2
    out = {}
    x = torch.randn(2048, d, device=dev).half()
    Hacc = torch.zeros(d, d, dtype=torch.float64, device=dev)
    t = ev(lambda: ops.hessian_accum(Hacc, x), reps=3)
    tiles = (d // 128) * (d // 128 + 1) // 2
    out["K7_add_batch_2048_tokens"] = {"ms": round(t, 2), "fp64_TFLOPs": round(2.0 * 2048 * tiles * 128 * 128 / t / 1e9, 1), "fp64_mfma_peak_TFLOPs": 78.6,
                                      "accumulator_GB": round(Hacc.numel() * 8 / 2 ** 30, 2)}
    H = ops.hessian_finish(Hacc, 5.0)
    del Hacc, x
    torch.cuda.empty_cache()
    # a well-conditioned Hessian for the factor: the calibration-like one above has rank 2048 -- add the trace ridge preproc adds
    X = torch.randn(d + 512, d, device=dev) * (torch.arange(1, d + 1, device=dev, dtype=torch.float32) ** -0.5)
    H = X.t() @ X / (d + 512)
    del X
    np.random.seed(0)
    torch.manual_seed(0)
    t0 = time.perf_counter()
    Vb = M.gen_rand_ortho_butterfly(d)
    t_v = time.perf_counter() - t0
    t0 = time.perf_counter()
    Ub = M.gen_rand_ortho_butterfly(m)
    t_u = time.perf_counter() - t0
    out["operator_sampling_host"] = {"V_28672_448x64_s": round(t_v, 3), "U_7168_224x32_s": round(t_u, 3),
                                     "factor_MB_fp32": round((64 * 448 * 448 + 448 * 64 * 64) * 4 / 2 ** 20, 1)}
    U, V = ops.OrthoOp(Ub, dev), ops.OrthoOp(Vb, dev)
    W = (0.02 * torch.randn(m, d, device=dev))
    t = ev(lambda: U.apply_cols(V.apply_rows(W)))
    out["K3_W_both_sides_7168x28672"] = {"ms": round(t, 2), "GBs_of_2_reads_2_writes": round(4 * m * d * 4 / t / 1e6, 0)}
    t = ev(lambda: V.apply_rows(V.apply_rows(H).t().contiguous()))
    out["K3_H_both_sides_28672"] = {"ms": round(t, 2), "GBs_of_3_reads_3_writes": round(6 * d * d * 4 / t / 1e6, 0)}
    H = ops.preproc_trace_ridge(H, 1e-2)
    H = V.apply_rows(V.apply_rows(H).t().contiguous())
    H.diagonal().add_(0.01 * H.diagonal().mean())
    t = ev(lambda: ops.cholesky_lt(H, check=False), reps=2)
    out["K8_ldl_factor_28672"] = {"ms": round(t, 1), "fp32_TFLOPs": round(d ** 3 / 3 / t / 1e9, 1), "fp32_matrix_peak_TFLOPs": 157.3, "LT_GB": round(d * d * 4 / 2 ** 30, 2)}
    LT = ops.cholesky_lt(H)
    del H
    Wp = U.apply_cols(V.apply_rows(W))
    scale = ops.qfnb_scale(Wp)
    Wg = ops.gridmap(Wp, "b", scale, None, 3)
    t = ev(lambda: ops.ldlq_round(Wg, LT, bits), reps=2)
    out["K4_ldlq_sweep_7168x28672"] = {"ms": round(t, 1), "far_field_TFLOPs": round(m * d * d / t / 1e9, 1), "us_per_column": round(t * 1e3 / d, 2)}
    codes = ops.ldlq_round(Wg, LT, bits)
    t = ev(lambda: ops.pack(codes, bits, ops.LAYOUT_STREAM), reps=3)
    out["K1_pack_stream"] = {"ms": round(t, 3), "packed_MB": round(m * d * bits / 8 / 2 ** 20, 1)}
    t = ev(lambda: ops.gridmap(Wp, "b", scale, None, 3), reps=3)
    out["K5_gridmap"] = {"ms": round(t, 3)}
    # the packed layer at this shape, one decode row group (bs 16): K2
    qs = ops.pack(codes, bits, ops.LAYOUT_STREAM)
    xx = torch.randn(16, d, device=dev).to(torch.bfloat16)
    t = ev(lambda: ops.dequant_gemm(xx, qs, bits, "b", scale, None, None), reps=20, warm=3)
    out["K2_dequant_gemm_bs16"] = {"us": round(t * 1e3, 2), "GBs": round((m * d // 4 + 2 * 16 * (m + d)) / t / 1e6, 0)}
    out["peak_device_GB"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=1)
    ap.add_argument("--nsamples", type=int, default=128)
    ap.add_argument("--seqlen", type=int, default=2048)
    ap.add_argument("--out", default=None)
    ap.add_argument("--skip-driver", action="store_true")
    ap.add_argument("--skip-kernels", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {"what": "BASELINE configs[4] geometry on one MI355X: OPT-30B block(s) (7168 / 28672 / 56 heads) through the reference's opt_sequential "
                   "on quip_amd, LDLQ w2 + incoherence processing; then the d = 28672 kernels alone"}
    if not a.skip_driver:
        import run_full_model as F
        ns = types.SimpleNamespace(model="opt-30b", nsamples=a.nsamples, seqlen=a.seqlen, layers=a.blocks, wbits=None, quant="ldlq", no_incoh=False, extra=0,
                                   restatement=False, fast_hessian=False, device_rng=False, prefetch_operators=True, out=None)
        res["block_through_reference_driver"] = F.run(ns)
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
    if not a.skip_kernels:
        res["kernels_alone_fc2_7168x28672"] = kernels_alone(dev)
    line = json.dumps(res)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(line + "\n")


if __name__ == "__main__":
    main()
