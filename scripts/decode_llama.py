#!/usr/bin/env python3
"""Llama-2-7B w2 decode throughput on one MI355X -- the Llama half of the decode row (reference llama.py:418-471 `benchmark`:
per-token latency of the quantised model, batch 1, layer by layer).

A random-init decoder of the Llama-2-7B architecture (hidden 4096, intermediate 11008, 32 blocks, 32 heads of 128, vocab
32000, RMSNorm eps 1e-5, rotary theta 10000, SiLU gate * up, no biases, untied lm_head; no checkpoint is reachable offline) whose
224 decoder Linears (what llama.py quantises: find_layers over model.model.layers) are packed 2-bit QuantLinear layers in the
incoherence-processed form  y = U^T ( What2 ( V (x (/) s) ) ).  The single-token step is captured in ONE hipGraph.
Per block, fused variant:
    [RMSNorm folded into the V-side operator launch of q / k / v] -> grouped dequant-GEMM -> [U^T, tiled] -> rotary (one launch,
    in place) -> single-launch decode attention -> o_proj (V, GEMM, U^T + residual) -> [RMSNorm folded into V of gate / up] ->
    grouped dequant-GEMM -> U^T of gate / up (n = 11008 = 688 x 16: the general K3 launches) -> silu * mul -> down_proj
    (V general, GEMM, U^T + residual tiled).
The 4096-wide operators (64 x 64) run on the tiled small-batch kernels (csrc/ortho_tile.hip); the 11008-wide ones do not fit a
workgroup's LDS and take quipamd_ortho_apply_rows.

usage: python scripts/decode_llama.py [--layers 32] [--tokens 64] [--prompt 64] [--check]"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops, method, decode  # noqa: E402
from quip_amd.quant import (QuantLinear, packed_forward_fused, fused_stage, fused_ok, fused_attention, fused_attention_ok,  # noqa: E402
                            fused_u_only, packed_u_stage, packed_v_stage_gate, fused_bigp_tail, bigp_tail_ok, fused_head)
import decode_opt as D  # noqa: E402  (time_decode: hipGraph capture + per-token timing)


RMSNorm = decode.RMSNorm


class Decoder(decode.LlamaDecoder):
    """quip_amd.decode.LlamaDecoder (plain / fused / v3 / fused-head steps live in the package) over random-init modules"""

    def __init__(self, layers=32, h=4096, ffn=11008, heads=32, vocab=32000, maxpos=4096, eps=1e-5, theta=10000.0, dtype=torch.float16):
        hd = h // heads
        inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))          # LlamaRotaryEmbedding
        super().__init__(nn.Embedding(vocab, h, dtype=dtype), [decode.LlamaBlock.random(h, ffn, heads, eps, dtype) for _ in range(layers)],
                         RMSNorm(h, eps, dtype), nn.Linear(h, vocab, bias=False, dtype=dtype), heads, inv, maxpos)


def pack_model(model, bits, dev, seed=0, twin=True):
    """the 7 Linears of every block -> packed QuantLinear (nearest-rounded qfn-b codes, random Kronecker U / V, random scaleWH);
    returns ({(layer, name): dense equivalent weight} if twin, packed bytes)."""
    np.random.seed(seed)
    torch.manual_seed(seed)
    maxq = 2 ** bits - 1
    dense, nbytes = {}, 0
    for li, blk in enumerate(model.blocks):
        for name in Decoder.NAMES:
            lin = getattr(blk, name)
            m, d = lin.weight.shape
            W = lin.weight.data
            s = ops.qfnb_scale(W)
            What, codes = ops.quantize(W, 'b', s, None, maxq, want_codes=True)
            U = ops.OrthoOp(method.gen_rand_ortho_butterfly_noblock(m), dev)      # every layer its own operators, like the reference
            V = ops.OrthoOp(method.gen_rand_ortho_butterfly_noblock(d), dev)
            sWH = (0.5 + torch.rand(d)).to(dev)
            ql = QuantLinear(d, m, bits=bits, qfn='b').to(dev)
            ql.pack(codes, s, None, bias=None, scaleWH=sWH, U=U, V=V)
            setattr(blk, name, ql)
            nbytes += ql.qweight.numel() * 4
            if twin:
                Wd = U.apply_cols(V.apply_rows(What.float(), transpose=True), transpose=True) / sWH[None, :]
                dense[(li, name)] = Wd.to(W.dtype)
    return dense, nbytes


def build(layers, dev, dtype, small=False):
    torch.manual_seed(0)
    kw = dict(h=2048, ffn=11008, heads=16) if small else {}     # small: 64 x 32 operators; 11008 keeps the 688 x 16 general path
    model = Decoder(layers=layers, dtype=dtype, **kw).to(dev).eval()
    for p_ in model.parameters():
        if p_.dim() > 1:
            p_.data.normal_(0, 0.02)
    return model


def run(layers=32, bits=2, bs=1, prompt=64, tokens=64, with_dense=True):
    dev, dtype = torch.device("cuda:0"), torch.float16
    maxlen = prompt + tokens + 8
    model = build(layers, dev, dtype)
    out = {"config": {"arch": "Llama-2-7B (hidden 4096, intermediate 11008, heads 32 x 128, vocab 32000)", "layers": layers, "bits": bits,
                      "bs": bs, "prompt": prompt, "tokens": tokens, "launch": "hipGraph",
                      "weights": "random init, nearest-rounded qfn-b codes, Kronecker U/V (64x64, 688x16), random scaleWH"}}
    if with_dense:
        med, _, _ = D.time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, False)
        out["dense_fp16"] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                             "what": "fp16 nn.Linear (rocBLAS) with the same rotary / attention launches"}
    _, nbytes = pack_model(model, bits, dev, twin=False)
    med, _, _ = D.time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, False)
    head = model.lm_head.weight.numel() * 2
    out["packed_w%d" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med, "packed_weight_MB": nbytes / 1e6,
                                "hbm_bound_tok_per_s": 8e12 / (nbytes + head), "what": "QuantLinear.forward per layer (3 launches each), torch RMSNorm / silu"}
    for blk in model.blocks:
        blk.fused = True
    med, _, _ = D.time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, False)
    out["packed_w%d_fused" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                      "what": "q/k/v and gate/up grouped; RMSNorm folded into the V-side launch; residual folded into U^T; "
                                              "4096-wide operators tiled over 16 workgroups"}
    if model.v3_ok(bs):
        model.v3 = True
        med, _, _ = D.time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, False)
        out["packed_w%d_v3" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                       "what": "csrc/decode_fused.hip / decode_attn.hip: operator chains in the prologues of the 4096-wide GEMMs and of the "
                                               "attention launch (rotary included); the 11008-wide operators cut over p (csrc/decode_bigp.hip): 6 launches per block"}
        model.fused_head = True
        med, _, _ = D.time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, False)
        out["packed_w%d_v3_head" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                            "what": "as v3 + csrc/decode_head.hip: the embedding (with the previous step's argmax) and [U_down^T + residual -> "
                                                    "RMSNorm -> lm_head -> argmax partials] as one launch each: 6 launches per block + 2 per token"}
        model.fused_head = False
        model.v3 = False
    del model
    torch.cuda.empty_cache()
    return out


def decode_check(layers=2, bits=2, small=True):
    """packed model (plain and fused) vs its dense twin on the same 4 tokens: relative logits error."""
    dev, dtype = torch.device("cuda:0"), torch.float16
    model = build(layers, dev, dtype, small=small)
    twin, _ = pack_model(model, bits, dev)
    torch.manual_seed(1)
    _, _, lq = D.time_decode(model, 2, 0, 3, 32, dev, dtype, True)
    for blk in model.blocks:
        blk.fused = True
    torch.manual_seed(1)
    _, _, lf = D.time_decode(model, 2, 0, 3, 32, dev, dtype, True)
    l3 = None
    if model.v3_ok(2):
        model.v3 = True
        torch.manual_seed(1)
        _, _, l3 = D.time_decode(model, 2, 0, 3, 32, dev, dtype, True)
        model.v3 = False
    for blk in model.blocks:
        blk.fused = False
    for (li, name), Wd in twin.items():
        lin = nn.Linear(Wd.shape[1], Wd.shape[0], bias=False, dtype=dtype, device=dev)
        lin.weight.data = Wd
        setattr(model.blocks[li], name, lin)
    torch.manual_seed(1)
    _, _, ld = D.time_decode(model, 2, 0, 3, 32, dev, dtype, True)
    return float((lq - ld).norm() / ld.norm()), float((lf - ld).norm() / ld.norm()), None if l3 is None else float((l3 - ld).norm() / ld.norm())


if __name__ == "__main__":
    if "--check" in sys.argv:
        e1, e2, e3 = decode_check()
        print(json.dumps({"llama_decode_logits_rel_err_packed_vs_dense_twin": e1, "fused_vs_dense_twin": e2, "v3_vs_dense_twin": e3}))
    else:
        ap = argparse.ArgumentParser()
        ap.add_argument("--layers", type=int, default=32)
        ap.add_argument("--bits", type=int, default=2)
        ap.add_argument("--bs", type=int, default=1)
        ap.add_argument("--prompt", type=int, default=64)
        ap.add_argument("--tokens", type=int, default=64)
        ap.add_argument("--no-dense", action="store_true")
        a = ap.parse_args()
        print(json.dumps(run(a.layers, a.bits, a.bs, a.prompt, a.tokens, with_dense=not a.no_dense)))
