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
from quip_amd import ops, method  # noqa: E402
from quip_amd.quant import (QuantLinear, packed_forward_fused, fused_stage, fused_ok, fused_attention, fused_attention_ok,  # noqa: E402
                            fused_u_only, packed_u_stage, packed_v_stage_gate, fused_bigp_tail, bigp_tail_ok, fused_head)
import decode_opt as D  # noqa: E402  (time_decode: hipGraph capture + per-token timing)


class RMSNorm(nn.Module):
    def __init__(self, h, eps, dtype):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(h, dtype=dtype))
        self.eps = eps

    def forward(self, x):                                   # HF LlamaRMSNorm: fp32 statistics, cast, then the gain
        xf = x.float()
        return self.weight * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)).to(x.dtype)


class Block(nn.Module):
    def __init__(self, h, ffn, heads, eps, dtype):
        super().__init__()
        self.h, self.heads, self.hd = h, heads, h // heads
        self.n1, self.n2 = RMSNorm(h, eps, dtype), RMSNorm(h, eps, dtype)
        mk = lambda i, o: nn.Linear(i, o, bias=False, dtype=dtype)
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = mk(h, h), mk(h, h), mk(h, h), mk(h, h)
        self.gate_proj, self.up_proj, self.down_proj = mk(h, ffn), mk(h, ffn), mk(ffn, h)
        self.fused = False

    def forward(self, x, kc, vc, pos, cos, sin):
        """x [bs, h]; kc / vc [bs, heads, maxlen, hd]; pos int64 [1] on the device; cos / sin fp32 [maxpos, hd]."""
        if self.fused:
            q, k, v = packed_forward_fused([self.q_proj, self.k_proj, self.v_proj], x, ln=self.n1)
        else:
            hn = self.n1(x)
            q, k, v = self.q_proj(hn), self.k_proj(hn), self.v_proj(hn)
        ops.rope_inplace(q, k, cos, sin, pos, self.heads)
        o = ops.decode_attention(q, k, v, kc, vc, pos)
        if self.fused:
            x = packed_forward_fused([self.o_proj], o, residual=x)[0]
            g, u = packed_forward_fused([self.gate_proj, self.up_proj], x, ln=self.n2)
            return packed_forward_fused([self.down_proj], g, residual=x, gate_up=u)[0]       # silu(g) * u formed inside the V launch
        x = x + self.o_proj(o)
        hn = self.n2(x)
        return x + self.down_proj(F.silu(self.gate_proj(hn)) * self.up_proj(hn))


class Decoder(nn.Module):
    NAMES = ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]

    def __init__(self, layers=32, h=4096, ffn=11008, heads=32, vocab=32000, maxpos=4096, eps=1e-5, theta=10000.0, dtype=torch.float16):
        super().__init__()
        self.h, self.layers_n, self.heads = h, layers, heads
        self.tok = nn.Embedding(vocab, h, dtype=dtype)
        self.blocks = nn.ModuleList([Block(h, ffn, heads, eps, dtype) for _ in range(layers)])
        self.norm = RMSNorm(h, eps, dtype)
        self.lm_head = nn.Linear(h, vocab, bias=False, dtype=dtype)
        hd = h // heads
        inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))          # LlamaRotaryEmbedding
        fr = torch.outer(torch.arange(maxpos, dtype=torch.float32), inv)
        emb = torch.cat((fr, fr), dim=-1)
        self.register_buffer("cos", emb.cos().contiguous(), persistent=False)
        self.register_buffer("sin", emb.sin().contiguous(), persistent=False)

    def _apply(self, fn, recurse=True):                     # .to(dev) / .half() must not narrow the rotary tables
        cos, sin = self.cos, self.sin
        super()._apply(fn, recurse)
        self.cos, self.sin = cos.to(self.tok.weight.device), sin.to(self.tok.weight.device)
        return self

    v3 = False               # csrc/decode_fused.hip + decode_attn.hip + decode_bigp.hip: 6 launches per block (13+ in the round-2 fused variant)

    def v3_ok(self, bs):
        b = self.blocks[0]
        qkv = [b.q_proj, b.k_proj, b.v_proj]
        return (isinstance(b.q_proj, QuantLinear) and fused_ok(qkv, bs, prev=b.down_proj) and fused_ok([b.o_proj], bs, norm=False)
                and fused_ok([b.gate_proj, b.up_proj], bs, prev=b.o_proj) and b.down_proj.U is not None and b.down_proj.U.fused_ok
                and bigp_tail_ok([b.gate_proj, b.up_proj], b.down_proj, bs))

    def step_v3(self, x, pos, caches):
        prev, yd, x = self.blocks_v3(x, pos, caches)
        return fused_u_only(prev, yd.to(torch.float16), residual=x)

    fused_head = False       # with v3: embedding (+ the previous step's argmax) and [U_down^T + residual -> final RMSNorm -> lm_head -> argmax
                             # partials, pos += 1] as one launch each (csrc/decode_head.hip)

    def step_fused_head(self, ids, pos, caches, logits, part_val, part_idx):
        x = torch.empty((ids.numel(), self.h), dtype=torch.float16, device=ids.device)
        ops.decode_embed(self.tok.weight, ids, x, part_val=part_val, part_idx=part_idx)
        prev, yd, x = self.blocks_v3(x, pos, caches)
        return fused_head(prev, yd, x, self.norm, self.lm_head.weight, logits, part_val, part_idx, pos_inc=pos)

    def blocks_v3(self, x, pos, caches):
        """per block, six launches: [U_down^T(prev) + residual -> RMSNorm -> V_qkv -> GEMM q,k,v] [U_qkv^T + rotary + attention]
        [V_o -> GEMM o] [U_o^T + residual -> RMSNorm -> V_gate/up -> GEMM gate, up] [U_gate^T, U_up^T (/) s: 688 x 16, decode_bigp.hip]
        [silu * up -> V_down -> GEMM down, K-slices through fp32 atomics]; a 688 x 688 factor is 0.9 MB, not a workgroup's pass: the
        11008-wide operators are cut over the p index (csrc/decode_bigp.hip)"""
        h16 = torch.float16
        prev, yd = None, None
        for blk, (kc, vc) in zip(self.blocks, caches):
            qkv = [blk.q_proj, blk.k_proj, blk.v_proj]
            if prev is None:
                ys, _ = fused_stage(qkv, x=x, ln=blk.n1, y_dtype=h16)
            else:
                ys, x = fused_stage(qkv, prev=prev, y_prev=yd, residual=x, ln=blk.n1, store=True, y_dtype=h16)
            o = fused_attention(qkv, ys, kc, vc, pos, self.cos, self.sin)
            yo = fused_stage([blk.o_proj], x=o, y_dtype=h16)[0][0]
            gu = [blk.gate_proj, blk.up_proj]
            ygu, x = fused_stage(gu, prev=blk.o_proj, y_prev=yo, residual=x, ln=blk.n2, store=True, y_dtype=h16)
            yd = fused_bigp_tail(gu, blk.down_proj, ygu)                       # fp32 accumulator, ZT order of down_proj's U
            prev = blk.down_proj
        return prev, yd, x

    def step(self, ids, pos, caches, arange):
        x = self.tok(ids)
        if self.v3:
            return self.lm_head(self.norm(self.step_v3(x, pos, caches)))
        for blk, (kc, vc) in zip(self.blocks, caches):
            x = blk(x, kc, vc, pos, self.cos, self.sin)
        return self.lm_head(self.norm(x))


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
