#!/usr/bin/env python3
"""OPT-1.3B w2 decode throughput on one MI355X (BASELINE.json configs[2]; SURVEY.md 8(d) "B decode").

A random-init decoder of the OPT-1.3B architecture (hidden 2048, ffn 8192, 24 blocks, 32 heads, vocab 50272, pre-LN,
ReLU, learned positions offset 2, tied lm_head -- transformers' OPTConfig defaults for that size; no checkpoint is
available offline) whose 144 decoder Linears (what opt.py quantises: find_layers over model.decoder.layers) are
replaced by packed 2-bit QuantLinear layers in the incoherence-processed form
        y = U^T ( What2 ( V (x (/) s) ) ) + bias                                        
with codes from round-to-nearest on the qfn-b grid (fast; LDLQ gives different codes, same kernels and bytes), random
Kronecker U / V (method.gen_rand_ortho_butterfly_noblock) and a random positive scaleWH.
The single-token step (all 24 blocks + head + greedy argmax, static KV cache, position on the device) is captured in
ONE hipGraph and replayed per token, the measurement the reference's dead benchmark() (opt.py:431-482) describes:
median per-token latency, batch 1.  For context the same harness times the dense fp16 model (what the reference
actually runs at inference).  A logits check against the dense model built from the SAME dequantised weights with
the transforms folded in guards the packed path.

usage: python scripts/decode_opt.py [--layers 24] [--tokens 128] [--prompt 128] [--eager]"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops, method, decode  # noqa: E402
from quip_amd.quant import (QuantLinear, packed_forward_fused, packed_v_stage, packed_gemm_stage, packed_u_stage,  # noqa: E402
                            packed_u_then_v, packed_vgemm_stage, vgemm_fusable, fused_stage, fused_ok, fused_attention, fused_attention_ok, fused_u_only,
                            fused_head, fused_head_ok)


class Decoder(decode.OPTDecoder):
    """quip_amd.decode.OPTDecoder (the package's engine: plain / fused / v3 / fused-head steps) built from random-init modules of the
    given geometry, plus the launch sequences of rounds 1-2 kept here as timing variants (chained, vfused, tiled)."""

    def __init__(self, layers=24, h=2048, ffn=8192, heads=32, vocab=50272, maxpos=2048, dtype=torch.float16):
        blocks = [decode.OPTBlock.random(h, ffn, heads, dtype) for _ in range(layers)]
        for b in blocks:
            b.fused_attn = False     # the variants below switch the single-launch attention on themselves
        super().__init__(nn.Embedding(vocab, h, dtype=dtype), nn.Embedding(maxpos + 2, h, dtype=dtype), blocks, nn.LayerNorm(h, dtype=dtype), heads)

    chained = False          # packed + fused attention + U^T->LN->V chains across layers: 10 launches per block

    def step_chained(self, x, pos, caches):
        """the packed block sequence with every `U^T y + residual -> LayerNorm -> V (x (/) s)` hand-over between two
        consecutive packed layers done in one launch (quant.packed_u_then_v), including across block boundaries."""
        dt = x.dtype
        b0 = self.blocks[0]
        xts = packed_v_stage([b0.q_proj, b0.k_proj, b0.v_proj], x, ln=b0.ln1)
        for i, (blk, (kc, vc)) in enumerate(zip(self.blocks, caches)):
            qkv = [blk.q_proj, blk.k_proj, blk.v_proj]
            q, k, v = packed_u_stage(qkv, packed_gemm_stage(qkv, xts), dt)
            o = ops.decode_attention(q, k, v, kc, vc, pos)
            yo = packed_gemm_stage([blk.out_proj], packed_v_stage([blk.out_proj], o))[0]
            x, xt1 = packed_u_then_v(blk.out_proj, yo, dt, [blk.fc1], residual=x, ln=blk.ln2)
            y1 = packed_gemm_stage([blk.fc1], xt1)[0]
            _, xt2 = packed_u_then_v(blk.fc1, y1, dt, [blk.fc2], relu=True, store=False)
            y2 = packed_gemm_stage([blk.fc2], xt2)[0]
            if i + 1 < len(self.blocks):
                nb = self.blocks[i + 1]
                x, xts = packed_u_then_v(blk.fc2, y2, dt, [nb.q_proj, nb.k_proj, nb.v_proj], residual=x, ln=nb.ln1)
            else:
                x = packed_u_stage([blk.fc2], [y2], dt, residual=x)[0]
        return x

    vfused = False           # the V-side operator in the prologue of the dequant-GEMM for the d = 2048 inputs: 9 launches per block

    def step_vfused(self, x, pos, caches):
        """q/k/v, out_proj and fc1 take their input through quant.packed_vgemm_stage (operator + GEMM in one launch);
        fc1 -> fc2 (n = 8192) stays a chained operator launch."""
        dt = x.dtype
        for blk, (kc, vc) in zip(self.blocks, caches):
            qkv = [blk.q_proj, blk.k_proj, blk.v_proj]
            q, k, v = packed_u_stage(qkv, packed_vgemm_stage(qkv, x, ln=blk.ln1), dt)
            o = ops.decode_attention(q, k, v, kc, vc, pos)
            x = packed_u_stage([blk.out_proj], packed_vgemm_stage([blk.out_proj], o), dt, residual=x)[0]
            y1 = packed_vgemm_stage([blk.fc1], x, ln=blk.ln2)[0]
            if self.split_handover:      # n = 8192: two tiled launches (32 workgroups each) beat the one-workgroup chain (17.4 us)
                xt2 = packed_v_stage([blk.fc2], packed_u_stage([blk.fc1], [y1], dt, relu=True)[0])
            else:
                _, xt2 = packed_u_then_v(blk.fc1, y1, dt, [blk.fc2], relu=True, store=False)
            x = packed_u_stage([blk.fc2], packed_gemm_stage([blk.fc2], xt2), dt, residual=x)[0]
        return x

    split_handover = False

    tiled = False            # every operator application cut into 16 x 16 output tiles over 8-32 workgroups (csrc/ortho_tile.hip): 13 launches

    def step_tiled(self, x, pos, caches):
        """V-op, GEMM, U-op as three launches per packed layer group, each operator launch spread over many CUs."""
        dt = x.dtype
        for blk, (kc, vc) in zip(self.blocks, caches):
            qkv = [blk.q_proj, blk.k_proj, blk.v_proj]
            q, k, v = packed_u_stage(qkv, packed_gemm_stage(qkv, packed_v_stage(qkv, x, ln=blk.ln1)), dt)
            o = ops.decode_attention(q, k, v, kc, vc, pos)
            x = packed_u_stage([blk.out_proj], packed_gemm_stage([blk.out_proj], packed_v_stage([blk.out_proj], o)), dt, residual=x)[0]
            h = packed_u_stage([blk.fc1], packed_gemm_stage([blk.fc1], packed_v_stage([blk.fc1], x, ln=blk.ln2)), dt, relu=True)[0]
            x = packed_u_stage([blk.fc2], packed_gemm_stage([blk.fc2], packed_v_stage([blk.fc2], h)), dt, residual=x)[0]
        return x

    def step(self, ids, pos, caches, arange):
        """one token for every batch row: ids int64 [bs], pos int64 [1]; returns logits [bs, vocab]."""
        if not self.v3:
            x = self.embed(ids, pos)
            if self.tiled:
                return self.head(self.step_tiled(x, pos, caches))
            if self.vfused:
                return self.head(self.step_vfused(x, pos, caches))
            if self.chained:
                return self.head(self.step_chained(x, pos, caches))
        return super().step(ids, pos, caches, arange)


def pack_model(model, bits, dev, seed=0, blocked=False, twin=True):
    """replace the 6 Linears of every block by packed QuantLinear; returns a dense twin state for the logits check.
    blocked: operators from gen_rand_ortho_butterfly (preproc_proj_extra = 0, what opt.py's --incoh_processing really selects) instead
    of the Kronecker form"""
    gen = method.gen_rand_ortho_butterfly if blocked else method.gen_rand_ortho_butterfly_noblock
    np.random.seed(seed)
    torch.manual_seed(seed)
    maxq = 2 ** bits - 1
    dense_twin = {}
    nbytes = 0
    for li, blk in enumerate(model.blocks):
        for name in ["k_proj", "v_proj", "q_proj", "out_proj", "fc1", "fc2"]:
            lin = getattr(blk, name)
            m, d = lin.weight.shape
            W = lin.weight.data                                                  # plays the role of the PROJECTED weights
            s = ops.qfnb_scale(W)
            What, codes = ops.quantize(W, 'b', s, None, maxq, want_codes=True)
            U = ops.OrthoOp(gen(m), dev)
            V = ops.OrthoOp(gen(d), dev)
            sWH = (0.5 + torch.rand(d)).to(dev)
            ql = QuantLinear(d, m, bits=bits, qfn='b').to(dev)
            ql.pack(codes, s, None, bias=lin.bias, scaleWH=sWH, U=U, V=V)
            setattr(blk, name, ql)
            nbytes += ql.qweight.numel() * 4
            if twin:
                # dense equivalent: W_dense = U^T What V diag(1/s)   (rows of What V^T... computed with the same operators)
                Wd = U.apply_cols(V.apply_rows(What.float(), transpose=True), transpose=True) / sWH[None, :]
                dense_twin[(li, name)] = (Wd.to(W.dtype), lin.bias.data.clone())
    return dense_twin, nbytes


FAST_ARGMAX = True


def time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager):
    heads, hd = model.heads, model.h // model.heads
    caches = [(torch.zeros(bs, heads, maxlen, hd, dtype=dtype, device=dev), torch.zeros(bs, heads, maxlen, hd, dtype=dtype, device=dev))
              for _ in range(model.layers_n)]
    arange = torch.arange(maxlen, device=dev)
    ids = torch.randint(0, min(50000, model.tok.weight.shape[0]), (bs,), device=dev)
    pos = torch.zeros(1, dtype=torch.int64, device=dev)
    logits_out = torch.zeros(bs, model.tok.weight.shape[0], dtype=dtype, device=dev)

    fh = bool(getattr(model, 'fused_head', False) and getattr(model, 'v3', False))
    if fh:        # the token comes out of the head launch's partials at the start of the next step; -1 = "none yet": the first step reads `ids`
        part_val = torch.full((bs, ops.HEAD_PARTS), float("-inf"), dtype=torch.float32, device=dev)
        part_idx = torch.full((bs, ops.HEAD_PARTS), -1, dtype=torch.int32, device=dev)

    def one():
        if fh:
            model.step_fused_head(ids, pos, caches, logits_out, part_val, part_idx)
            return
        lg = model.step(ids, pos, caches, arange)
        logits_out.copy_(lg)
        if FAST_ARGMAX and lg.is_cuda:
            ops.argmax_rows(lg, out=ids)                     # one 3 us launch; torch's generic reduction: 18 us for 50272 logits
        else:
            ids.copy_(lg.argmax(-1))
        pos.add_(1)

    with torch.no_grad():
        one()                                            # warm-up (allocator, attribute calls)
        pos.zero_()
        graph = None
        if not eager:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                one()
                pos.zero_()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                one()
            pos.zero_()
        run = graph.replay if graph is not None else one
        for _ in range(prompt):                          # "prompt": fills the cache token by token (untimed)
            run()
        torch.cuda.synchronize()
        lat = []
        for _ in range(tokens):
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
    if not lat:
        lat = [0.0]
    return float(np.median(lat)), float(np.mean(lat)), logits_out.float().clone()


def run(layers=24, bits=2, bs=1, prompt=128, tokens=128, eager=False, with_dense=True, only_chained=False, v3_only=False):
    """build the model, time dense fp16 (optional), packed, and fused-packed decode; returns a dict."""
    dev, dtype = torch.device("cuda:0"), torch.float16
    maxlen = prompt + tokens + 8
    torch.manual_seed(0)
    model = Decoder(layers=layers, dtype=dtype).to(dev).eval()
    for p_ in model.parameters():                         # OPT-like init scale keeps activations finite in fp16
        if p_.dim() > 1:
            p_.data.normal_(0, 0.02)
    out = {"config": {"arch": "OPT-1.3B (hidden 2048, ffn 8192, heads 32, vocab 50272)", "layers": layers, "bits": bits,
                      "bs": bs, "prompt": prompt, "tokens": tokens, "launch": "eager" if eager else "hipGraph",
                      "weights": "random init, nearest-rounded qfn-b codes, Kronecker U/V, random scaleWH"}}
    def set_attn(flag):
        for blk in model.blocks:
            blk.fused_attn = flag
    if with_dense:
        med, mean, _ = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
        out["dense_fp16"] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med}
        set_attn(True)
        med, mean, _ = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
        out["dense_fp16_fused_attn"] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med}
        set_attn(False)
    twin, nbytes = pack_model(model, bits, dev)
    del twin
    if only_chained:                                      # for kernel traces: just the best variants
        for blk in model.blocks:
            blk.fused = True
        set_attn(True)
        model.chained = True
        torch.manual_seed(7)
        med, mean, lc = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
        out["packed_w%d_chained" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med}
        if vgemm_fusable([model.blocks[0].q_proj, model.blocks[0].k_proj, model.blocks[0].v_proj], bs):
            model.vfused = True
            torch.manual_seed(7)
            med, mean, lv = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
            out["packed_w%d_vfused" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                               "logits_bit_identical_to_chained": bool(torch.equal(lc, lv))}
            model.split_handover = True
            torch.manual_seed(7)
            med, mean, lh = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
            out["packed_w%d_vfused_split_handover" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                                              "logits_bit_identical_to_vfused": bool(torch.equal(lh, lv))}
        if model.v3_ok(bs):
            model.v3 = True
            for flag, key in ((False, "packed_w%d_v3_6launch" % bits), (True, "packed_w%d_v3" % bits)):
                type(model).v3_attn = flag
                torch.manual_seed(7)
                med, mean, l3 = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
                out[key] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                            "logits_rel_diff_vs_chained": float((l3 - lc).norm() / lc.norm())}
            if fused_head_ok(model.blocks[-1].fc2, bs, model.lnf):
                model.fused_head = True                      # + csrc/decode_head.hip at both ends of the step: 5 launches per block + 2 per token
                torch.manual_seed(7)
                med, mean, lh3 = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
                out["packed_w%d_v3_head" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                                    "logits_rel_diff_vs_v3": float((lh3 - l3).norm() / l3.norm())}
                model.fused_head = False
            model.v3 = False
        if v3_only:
            return out
        model.tiled = True
        torch.manual_seed(7)
        med, mean, lt = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
        out["packed_w%d_tiled" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                          "logits_rel_diff_vs_chained": float((lt - lc).norm() / lc.norm())}
        return out
    med, mean, _ = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
    out["packed_w%d" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med, "packed_weight_MB": nbytes / 1e6,
                                "hbm_bound_tok_per_s": 8e12 / (nbytes + model.tok.weight.numel() * 2)}
    for blk in model.blocks:
        blk.fused = True
    med, mean, _ = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
    out["packed_w%d_fused" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                      "what": "q/k/v grouped; LayerNorm folded into V(x/s); bias + residual + ReLU folded into U^T y"}
    set_attn(True)
    med, mean, _ = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
    out["packed_w%d_fused_attn" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                           "what": "as packed_fused + single-launch decode attention (13 launches per block)"}
    model.chained = True
    med, mean, _ = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
    out["packed_w%d_chained" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                        "what": "as packed_fused_attn + U^T->LN->V hand-overs chained in one launch (10 launches per block)"}
    if vgemm_fusable([model.blocks[0].q_proj, model.blocks[0].k_proj, model.blocks[0].v_proj], bs):
        model.vfused = True
        med, mean, _ = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
        out["packed_w%d_vfused" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                           "what": "V-side operator in the prologue of the dequant-GEMM for the d = 2048 inputs (9 launches per block)"}
        model.split_handover = True
        med, mean, _ = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
        out["packed_w%d_vfused_split_handover" % bits] = {
            "ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
            "what": "as vfused, the n = 8192 fc1 -> fc2 hand-over as two tiled operator launches (32 workgroups each) instead of "
                    "one one-workgroup chain launch (10 launches per block)"}
    if model.v3_ok(bs):
        model.v3 = True
        med, mean, _ = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
        out["packed_w%d_v3" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                       "what": "csrc/decode_fused.hip: U^T(prev) + residual -> norm -> V -> GEMM in ONE launch per packed layer group "
                                               "(fp16 operator pass in the GEMM prologue); 5 launches per block"}
        if fused_head_ok(model.blocks[-1].fc2, bs, model.lnf):
            model.fused_head = True
            med, mean, _ = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
            out["packed_w%d_v3_head" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                                "what": "as v3 + csrc/decode_head.hip: the embedding (with the previous step's argmax) and [U_fc2^T + residual -> "
                                                        "final LayerNorm -> lm_head -> argmax partials] as one launch each: 5 launches per block + 2 per token"}
            model.fused_head = False
        model.v3 = False
    model.tiled = True
    med, mean, _ = time_decode(model, bs, prompt, tokens, maxlen, dev, dtype, eager)
    out["packed_w%d_tiled" % bits] = {"ms_per_token_median": med * 1e3, "tok_per_s": bs / med,
                                      "what": "V-op, GEMM, U-op as separate launches, every operator tiled over 8-32 workgroups (13 launches per block)"}
    del model
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--bits", type=int, default=2)
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--tokens", type=int, default=128)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--only-chained", action="store_true", help="time only the chained packed variant (for kernel traces)")
    ap.add_argument("--v3-only", action="store_true", help="with --only-chained: stop after the v3 variant")
    args = ap.parse_args()
    print(json.dumps(run(args.layers, args.bits, args.bs, args.prompt, args.tokens, args.eager,
                         with_dense=not args.only_chained, only_chained=args.only_chained, v3_only=args.v3_only)))


def decode_check(layers=2, bits=2):
    """packed model vs its dense twin (transforms folded into fp16 weights) on the same first token: relative logits error."""
    dev, dtype = torch.device("cuda:0"), torch.float16
    torch.manual_seed(0)
    model = Decoder(layers=layers, dtype=dtype).to(dev).eval()
    for p_ in model.parameters():
        if p_.dim() > 1:
            p_.data.normal_(0, 0.02)
    twin, _ = pack_model(model, bits, dev)
    torch.manual_seed(1)
    _, _, lq = time_decode(model, 2, 0, 0, 32, dev, dtype, True)
    for blk in model.blocks:
        blk.fused = True
    torch.manual_seed(1)
    _, _, lf = time_decode(model, 2, 0, 0, 32, dev, dtype, True)
    for blk in model.blocks:
        blk.fused_attn = True
    torch.manual_seed(1)
    _, _, la = time_decode(model, 2, 0, 0, 32, dev, dtype, True)
    model.chained = True
    torch.manual_seed(1)
    _, _, lc = time_decode(model, 2, 0, 3, 32, dev, dtype, True)
    model.chained = False
    torch.manual_seed(1)
    _, _, la3 = time_decode(model, 2, 0, 3, 32, dev, dtype, True)
    chain_equal = bool(torch.equal(lc, la3))
    l3 = None
    if model.v3_ok(2):
        model.v3 = True
        torch.manual_seed(1)
        _, _, l3 = time_decode(model, 2, 0, 0, 32, dev, dtype, True)
        model.v3 = False
    for blk in model.blocks:
        blk.fused = False
        blk.fused_attn = False
    for (li, name), (Wd, b) in twin.items():
        lin = nn.Linear(Wd.shape[1], Wd.shape[0], bias=True, dtype=dtype, device=dev)
        lin.weight.data, lin.bias.data = Wd, b
        setattr(model.blocks[li], name, lin)
    torch.manual_seed(1)
    _, _, ld = time_decode(model, 2, 0, 0, 32, dev, dtype, True)
    return (float((lq - ld).norm() / ld.norm()), float((lf - ld).norm() / ld.norm()), float((la - ld).norm() / ld.norm()),
            chain_equal, None if l3 is None else float((l3 - ld).norm() / ld.norm()))


if __name__ == "__main__":
    if "--check" in sys.argv:
        e1, e2, e3, ceq, e4 = decode_check()
        print(json.dumps({"decode_logits_rel_err_packed_vs_dense_twin": e1, "fused_packed_vs_dense_twin": e2,
                          "fused_packed_fused_attn_vs_dense_twin": e3, "v3_vs_dense_twin": e4,
                          "chained_logits_bit_identical_to_fused_attn_after_4_tokens": ceq}))
    else:
        main()
