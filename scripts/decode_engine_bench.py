#!/usr/bin/env python3
"""quip_amd.decode.DecodeEngine at full model size, for the packed configurations the fused launches do NOT cover -- the measured statement
of what such a model decodes at (VERDICT r3 next #2b):

    --blocked        operators from gen_rand_ortho_butterfly (preproc_proj_extra = 0: what the reference's --incoh_processing really
                     selects, opt.py:596 sets an unused `proj_extra`): factor storage n (p + q) values per side instead of p^2 + q^2, a
                     workgroup cannot redo the pass in its prologue -> engine mode "fused" (operator / grouped GEMM / operator launches;
                     the blocked operators on the general two-stage K3 kernel)
    --bits 4         w4 qfn b (csrc/decode_fused.hip is 2-bit): mode "fused" as well
    (neither)        the Kronecker w2 model: mode v3_head, the number bench.py reports

    python scripts/decode_engine_bench.py --arch opt|llama [--blocked] [--bits 4] [--layers N] [--tokens 64] [--prompt 64]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from quip_amd import decode  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="opt", choices=["opt", "llama"])
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--bits", type=int, default=2)
    ap.add_argument("--blocked", action="store_true")
    ap.add_argument("--prompt", type=int, default=64)
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--mode", default="auto")
    ap.add_argument("--bs", type=int, default=1, help="sequences decoded side by side (1..4 on the fused launches); tok/s is the aggregate")
    ap.add_argument("--blk-fused-n", type=int, default=-1, help="csrc/ortho_blk.hip: one launch per blocked operator up to this n (0: never)")
    ap.add_argument("--two-launch-rows", default="", help="comma list: repeat every measurement with quant.TWO_LAUNCH_ROWS set to each value")
    ap.add_argument("--sweep", default="", help="'bs:fused_n,bs:fused_n,...' -- the model is built once, one JSON line per entry")
    a = ap.parse_args()
    if a.sweep:
        model = None
        for item in a.sweep.split(","):
            b, f = item.split(":")
            a.bs, a.blk_fused_n = int(b), int(f)
            for tl in ([int(v) for v in a.two_launch_rows.split(",")] if a.two_launch_rows else [None]):
                if tl is not None:
                    from quip_amd import quant
                    quant.TWO_LAUNCH_ROWS = tl
                try:
                    res, model = run(a, model=model, keep=True)
                except Exception as e:                                    # a batch size a mode does not take: say so, go on
                    res = {"bs": a.bs, "blk_fused_n": a.blk_fused_n, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                    if model is None:
                        raise
                if tl is not None:
                    res["two_launch_rows"] = tl
                print(json.dumps(res), flush=True)
        return
    print(json.dumps(run(a)))


def run(a, model=None, keep=False):
    """a: namespace with arch, layers, bits, blocked, prompt, tokens, mode (bench.py builds one for its `decode_blocked` leg)"""
    if model is None:
        model = build(a)
    out = measure(a, *model)
    if keep:
        return out, model
    del model
    torch.cuda.empty_cache()
    return out


def build(a):
    dev, dtype = torch.device("cuda:0"), torch.float16
    torch.manual_seed(0)
    if a.arch == "opt":
        import decode_opt as D
        model = D.Decoder(layers=a.layers or 24, dtype=dtype).to(dev).eval()
        arch = "OPT-1.3B (hidden 2048, ffn 8192, heads 32, vocab 50272)"
    else:
        import decode_llama as D
        model = D.Decoder(layers=a.layers or 32, dtype=dtype).to(dev).eval()
        arch = "Llama-2-7B (hidden 4096, intermediate 11008, heads 32 x 128, vocab 32000)"
    for p_ in model.parameters():
        if p_.dim() > 1:
            p_.data.normal_(0, 0.02)
    if a.arch == "opt":
        _, nbytes = D.pack_model(model, a.bits, dev, blocked=a.blocked, twin=False)
    else:
        if a.blocked:
            from quip_amd import method
            keep = method.gen_rand_ortho_butterfly_noblock
            method.gen_rand_ortho_butterfly_noblock = method.gen_rand_ortho_butterfly     # decode_llama.pack_model draws through this name
            try:
                _, nbytes = D.pack_model(model, a.bits, dev, twin=False)
            finally:
                method.gen_rand_ortho_butterfly_noblock = keep
        else:
            _, nbytes = D.pack_model(model, a.bits, dev, twin=False)
    return model, nbytes, arch


def measure(a, model, nbytes, arch):
    dev = torch.device("cuda:0")
    bs = getattr(a, "bs", 1)
    if getattr(a, "blk_fused_n", -1) >= 0:
        from quip_amd import ops
        ops.ortho_blocked_config(a.blk_fused_n)
    maxlen = a.prompt + a.tokens + 8
    eng = decode.DecodeEngine(model, bs=bs, max_len=maxlen, mode=a.mode)
    ids = torch.randint(0, 30000, (bs, a.prompt + a.tokens), device=dev)
    res = eng.benchmark(ids)
    lat = res["times"][a.prompt:]
    med = float(np.median(lat))
    head = model.head_weight.numel() * 2
    fact = 0
    for blk in model.blocks:
        for m in blk.modules():
            if isinstance(m, decode.QuantLinear):
                for op in (m.U, m.V):
                    fact += (op._B0.numel() + op._B1.numel()) * 2      # what a decode launch would read as fp16
    out = {"arch": arch, "layers": len(model.blocks), "bits": a.bits, "operators": "blocked butterfly (extra 0)" if a.blocked else "Kronecker (extra 1)",
           "engine_mode": eng.mode, "bs": bs, "blk_fused_n": getattr(a, "blk_fused_n", -1), "prompt": a.prompt, "tokens": a.tokens,
           "ms_per_step_median": med * 1e3, "ms_per_token_median": med * 1e3 / bs, "tok_per_s": bs / med,
           "packed_weight_MB": nbytes / 1e6, "operator_factor_MB_fp16": fact / 1e6,
           "hbm_bound_tok_per_s": bs * 8e12 / (nbytes + head + fact), "frac_of_byte_bound": (1.0 / med) / (8e12 / (nbytes + head + fact))}
    del eng
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
