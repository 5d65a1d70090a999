#!/usr/bin/env python3
"""End-to-end use of the quip_amd surface on a Hugging Face OPT model, the way the reference's driver does it
(opt.py:29-190 `opt_sequential`: catch the inputs of decoder layer 0, then per layer: hook every Linear with
QuantMethod.add_batch, run the calibration samples, post_batch -> preproc -> fasterquant -> free, re-run the layer with the
quantised weights to get the next layer's inputs).  Not a rebuild of opt.py -- no datasets, no eval harness: a random-init
OPT-shaped model (transformers' OPTForCausalLM, no checkpoint is reachable offline) and random calibration tokens, to show
that the reference's call sequence runs unchanged against `quip_amd.*` on the GPU, and how the quantised layers become
packed `QuantLinear`s.

    python scripts/quantize_opt.py [--hidden 768 --ffn 3072 --heads 12 --layers 2 --nsamples 8 --seqlen 128]
                                   [--wbits 2 --quant ldlq|gptq|nearest|ldlqRG --npasses 0 --incoh --pack]

Prints one JSON line: per-layer proxy errors, wall time, and the relative change of the model's logits."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import bal, gptq, near, quant  # noqa: E402
from quip_amd.modelutils import find_layers  # noqa: E402


def build_model(args, dev):
    if args.arch == "llama":                                       # llama.py:87-156 drives the same sequence over model.model.layers
        from transformers import LlamaConfig, LlamaForCausalLM
        cfg = LlamaConfig(hidden_size=args.hidden, intermediate_size=args.ffn, num_hidden_layers=args.layers,
                          num_attention_heads=args.heads, num_key_value_heads=args.heads, vocab_size=args.vocab,
                          max_position_embeddings=args.seqlen)
        torch.manual_seed(0)
        model = LlamaForCausalLM(cfg).half().to(dev).eval()
        model.seqlen = args.seqlen
        return model
    from transformers import OPTConfig, OPTForCausalLM
    cfg = OPTConfig(hidden_size=args.hidden, ffn_dim=args.ffn, num_hidden_layers=args.layers, num_attention_heads=args.heads,
                    word_embed_proj_dim=args.hidden, vocab_size=args.vocab, max_position_embeddings=args.seqlen)
    torch.manual_seed(0)
    model = OPTForCausalLM(cfg).half().to(dev).eval()
    model.seqlen = args.seqlen
    return model


@torch.no_grad()
def opt_sequential(model, batches, dev, args):
    """opt.py:29-190 with the quantisation classes of quip_amd; everything stays on the GPU."""
    model.config.use_cache = False
    layers = model.model.layers if args.arch == 'llama' else model.model.decoder.layers
    prefix = 'model.layers' if args.arch == 'llama' else 'model.decoder.layers'
    dtype = next(iter(model.parameters())).dtype
    inps = torch.zeros((args.nsamples, model.seqlen, model.config.hidden_size), dtype=dtype, device=dev)
    cache = {'i': 0, 'kwargs': None}

    class Catcher(nn.Module):                                     # opt.py:57-68
        def __init__(self, module):
            super().__init__()
            self.module = module

        def forward(self, inp, **kwargs):
            inps[cache['i']] = inp
            cache['i'] += 1
            cache['kwargs'] = kwargs
            raise ValueError

    layers[0] = Catcher(layers[0])
    for batch in batches:
        try:
            model(batch.to(dev))
        except ValueError:
            pass
    layers[0] = layers[0].module
    kwargs = {k: v for k, v in cache['kwargs'].items() if 'past' not in k and 'cache' not in k}
    outs = torch.zeros_like(inps)

    def run_layer(layer, j):
        out = layer(inps[j].unsqueeze(0), **kwargs)
        return out[0] if isinstance(out, (tuple, list)) else out

    report, packed = [], {}
    for i, layer in enumerate(layers):
        subset = find_layers(layer)                               # opt.py:97
        methods = {}
        for name, lin in subset.items():                          # opt.py:99-129
            if args.quant == 'gptq':
                m = gptq.GPTQ(lin)
            elif args.quant == 'nearest':
                m = near.Nearest(lin)
            else:
                m = bal.Balance(lin)
                m.configure(args.quant, args.wbits, args.npasses, unbiased=False)
            m.quantizer = quant.Quantizer()
            m.quantizer.configure(args.wbits, perchannel=True, sym=False, qfn=args.qfn, mse=False)
            methods[name] = m
        handles = [subset[name].register_forward_hook(lambda _, inp, out, name=name: methods[name].add_batch(inp[0].data, out.data))
                   for name in subset]                            # opt.py:131-140
        for j in range(args.nsamples):
            outs[j] = run_layer(layer, j)                         # opt.py:141-143
        for h in handles:
            h.remove()
        for name, m in methods.items():                           # opt.py:147-170
            t0 = time.perf_counter()
            m.post_batch()
            m.preproc(preproc_gptqH=True, percdamp=args.percdamp, preproc_rescale=args.incoh, preproc_proj=args.incoh,
                      preproc_proj_extra=1 if args.pack else 0)
            if args.quant == 'gptq':
                m.fasterquant(groupsize=getattr(args, 'groupsize', -1))
            elif args.quant == 'nearest':
                m.fasterquant()
            else:
                m.fasterquant(lazy_batch=False)
            torch.cuda.synchronize()
            report.append({"layer": i, "name": name, "error": float(m.error), "Hmag": float(m.Hmag),
                           "seconds": round(time.perf_counter() - t0, 4)})
            if args.pack and hasattr(m, 'codes') and args.quant != 'gptq':
                packed[f"{prefix}.{i}.{name}"] = quant.QuantLinear.from_method(m, subset[name])
            m.free()
        for j in range(args.nsamples):                            # opt.py:172-174: next layer sees the quantised block
            outs[j] = run_layer(layer, j)
        inps, outs = outs, inps
    return report, packed


def llama_sequential(model, batches, dev, args):
    """llama.py:36-171 on quip_amd: the same block-sequential loop over model.model.layers.  Carries the fixes the reference's
    Llama driver needs before any method runs under transformers 5: `args` is a parameter, not a module
    global; Balance.configure gets its four arguments and fasterquant its lazy_batch; and EVERY keyword the first decoder
    layer was called with is captured and passed back -- in particular position_embeddings = (cos, sin), without which
    LlamaDecoderLayer cannot run (the reference forwards only attention_mask and position_ids, llama.py:58-63,134,160)."""
    args.arch = 'llama'
    return opt_sequential(model, batches, dev, args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="opt", choices=["opt", "llama"])
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--ffn", type=int, default=3072)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--vocab", type=int, default=4096)
    ap.add_argument("--nsamples", type=int, default=8)
    ap.add_argument("--seqlen", type=int, default=128)
    ap.add_argument("--wbits", type=int, default=2)
    ap.add_argument("--quant", default="ldlq", choices=["ldlq", "ldlqRG", "gptq", "nearest"])
    ap.add_argument("--npasses", type=int, default=0)
    ap.add_argument("--qfn", default=None)
    ap.add_argument("--percdamp", type=float, default=0.01)
    ap.add_argument("--groupsize", type=int, default=-1, help="GPTQ group size (gptq.py:69-76): 16 / 32 / 64 / 128 run in the kernel")
    ap.add_argument("--incoh", action="store_true", help="--incoh_processing: rescale + random orthogonal projection")
    ap.add_argument("--pack", action="store_true", help="swap the quantised Linears for packed QuantLinear layers afterwards")
    args = ap.parse_args()
    if args.qfn is None:
        args.qfn = 'b' if (args.incoh and args.quant != 'gptq') else 'a'      # opt.py:560-563
    dev = torch.device("cuda:0")
    np.random.seed(0)
    model = build_model(args, dev)
    g = torch.Generator().manual_seed(1)
    batches = [torch.randint(0, args.vocab, (1, args.seqlen), generator=g) for _ in range(args.nsamples)]
    probe = torch.randint(0, args.vocab, (2, args.seqlen), generator=g).to(dev)
    with torch.no_grad():
        ref_logits = model(probe).logits.float()
    t0 = time.perf_counter()
    report, packed = opt_sequential(model, batches, dev, args)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    with torch.no_grad():
        q_logits = model(probe).logits.float()
    out = {"config": vars(args), "wall_s": round(wall, 3), "linears": len(report),
           "logits_rel_change_fake_quant": float((q_logits - ref_logits).norm() / ref_logits.norm()),
           "mean_proxy_error": float(np.mean([r["error"] for r in report])), "per_linear": report}
    if packed:
        quant.make_quant(model, packed)
        with torch.no_grad():
            p_logits = model(probe).logits.float()
        out["packed_layers"] = len(packed)
        out["logits_rel_diff_packed_vs_fake_quant"] = float((p_logits - q_logits).norm() / q_logits.norm())
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
