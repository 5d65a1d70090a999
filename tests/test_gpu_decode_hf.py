"""-m gpu: the decode harnesses against what the reference's benchmark() actually drives -- the Hugging Face model with
past_key_values (opt.py:431-482, llama.py:418-471: `model(input_ids[:, i:i+1], past_key_values=..., use_cache=True)` token by
token).  scripts/decode_opt.py / decode_llama.py are hand-rolled blocks; their other test compares them with their own dense
twins, which cannot catch an architecture mistake shared by both sides.  Here:

  1. an HF OPTForCausalLM / LlamaForCausalLM (random init, fp16, hidden 2048 so that the fused launch paths are the ones that
     run) receives EXACTLY the weights the harness computes with: embeddings, norms, and for every decoder Linear the dense
     equivalent  U^T What V diag(1/s)  (+ bias) of the packed layer;
  2. HF generates 16 tokens greedily with use_cache=True, one token per call, like benchmark();
  3. the harness is fed the same tokens, one hipGraph replay per token (device-resident position, static KV cache): dense
     (fp16 Linears holding the same twin weights), packed (QuantLinear.forward per layer) and the fused decode variants.

Gates: logits within 1e-2 of HF's (relative l2, every token); HF's greedy token is the harness's greedy token wherever HF's
top-1 margin exceeds the logits difference (a random-init model has near-ties no two fp16 implementations agree on), and the
free-running greedy continuations are identical as long as every step had such a margin."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NTOK = 16


def _load(name):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def hf_opt_from_decoder(dec, twin, heads, maxpos, dtype, dev):
    """an HF OPTForCausalLM holding the harness's tensors: tok / pos embeddings, LayerNorms, final norm, and per Linear either
    the twin's dense equivalent (packed harness) or the harness's own weight"""
    from transformers import OPTConfig, OPTForCausalLM
    h, ffn, vocab = dec.h, dec.blocks[0].fc1.out_features if hasattr(dec.blocks[0].fc1, "out_features") else dec.blocks[0].fc1.outfeatures, dec.tok.weight.shape[0]
    cfg = OPTConfig(hidden_size=h, ffn_dim=ffn, num_hidden_layers=len(dec.blocks), num_attention_heads=heads, word_embed_proj_dim=h,
                    vocab_size=vocab, max_position_embeddings=maxpos, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                    layerdrop=0.0, do_layer_norm_before=True)
    hf = OPTForCausalLM(cfg).to(dtype).to(dev).eval()
    d = hf.model.decoder
    with torch.no_grad():
        d.embed_tokens.weight.copy_(dec.tok.weight)
        d.embed_positions.weight.copy_(dec.posemb.weight)                      # OPTLearnedPositionalEmbedding: maxpos + 2 rows, offset 2
        d.final_layer_norm.weight.copy_(dec.lnf.weight)
        d.final_layer_norm.bias.copy_(dec.lnf.bias)
        for li, (blk, lay) in enumerate(zip(dec.blocks, d.layers)):
            lay.self_attn_layer_norm.load_state_dict(blk.ln1.state_dict())
            lay.final_layer_norm.load_state_dict(blk.ln2.state_dict())
            for name, tgt in (("q_proj", lay.self_attn.q_proj), ("k_proj", lay.self_attn.k_proj), ("v_proj", lay.self_attn.v_proj),
                              ("out_proj", lay.self_attn.out_proj), ("fc1", lay.fc1), ("fc2", lay.fc2)):
                W, b = twin[(li, name)]
                tgt.weight.copy_(W)
                tgt.bias.copy_(b)
    assert hf.lm_head.weight.data_ptr() == d.embed_tokens.weight.data_ptr()    # tied, like the harness's F.linear(.., tok.weight)
    return hf


def hf_llama_from_decoder(dec, twin, heads, maxpos, eps, dtype, dev):
    from transformers import LlamaConfig, LlamaForCausalLM
    ffn = next(W.shape[0] for (li, n), W in twin.items() if n == "gate_proj")
    cfg = LlamaConfig(hidden_size=dec.h, intermediate_size=ffn, num_hidden_layers=len(dec.blocks), num_attention_heads=heads,
                      num_key_value_heads=heads, vocab_size=dec.tok.weight.shape[0], max_position_embeddings=maxpos, rms_norm_eps=eps,
                      rope_theta=10000.0, attention_dropout=0.0, tie_word_embeddings=False, attention_bias=False, mlp_bias=False)
    hf = LlamaForCausalLM(cfg).to(dtype).to(dev).eval()
    with torch.no_grad():
        hf.model.embed_tokens.weight.copy_(dec.tok.weight)
        hf.model.norm.weight.copy_(dec.norm.weight)
        hf.lm_head.weight.copy_(dec.lm_head.weight)
        for li, (blk, lay) in enumerate(zip(dec.blocks, hf.model.layers)):
            lay.input_layernorm.weight.copy_(blk.n1.weight)
            lay.post_attention_layernorm.weight.copy_(blk.n2.weight)
            for name, tgt in (("q_proj", lay.self_attn.q_proj), ("k_proj", lay.self_attn.k_proj), ("v_proj", lay.self_attn.v_proj),
                              ("o_proj", lay.self_attn.o_proj), ("gate_proj", lay.mlp.gate_proj), ("up_proj", lay.mlp.up_proj),
                              ("down_proj", lay.mlp.down_proj)):
                tgt.weight.copy_(twin[(li, name)])
    return hf


@torch.no_grad()
def hf_generate(hf, first, ntok):
    """benchmark()'s loop (opt.py:463-480): one token per call with the cache handed back; returns (tokens fed [ntok], logits [ntok, vocab])"""
    ids = torch.tensor([[first]], device=DEV)
    pkv, toks, logits = None, [], []
    for _ in range(ntok):
        out = hf(input_ids=ids, past_key_values=pkv, use_cache=True)
        pkv = out.past_key_values
        lg = out.logits[0, -1].float()
        toks.append(int(ids[0, 0]))
        logits.append(lg.clone())
        ids = lg.argmax().reshape(1, 1)
    return toks, torch.stack(logits)


@torch.no_grad()
def harness_logits(dec, toks, maxlen, dtype, graph=True):
    """feed `toks` one per step through Decoder.step under ONE captured hipGraph replayed per token (eager when graph=False)"""
    heads, hd = dec.heads, dec.h // dec.heads
    caches = [(torch.zeros(1, heads, maxlen, hd, dtype=dtype, device=DEV), torch.zeros(1, heads, maxlen, hd, dtype=dtype, device=DEV))
              for _ in range(dec.layers_n)]
    arange = torch.arange(maxlen, device=DEV)
    ids = torch.zeros(1, dtype=torch.int64, device=DEV)
    pos = torch.zeros(1, dtype=torch.int64, device=DEV)
    vocab = dec.tok.weight.shape[0]
    out = torch.zeros(1, vocab, dtype=torch.float32, device=DEV)

    fh = bool(getattr(dec, 'fused_head', False) and dec.v3)    # csrc/decode_head.hip at both ends of the step (teacher-forced: no argmax partials)
    out16 = torch.zeros(1, vocab, dtype=torch.float16, device=DEV)

    def one():
        if fh:
            dec.step_fused_head(ids, pos, caches, out16, None, None)      # increments pos itself
            out.copy_(out16)
            return
        out.copy_(dec.step(ids, pos, caches, arange))
        pos.add_(1)
    ids.fill_(toks[0])
    one()                                                    # warm-up; the caches are rewritten from position 0 below
    pos.zero_()
    run = one
    if graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            one()
            pos.zero_()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            one()
        pos.zero_()
        run = g.replay
    logits = []
    for t in toks:
        ids.fill_(t)
        run()
        logits.append(out[0].clone())
    if graph:
        torch.cuda.synchronize()
    return torch.stack(logits)


def _gate(name, got, ref, toks):
    rel = ((got - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
    assert rel <= 1e-2, (name, rel)
    top2 = ref.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    diff = (got - ref).abs().max(dim=1).values
    decisive = margin > 2 * diff
    assert bool((got.argmax(1)[decisive] == ref.argmax(1)[decisive]).all()), name
    # free-running greedy: toks[i + 1] is HF's argmax at step i; the harness picks the same token at every decisive step
    agree = int((got.argmax(1) == ref.argmax(1)).sum())
    return rel, int(decisive.sum()), agree


def test_opt_decode_harness_matches_hf_with_past_key_values():
    D = _load("decode_opt")
    dtype, maxpos = torch.float16, 64
    torch.manual_seed(0)
    dec = D.Decoder(layers=2, h=2048, ffn=8192, heads=32, vocab=512, maxpos=maxpos, dtype=dtype).to(DEV).eval()
    for p_ in dec.parameters():
        if p_.dim() > 1:
            p_.data.normal_(0, 0.02)
        else:
            p_.data.add_(0.05 * torch.randn_like(p_))       # non-trivial LayerNorm gains / biases, Linear biases
    twin, _ = D.pack_model(dec, 2, DEV)                      # the 12 Linears are packed w2 QuantLinear layers now
    hf = hf_opt_from_decoder(dec, twin, 32, maxpos, dtype, DEV)
    toks, ref = hf_generate(hf, 7, NTOK)
    report = {}
    # packed, layer by layer (QuantLinear.forward, eager torch attention): the plain path
    report["packed"] = _gate("packed", harness_logits(dec, toks, maxpos, dtype), ref, toks)
    for blk in dec.blocks:
        blk.fused, blk.fused_attn = True, True
    report["fused"] = _gate("fused", harness_logits(dec, toks, maxpos, dtype), ref, toks)
    dec.chained = True
    report["chained"] = _gate("chained", harness_logits(dec, toks, maxpos, dtype), ref, toks)
    if D.vgemm_fusable([dec.blocks[0].q_proj, dec.blocks[0].k_proj, dec.blocks[0].v_proj], 1):
        dec.vfused, dec.split_handover = True, True
        report["vfused_split_handover"] = _gate("vfused", harness_logits(dec, toks, maxpos, dtype), ref, toks)
    assert dec.v3_ok(1)
    dec.v3 = True                                            # csrc/decode_fused.hip: 5 launches per block
    report["v3"] = _gate("v3", harness_logits(dec, toks, maxpos, dtype), ref, toks)
    dec.fused_head = True                                    # + embedding and [U^T + residual -> final norm -> lm_head] as one launch each
    report["v3_head"] = _gate("v3_head", harness_logits(dec, toks, maxpos, dtype), ref, toks)
    dec.fused_head = False
    dec.v3 = False
    # and the dense harness (fp16 Linears with the twin weights): the architecture alone, no packed kernels
    dec.chained = dec.vfused = dec.split_handover = dec.tiled = False
    for li, blk in enumerate(dec.blocks):
        blk.fused = blk.fused_attn = False
        for name in ["q_proj", "k_proj", "v_proj", "out_proj", "fc1", "fc2"]:
            W, b = twin[(li, name)]
            lin = nn.Linear(W.shape[1], W.shape[0], bias=True, dtype=dtype, device=DEV)
            lin.weight.data, lin.bias.data = W, b
            setattr(blk, name, lin)
    report["dense"] = _gate("dense", harness_logits(dec, toks, maxpos, dtype), ref, toks)
    print("opt decode vs HF (max rel logits err, decisive steps, greedy agreement of %d):" % NTOK, report)


def test_llama_decode_harness_matches_hf_with_past_key_values():
    L = _load("decode_llama")
    dtype, maxpos, eps = torch.float16, 64, 1e-5
    torch.manual_seed(0)
    dec = L.Decoder(layers=2, h=2048, ffn=11008, heads=16, vocab=512, maxpos=maxpos, eps=eps, dtype=dtype).to(DEV).eval()
    for p_ in dec.parameters():
        if p_.dim() > 1:
            p_.data.normal_(0, 0.02)
        else:
            p_.data.add_(0.05 * torch.randn_like(p_))
    twin, _ = L.pack_model(dec, 2, DEV)
    hf = hf_llama_from_decoder(dec, twin, 16, maxpos, eps, dtype, DEV)
    toks, ref = hf_generate(hf, 7, NTOK)
    report = {}
    report["packed"] = _gate("packed", harness_logits(dec, toks, maxpos, dtype), ref, toks)
    for blk in dec.blocks:
        blk.fused = True
    report["fused"] = _gate("fused", harness_logits(dec, toks, maxpos, dtype), ref, toks)
    assert dec.v3_ok(1)
    dec.v3 = True                                            # fused launches (64 x 32 operators at hidden 2048) + rotary in the attention prologue
    report["v3"] = _gate("v3", harness_logits(dec, toks, maxpos, dtype), ref, toks)
    dec.fused_head = True                                    # + embedding and [U^T + residual -> final norm -> lm_head] as one launch each
    report["v3_head"] = _gate("v3_head", harness_logits(dec, toks, maxpos, dtype), ref, toks)
    dec.fused_head = False
    dec.v3 = False
    for li, blk in enumerate(dec.blocks):
        blk.fused = False
        for name in L.Decoder.NAMES:
            W = twin[(li, name)]
            lin = nn.Linear(W.shape[1], W.shape[0], bias=False, dtype=dtype, device=DEV)
            lin.weight.data = W
            setattr(blk, name, lin)
    report["dense"] = _gate("dense", harness_logits(dec, toks, maxpos, dtype), ref, toks)
    print("llama decode vs HF (max rel logits err, decisive steps, greedy agreement of %d):" % NTOK, report)


@pytest.mark.parametrize("bs", [3, 4, 8, 16])
def test_opt_engine_many_sequences_matches_hf(bs):
    """round 5: 8 and 16 sequences per step stay on the fused launch family -- engine mode v3, every layer group as
    [prologue-only launch, one workgroup per row] + [dequant-GEMM on the decode-order codes] -- and every sequence's logits are HF's
    (past_key_values, one token per call) within 1e-2, greedy tokens equal wherever HF's margin is decisive.
    3 and 4 sequences: the same two-launch groups (quant.two_launch_from: hidden 2048 takes them from 3 rows on) under the fused
    embedding / head launches of mode v3_head."""
    from quip_amd import decode
    D = _load("decode_opt")
    dtype, maxpos = torch.float16, 64
    torch.manual_seed(0)
    dec = D.Decoder(layers=2, h=2048, ffn=8192, heads=32, vocab=512, maxpos=maxpos, dtype=dtype).to(DEV).eval()
    for p_ in dec.parameters():
        if p_.dim() > 1:
            p_.data.normal_(0, 0.02)
        else:
            p_.data.add_(0.05 * torch.randn_like(p_))
    twin, _ = D.pack_model(dec, 2, DEV)
    hf = hf_opt_from_decoder(dec, twin, 32, maxpos, dtype, DEV)
    seqs = [hf_generate(hf, 7 + 31 * s, NTOK) for s in range(bs)]           # (tokens fed, HF logits) per sequence
    eng = decode.DecodeEngine(dec, bs=bs, max_len=maxpos)
    assert eng.mode == ("v3" if bs > 4 else "v3_head"), eng.mode
    got = []
    for i in range(NTOK):
        ids = torch.tensor([seqs[s][0][i] for s in range(bs)], device=DEV)
        got.append(eng.forward(ids).float().clone())
    got = torch.stack(got)                                                   # [NTOK, bs, vocab]
    worst = 0.0
    for s in range(bs):
        rel, decisive, agree = _gate(f"opt-bs{bs}-seq{s}", got[:, s], seqs[s][1], seqs[s][0])
        worst = max(worst, rel)
    # the same engine one sequence at a time (mode v3_head, the single fused launches) sees the same logits up to fp16 hand-over rounding
    e1 = decode.DecodeEngine(dec, bs=1, max_len=maxpos)
    one = torch.stack([e1.forward(t)[0].float().clone() for t in seqs[0][0]])
    assert float((one - got[:, 0]).norm() / one.norm()) <= 5e-3
    print(f"opt engine, {bs} sequences per step vs HF: worst max-rel logits err {worst:.2e}")


@pytest.mark.parametrize("bs", [6, 16])
def test_llama_engine_many_sequences_matches_hf(bs):
    """the Llama block at 6 and 16 sequences per step: mode v3 -- fused-stage pairs for q / k / v, o, gate / up, and the 11008-wide MLP tail
    on csrc/decode_bigp.hip's multi-row launches (bigp_u: row groups of 4; bigp_v_gemm: ONE weight pass for all rows) -- against HF with
    past_key_values, every sequence within 1e-2."""
    from quip_amd import decode
    L = _load("decode_llama")
    dtype, maxpos, eps = torch.float16, 64, 1e-5
    torch.manual_seed(0)
    dec = L.Decoder(layers=2, h=2048, ffn=11008, heads=16, vocab=512, maxpos=maxpos, eps=eps, dtype=dtype).to(DEV).eval()
    for p_ in dec.parameters():
        if p_.dim() > 1:
            p_.data.normal_(0, 0.02)
        else:
            p_.data.add_(0.05 * torch.randn_like(p_))
    twin, _ = L.pack_model(dec, 2, DEV)
    hf = hf_llama_from_decoder(dec, twin, 16, maxpos, eps, dtype, DEV)
    seqs = [hf_generate(hf, 7 + 31 * s, NTOK) for s in range(bs)]
    eng = decode.DecodeEngine(dec, bs=bs, max_len=maxpos)
    assert eng.mode == "v3", eng.mode
    got = []
    for i in range(NTOK):
        ids = torch.tensor([seqs[s][0][i] for s in range(bs)], device=DEV)
        got.append(eng.forward(ids).float().clone())
    got = torch.stack(got)
    worst = max(_gate(f"llama-bs{bs}-seq{s}", got[:, s], seqs[s][1], seqs[s][0])[0] for s in range(bs))
    print(f"llama engine, {bs} sequences per step vs HF: worst max-rel logits err {worst:.2e}")
