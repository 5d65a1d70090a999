"""-m gpu: the two halves of the path meet.  A Hugging Face OPT / Llama model is quantised by the driver's call sequence
(the reference's own opt.py:29-190 / llama.py:36-171 when oracle/stage_ref.py staged them, scripts/quantize_opt.py otherwise)
with LDLQ, 2 bits, incoherence processing -- `pre_proj_extra` 0 (the blocked butterfly `--incoh_processing` really selects,
opt.py:596) and 1 (the Kronecker operator the north_star names) -- the packed layers are taken out of that run
(`decode.collect_packed`), swapped into the model and decoded by `decode.DecodeEngine.from_hf`.

Reference side: the SAME model object as the driver left it -- dense fp16 weights W = U^T What V diag(1/s) written back by
`postproc` (what the reference evaluates, opt.py:193-299, and what its benchmark() would time) -- run by Hugging Face with
`past_key_values`, one token per call (opt.py:463-480).

Gates: engine logits within 1e-2 of HF's on every one of 16 tokens (relative l2; measured 2-6e-3: fp16 roundings of the
written-back dense weights + the 16-bit activations of the packed kernels); greedy token equal wherever HF's top-1 margin
exceeds twice the logits difference; free-running `generate` identical to HF's greedy continuation when every step is decisive;
`benchmark()` reproduces the perplexity of the teacher-forced run.  Hidden 2048 / ffn 8192 (OPT) and 2048 / 11008 (Llama) so that
with pre_proj_extra = 1 the engine MUST pick the fused launches (v3_head), which the test asserts."""
import copy
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "scripts"))
NTOK, VOCAB, SEQLEN, NSAMPLES = 16, 512, 64, 8
STAGED = os.path.exists(os.path.join(os.path.dirname(HERE), "oracle", "_ref", "opt.py"))


def _init(model, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):
            r = torch.randn(p.shape, generator=g, device=DEV, dtype=torch.float32)
            if "norm" in name:
                v = (1.0 + 0.1 * r) if name.endswith("weight") else 0.05 * r
            elif name.endswith("bias"):
                v = 0.02 * r
            else:
                v = 0.02 * r
            p.copy_(v.to(p.dtype))


def build(arch):
    if arch == "opt":
        from transformers import OPTConfig, OPTForCausalLM
        cfg = OPTConfig(hidden_size=2048, ffn_dim=8192, num_hidden_layers=2, num_attention_heads=32, word_embed_proj_dim=2048,
                        vocab_size=VOCAB, max_position_embeddings=SEQLEN, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                        layerdrop=0.0)
        model = OPTForCausalLM(cfg)
    else:
        from transformers import LlamaConfig, LlamaForCausalLM
        cfg = LlamaConfig(hidden_size=2048, intermediate_size=11008, num_hidden_layers=2, num_attention_heads=16, num_key_value_heads=16,
                          vocab_size=VOCAB, max_position_embeddings=SEQLEN, rms_norm_eps=1e-5, attention_dropout=0.0,
                          tie_word_embeddings=False)
        model = LlamaForCausalLM(cfg)
    model = model.half().to(DEV).eval()
    _init(model, 11 if arch == "opt" else 12)
    model.seqlen = SEQLEN
    return model


def quantise(model, arch, extra):
    """LDLQ w2 + incoherence processing through the driver, packed layers collected on the way; returns (collector, which driver ran)"""
    import run_reference_driver as R
    from quip_amd import decode
    drv, is_ref = (R.load_driver() if arch == "opt" else R.load_llama_driver())
    assert is_ref == STAGED
    rs = np.random.RandomState(5)
    batches = [(torch.from_numpy(rs.randint(0, VOCAB, size=(1, SEQLEN))).long(), None) for _ in range(NSAMPLES)]
    args = types.SimpleNamespace(nsamples=NSAMPLES, quant="ldlq", wbits=2, qbits=2, qfn="b", npasses=0, unbiased=False, lazy_batch=False,
                                 percdamp=0.01, pre_gptqH=True, pre_rescale=True, pre_proj=True, pre_proj_extra=extra, groupsize=-1)
    np.random.seed(0)
    torch.manual_seed(0)
    with decode.collect_packed() as packed:
        drv(model, batches, torch.device(DEV), args)
    model.to(DEV)                                            # the reference's drivers park finished blocks on the CPU
    return packed, ("reference" if is_ref else "restatement")


@pytest.mark.parametrize("arch,extra", [("opt", 1), ("opt", 0), ("llama", 1), ("llama", 0)])
def test_ldlq_pack_decode_matches_the_fake_quant_model_under_hf(arch, extra):
    from quip_amd import decode, quant
    from test_gpu_decode_hf import hf_generate, _gate
    model = build(arch)
    packed, which = quantise(model, arch, extra)
    n_lin = 12 if arch == "opt" else 14
    assert len(packed.layers) == n_lin
    hf = copy.deepcopy(model)                                # the dense fake-quant model the reference evaluates
    toks, ref = hf_generate(hf, 7, NTOK)
    named = packed.install(model)
    assert len(named) == n_lin and all(isinstance(m, quant.QuantLinear) for m in named.values())
    blocked = {bool(q.V.blocked) for q in named.values()} | {bool(q.U.blocked) for q in named.values()}
    assert blocked == {extra == 0}                           # extra 0: blocked butterfly on both sides of every layer
    eng = decode.DecodeEngine.from_hf(model, max_len=SEQLEN)
    assert eng.mode == ("v3_head" if extra == 1 else "fused"), eng.mode
    got = torch.stack([eng.forward(t)[0].float().clone() for t in toks])
    rel, decisive, agree = _gate(f"{arch}-extra{extra}", got, ref, toks)
    report = {"driver": which, "mode": eng.mode, "max_rel_logits_err": rel, "decisive": decisive, "agree": agree}
    # free-running greedy continuation vs HF's own (toks[1:] are HF's argmaxes): identical when every step was decisive
    eng.reset()
    gen = eng.generate(toks[0], NTOK - 1)[:, 0].tolist()
    if decisive == NTOK:
        assert gen == toks[1:], (gen, toks[1:])
    report["greedy_equal_prefix"] = next((i for i, (a, b) in enumerate(zip(gen, toks[1:])) if a != b), NTOK - 1)
    # benchmark()'s loop: same tokens teacher-forced, perplexity as the reference prints it
    ids = torch.tensor([toks], device=DEV)
    res = decode.benchmark(model, ids, check=True, engine=eng)
    lp = torch.log_softmax(ref[:-1], -1)
    ppl_ref = float(torch.exp(-lp[torch.arange(NTOK - 1), ids[0, 1:]].mean()))
    assert abs(res["ppl"] / ppl_ref - 1.0) <= 2e-2, (res["ppl"], ppl_ref)
    # and the slower launch sequences of the same engine agree with the chosen one
    for mode in (("v3", "fused", "plain") if extra == 1 else ("plain",)):
        e2 = decode.DecodeEngine(eng.dec, max_len=SEQLEN, mode=mode)
        got2 = torch.stack([e2.forward(t)[0].float().clone() for t in toks])
        report[mode] = _gate(f"{arch}-extra{extra}-{mode}", got2, ref, toks)[0]
    decode.set_mode(eng.dec, eng.mode)
    if extra == 1:
        # serving form: only the decode-order codes stay resident (2 bits per weight, not 4); the fused engine does not notice,
        # the layer-by-layer paths refuse loudly
        before = sum(q.packed_bytes() for q in named.values())
        for q in named.values():
            q.decode_only()
        assert sum(q.packed_bytes() for q in named.values()) * 2 == before
        eng.reset()
        got3 = torch.stack([eng.forward(t)[0].float().clone() for t in toks])
        if arch == "opt":
            assert torch.equal(got3, got)
        else:       # Llama's down_proj K-slices meet through fp32 atomics (csrc/decode_bigp.hip): two runs of ONE launch agree to ~1e-5, and an
            #         fp16 rounding downstream turns that into single-ulp flips: logits of two runs agree to ~1e-3 (measured), not bit for bit
            assert float((got3 - got).norm() / got.norm()) <= 3e-3
            # ... unless the fixed-order meet is on (quant.DETERMINISTIC_SPLITK): then run == run, bit for bit, like OPT
            from quip_amd import quant as _Q
            _Q.DETERMINISTIC_SPLITK = True
            try:
                ed = decode.DecodeEngine(eng.dec, max_len=SEQLEN, mode=eng.mode)
                d1 = torch.stack([ed.forward(t)[0].float().clone() for t in toks])
                ed.reset()
                d2 = torch.stack([ed.forward(t)[0].float().clone() for t in toks])
                assert torch.equal(d1, d2)
                assert float((d1 - got).norm() / got.norm()) <= 3e-3
                ed.reset()
                g1 = ed.generate(toks[0], NTOK - 1)[:, 0].tolist()
                ed.reset()
                assert ed.generate(toks[0], NTOK - 1)[:, 0].tolist() == g1
            finally:
                _Q.DETERMINISTIC_SPLITK = False
        with pytest.raises(RuntimeError):
            next(iter(named.values()))(torch.zeros(1, next(iter(named.values())).infeatures, dtype=torch.float16, device=DEV))
    print(f"e2e {arch} pre_proj_extra={extra}:", report)


def test_engine_on_a_dense_model_is_the_hf_model():
    """no packed layer at all: from_hf on the untouched fp16 model = the architecture alone (plain mode, rocBLAS Linears)"""
    from quip_amd import decode
    from test_gpu_decode_hf import hf_generate, _gate
    for arch in ("opt", "llama"):
        model = build(arch)
        toks, ref = hf_generate(model, 3, NTOK)
        eng = decode.DecodeEngine.from_hf(model, max_len=SEQLEN)
        assert eng.mode == "plain"
        got = torch.stack([eng.forward(t)[0].float().clone() for t in toks])
        rel, _, _ = _gate(arch, got, ref, toks)
        assert rel <= 5e-3, rel
