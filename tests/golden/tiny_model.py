"""The tiny random-init OPT model and calibration tokens shared by tests/golden/make_golden.py (which runs the REFERENCE's
own opt_sequential on it, on CPU, to produce driver.npz) and tests/test_gpu_driver.py (which runs the same call sequence
on quip_amd, on the GPU, and compares).  Weights come from a seeded numpy RandomState in sorted parameter order, so the
model does not depend on Hugging Face's initialisers or on torch's RNG."""
import numpy as np
import torch

HIDDEN, FFN, LAYERS, HEADS, VOCAB, SEQLEN, NSAMPLES = 256, 1024, 2, 4, 512, 64, 8


def build_tiny_opt(seed=1234):
    from transformers import OPTConfig, OPTForCausalLM
    cfg = OPTConfig(hidden_size=HIDDEN, ffn_dim=FFN, num_hidden_layers=LAYERS, num_attention_heads=HEADS,
                    word_embed_proj_dim=HIDDEN, vocab_size=VOCAB, max_position_embeddings=SEQLEN, dropout=0.0,
                    attention_dropout=0.0, activation_dropout=0.0, layerdrop=0.0)
    model = OPTForCausalLM(cfg)
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):
            if "layer_norm" in name or "layernorm" in name:
                v = (1.0 + 0.1 * rs.randn(*p.shape)) if name.endswith("weight") else 0.05 * rs.randn(*p.shape)
            elif name.endswith("bias"):
                v = 0.02 * rs.randn(*p.shape)
            else:
                v = 0.05 * rs.randn(*p.shape)
            p.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    model = model.half().eval()
    model.seqlen = SEQLEN
    return model


def calibration_batches(seed=1):
    rs = np.random.RandomState(seed)
    toks = rs.randint(0, VOCAB, size=(NSAMPLES, SEQLEN))
    return [(torch.from_numpy(toks[i:i + 1]).long(), None) for i in range(NSAMPLES)]   # (input_ids, _) like datautils loaders


def probe_tokens(seed=2):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.randint(0, VOCAB, size=(2, SEQLEN))).long()


CONFIGS = {
    # name: the reference driver's argparse namespace for this run (opt.py:485-600)
    "nearest_w4": dict(quant="nearest", wbits=4, qfn="a", npasses=0, unbiased=False, lazy_batch=False, percdamp=0.01,
                       pre_gptqH=True, pre_rescale=False, pre_proj=False, pre_proj_extra=0, groupsize=-1),
    "ldlq_w4": dict(quant="ldlq", wbits=4, qfn="a", npasses=0, unbiased=False, lazy_batch=False, percdamp=0.01,
                    pre_gptqH=True, pre_rescale=False, pre_proj=False, pre_proj_extra=0, groupsize=-1),
    "ldlq_w2_incoh": dict(quant="ldlq", wbits=2, qfn="b", npasses=0, unbiased=False, lazy_batch=False, percdamp=0.01,
                          pre_gptqH=True, pre_rescale=True, pre_proj=True, pre_proj_extra=0, groupsize=-1),
}


# ---- the Llama twin (llama.py:36-171) ------------------------------------------------------------------------------------------
L_HIDDEN, L_FFN, L_LAYERS, L_HEADS, L_KV_HEADS = 256, 688, 2, 4, 4          # 688 = 16 * 43: Llama's 11008 = 256 * 43 in small


def build_tiny_llama(seed=4321):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=L_HIDDEN, intermediate_size=L_FFN, num_hidden_layers=L_LAYERS, num_attention_heads=L_HEADS,
                      num_key_value_heads=L_KV_HEADS, vocab_size=VOCAB, max_position_embeddings=SEQLEN, rms_norm_eps=1e-5,
                      attention_dropout=0.0, tie_word_embeddings=False)
    model = LlamaForCausalLM(cfg)
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):
            v = (1.0 + 0.1 * rs.randn(*p.shape)) if "norm" in name else 0.05 * rs.randn(*p.shape)
            p.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    model = model.half().eval()
    model.seqlen = SEQLEN
    return model


LLAMA_CONFIGS = {
    # the branches of llama.py:91-116 that run as shipped (the Balance branch reads an undefined args.qbits)
    "nearest_w4": dict(quant="nearest", wbits=4, qfn="a", npasses=0, unbiased=False, percdamp=0.01,
                       pre_gptqH=True, pre_rescale=False, pre_proj=False, pre_proj_extra=0, groupsize=-1),
    "gptq_w4": dict(quant="gptq", wbits=4, qfn="a", npasses=0, unbiased=False, percdamp=0.01,
                    pre_gptqH=True, pre_rescale=False, pre_proj=False, pre_proj_extra=0, groupsize=-1),
    "gptq_w3_g64": dict(quant="gptq", wbits=3, qfn="a", npasses=0, unbiased=False, percdamp=0.01,
                        pre_gptqH=True, pre_rescale=False, pre_proj=False, pre_proj_extra=0, groupsize=64),
    # BASELINE configs[3] (Llama w2, --incoh_processing): llama.py's Balance branch (llama.py:107-115) passes FIVE arguments
    # (args.quant, args.wbits, args.qbits, args.npasses, unbiased=) to Balance.configure(qmethod, nbits, npasses, unbiased)
    # (bal.py:15) and dies with a TypeError as shipped.  `balance_configure_shim` below drops the stray args.qbits; with that
    # one adaptation the reference's own llama_sequential + Balance + round_ldl produce this golden.
    "ldlq_w2_incoh": dict(quant="ldlq", wbits=2, qbits=2, qfn="b", npasses=0, unbiased=False, percdamp=0.01,
                          pre_gptqH=True, pre_rescale=True, pre_proj=True, pre_proj_extra=0, groupsize=-1),
}


class balance_configure_shim:
    """context manager: Balance.configure(qmethod, nbits, [qbits,] npasses, unbiased=...) -- accepts llama.py:110-115's call
    by dropping the stray third positional (args.qbits, never defined by llama.py's parser).  Applied to the REFERENCE's
    Balance when the golden is generated and to quip_amd's Balance when the reference's llama.py runs on it."""

    def __init__(self, balance_cls):
        self.cls = balance_cls

    def __enter__(self):
        self.orig = orig = self.cls.configure

        def configure(self, qmethod, nbits, *rest, **kw):
            if len(rest) == 2 and 'unbiased' in kw:          # llama.py:110-115: (qbits, npasses), unbiased=...
                rest = rest[1:]
            return orig(self, qmethod, nbits, *rest, **kw)
        self.cls.configure = configure
        return self

    def __exit__(self, *exc):
        self.cls.configure = self.orig
        return False
