#!/usr/bin/env python3
"""Launcher that puts quip_amd under the reference driver's module names and runs the driver's
`opt_sequential` on it:

    from gptq import *; from bal import Balance; from near import Nearest; from modelutils import *; from quant import *   (opt.py:6-10)

all resolve to quip_amd.{gptq,bal,near,modelutils,quant} (+ method, vector_balance behind them).  With the reference's
driver files at hand -- QUIP_REFERENCE=/path/to/QuIP, or the copies oracle/stage_ref.py stages into the git-ignored
oracle/_ref/ (they travel to the GPU box with the snapshot) -- the reference's own, unmodified opt.py / llama.py is imported
and its opt_sequential (opt.py:29-190) / llama_sequential (llama.py:36-171) is called; without them the same call sequence
restated in scripts/quantize_opt.py runs and `is_reference` comes back False (tests/test_gpu_driver.py then says so instead
of claiming the drop-in).  Either way every quantisation call lands in libquip_amd.so."""
import importlib
import importlib.util
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ALIASES = ("quant", "method", "vector_balance", "bal", "gptq", "near", "modelutils")


def alias_modules():
    import quip_amd
    for name in ALIASES:
        sys.modules[name] = importlib.import_module(f"quip_amd.{name}")
    return quip_amd


def reference_dir():
    """where the reference's driver files are: $QUIP_REFERENCE, else oracle/_ref when oracle/stage_ref.py has staged them"""
    ref = os.environ.get("QUIP_REFERENCE")
    if ref:
        return ref
    staged = os.path.join(ROOT, "oracle", "_ref")
    return staged if os.path.exists(os.path.join(staged, "opt.py")) else None


def load_driver(restatement=False):
    """(opt_sequential, is_reference): signature opt_sequential(model, dataloader, dev, args) -> (quantizers | report, errors);
    restatement=True forces scripts/quantize_opt.py's restated call sequence even when the reference's opt.py is at hand"""
    alias_modules()
    ref = None if restatement else reference_dir()
    if ref and os.path.exists(os.path.join(ref, "opt.py")):
        spec = importlib.util.spec_from_file_location("opt", os.path.join(ref, "opt.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)              # `from quant import *` etc. bind to quip_amd here
        return mod.opt_sequential, True
    spec = importlib.util.spec_from_file_location("quantize_opt", os.path.join(ROOT, "scripts", "quantize_opt.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def opt_sequential(model, dataloader, dev, args):
        a = types.SimpleNamespace(arch="opt", nsamples=args.nsamples, quant=args.quant, wbits=args.wbits, npasses=args.npasses,
                                  qfn=args.qfn, percdamp=args.percdamp, incoh=bool(args.pre_proj), pack=bool(args.pre_proj_extra),
                                  groupsize=getattr(args, "groupsize", -1))
        report, _ = mod.opt_sequential(model, [b[0] for b in dataloader], dev, a)
        return report, [r["error"] for r in report]
    return opt_sequential, False


def load_llama_driver(restatement=False):
    """llama_sequential(model, dataloader, dev, args) -> (report | quantizers, errors).  With the reference's files at hand its
    own llama.py:36-171 runs on quip_amd (its module global `args` injected, HF's LlamaDecoderLayer.forward wrapped to derive
    the position_embeddings llama.py does not forward, Balance.configure accepting the stray args.qbits of llama.py:110-115 --
    the same three adaptations tests/golden/make_golden.py needed to run it on CPU); otherwise scripts/quantize_opt.py's
    llama_sequential, which carries the fixes itself."""
    alias_modules()
    ref = None if restatement else reference_dir()
    if ref and os.path.exists(os.path.join(ref, "llama.py")):
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "_shims"))     # texttable
        sys.path.insert(0, ref)                                                  # datautils
        spec = importlib.util.spec_from_file_location("llama", os.path.join(ref, "llama.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        from transformers.models.llama import modeling_llama as ML
        import quip_amd.method as M
        import quip_amd.bal as B
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import tiny_model as TM                                                  # balance_configure_shim

        def llama_sequential(model, dataloader, dev, args):
            orig_fwd, orig_free, errors = ML.LlamaDecoderLayer.forward, M.QuantMethod.free, []

            def fwd(self, hidden_states, *a, position_embeddings=None, position_ids=None, **kw):
                if position_embeddings is None:
                    position_embeddings = model.model.rotary_emb(hidden_states, position_ids=position_ids)
                return orig_fwd(self, hidden_states, *a, position_embeddings=position_embeddings, position_ids=position_ids, **kw)

            def free(self):
                errors.append(float(self.error))
                return orig_free(self)
            ML.LlamaDecoderLayer.forward, M.QuantMethod.free, mod.args = fwd, free, args
            try:
                with TM.balance_configure_shim(B.Balance):
                    return mod.llama_sequential(model, dataloader, dev), errors
            finally:
                ML.LlamaDecoderLayer.forward, M.QuantMethod.free = orig_fwd, orig_free
        return llama_sequential, True
    spec = importlib.util.spec_from_file_location("quantize_opt", os.path.join(ROOT, "scripts", "quantize_opt.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def llama_sequential(model, dataloader, dev, args):
        a = types.SimpleNamespace(nsamples=args.nsamples, quant=args.quant, wbits=args.wbits, npasses=args.npasses, qfn=args.qfn,
                                  percdamp=args.percdamp, incoh=bool(args.pre_proj), pack=bool(args.pre_proj_extra),
                                  groupsize=args.groupsize)
        report, _ = mod.llama_sequential(model, [b[0] for b in dataloader], dev, a)
        return report, [r["error"] for r in report]
    return llama_sequential, False


if __name__ == "__main__":
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import tiny_model as TM
    name = sys.argv[1] if len(sys.argv) > 1 else "ldlq_w2_incoh"
    drv, is_ref = load_driver()
    dev = torch.device("cuda:0")
    model = TM.build_tiny_opt().to(dev)
    np.random.seed(0)
    torch.manual_seed(0)
    _, errors = drv(model, TM.calibration_batches(), dev, types.SimpleNamespace(nsamples=TM.NSAMPLES, **TM.CONFIGS[name]))
    print({"config": name, "reference_driver": is_ref, "errors": [float(e) for e in errors]})
