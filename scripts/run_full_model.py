#!/usr/bin/env python3
"""The reference's OWN block-sequential driver at a model's full size, on quip_amd, on one MI355X -- wall time and phase split.

    python scripts/run_full_model.py --model llama-2-7b   # BASELINE configs[3]: llama.py:36-171, 32 blocks x 7 Linears, seqlen 2048
                                                          # (llama.py:31), --wbits 2 --quant ldlq + incoherence processing
    python scripts/run_full_model.py --model opt-125m     # BASELINE configs[0]: opt.py --wbits 4 --quant ldlq at the opt-125m geometry
                                                          # (768 / 3072 / 12), on the GPU (the package has no CPU path)
    python scripts/run_full_model.py --model opt-1.3b     # configs[2]'s model: 24 blocks, w2 + incoherence

The driver file that runs is the reference's (oracle/_ref/opt.py / llama.py, staged by oracle/stage_ref.py; $QUIP_REFERENCE), imported
over the module aliases of scripts/run_reference_driver.py; without it the restated call sequence (scripts/quantize_opt.py) runs and
the output says so.  Model: random init of the named architecture in fp16 (no checkpoint is reachable offline); calibration: nsamples
sequences of seqlen random tokens -- the reference's own default sizes (128 x 2048) unless overridden.

Phase split (seconds, summed over the run): `hessian` = QuantMethod.add_batch (HIP events around every hook call, no synchronisation
added), `post_batch`, `operator_sampling` = method.gen_rand_orthos inside preproc (host RNG + Householder accumulation), `preproc` (the
rest of it: rescale, projection of W and H, damping), `rounding` = fasterquant (grid map, LDL factor, LDLQ sweep, postproc, proxy error),
`free`, and `forwards_and_moves` = everything else inside the driver (the fp16 block forwards -- twice per block, opt.py:141-143,172-174 --
input capture, parking blocks on the CPU).  preproc / fasterquant are bracketed by synchronisations (they synchronise inside anyway)."""
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

MODELS = {
    "llama-2-7b": dict(arch="llama", hidden=4096, ffn=11008, layers=32, heads=32, vocab=32000),
    "opt-125m": dict(arch="opt", hidden=768, ffn=3072, layers=12, heads=12, vocab=50272),
    "opt-1.3b": dict(arch="opt", hidden=2048, ffn=8192, layers=24, heads=32, vocab=50272),
    "opt-6.7b": dict(arch="opt", hidden=4096, ffn=16384, layers=32, heads=32, vocab=50272),
    "opt-30b": dict(arch="opt", hidden=7168, ffn=28672, layers=48, heads=56, vocab=50272),      # BASELINE configs[4]; use --layers 1..N
}


def build(spec, seqlen, dev):
    """random init straight on the GPU in fp16 (HF's own initialisers)"""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float16)
    try:
        with torch.device(dev):
            if spec["arch"] == "llama":
                from transformers import LlamaConfig, LlamaForCausalLM
                cfg = LlamaConfig(hidden_size=spec["hidden"], intermediate_size=spec["ffn"], num_hidden_layers=spec["layers"],
                                  num_attention_heads=spec["heads"], num_key_value_heads=spec["heads"], vocab_size=spec["vocab"],
                                  max_position_embeddings=max(seqlen, 2048), rms_norm_eps=1e-5, tie_word_embeddings=False)
                model = LlamaForCausalLM(cfg)
            else:
                from transformers import OPTConfig, OPTForCausalLM
                cfg = OPTConfig(hidden_size=spec["hidden"], ffn_dim=spec["ffn"], num_hidden_layers=spec["layers"],
                                num_attention_heads=spec["heads"], word_embed_proj_dim=spec["hidden"], vocab_size=spec["vocab"],
                                max_position_embeddings=max(seqlen, 2048))
                model = OPTForCausalLM(cfg)
    finally:
        torch.set_default_dtype(old)
    model = model.half().to(dev).eval()
    model.seqlen = seqlen
    return model


class Phases:
    """wraps the QuantMethod protocol calls of quip_amd with timers (restored on exit)"""

    def __init__(self):
        self.t = {k: 0.0 for k in ("post_batch", "operator_sampling", "preproc", "rounding", "free")}
        self.events = []
        self.offpath = 0.0           # seconds of operator sampling done by the prefetch thread (not part of the wall-clock split)
        self.calls = {"add_batch": 0, "linears": 0}
        self.per_linear = []

    def __enter__(self):
        import quip_amd.method as M
        import quip_amd.bal as B
        import quip_amd.gptq as G
        import quip_amd.near as N
        self._saved = [(M.QuantMethod, "add_batch", M.QuantMethod.add_batch), (M.QuantMethod, "post_batch", M.QuantMethod.post_batch),
                       (M.QuantMethod, "preproc", M.QuantMethod.preproc), (M.QuantMethod, "free", M.QuantMethod.free),
                       (M, "gen_rand_orthos", M.gen_rand_orthos), (B.Balance, "fasterquant", B.Balance.fasterquant),
                       (G.GPTQ, "fasterquant", G.GPTQ.fasterquant), (N.Nearest, "fasterquant", N.Nearest.fasterquant)]
        ph = self

        def timed(key, fn):
            def w(*a, **k):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = fn(*a, **k)
                torch.cuda.synchronize()
                ph.t[key] += time.perf_counter() - t0
                return out
            return w

        add_batch = M.QuantMethod.add_batch

        def add_batch_ev(self_, inp, out):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            add_batch(self_, inp, out)
            e1.record()
            ph.events.append((e0, e1))
            ph.calls["add_batch"] += 1
        M.QuantMethod.add_batch = add_batch_ev
        M.QuantMethod.post_batch = timed("post_batch", M.QuantMethod.post_batch)
        gen = M.gen_rand_orthos
        import threading
        main_thread = threading.get_ident()

        def gen_timed(m, p):
            t0 = time.perf_counter()
            out = gen(m, p)
            dt = time.perf_counter() - t0
            if threading.get_ident() == main_thread:
                ph.t["operator_sampling"] += dt                 # on the critical path, inside preproc
            else:
                ph.offpath += dt                                # method.OPERATOR_PREFETCH: on the host thread, under the GPU phases
            return out
        M.gen_rand_orthos = gen_timed
        M.QuantMethod.preproc = timed("preproc", M.QuantMethod.preproc)
        free = M.QuantMethod.free

        def free_rec(self_):
            ph.per_linear.append({"rows": self_.rows, "columns": self_.columns, "error": float(getattr(self_, "error", float("nan"))),
                                  "Hmag": float(getattr(self_, "Hmag", float("nan"))), "time": float(getattr(self_, "time", float("nan")))})
            ph.calls["linears"] += 1
            return free(self_)
        M.QuantMethod.free = timed("free", free_rec)
        B.Balance.fasterquant = timed("rounding", B.Balance.fasterquant)
        G.GPTQ.fasterquant = timed("rounding", G.GPTQ.fasterquant)
        N.Nearest.fasterquant = timed("rounding", N.Nearest.fasterquant)
        return self

    def __exit__(self, *exc):
        for obj, name, fn in self._saved:
            setattr(obj, name, fn)
        return False

    def report(self, wall):
        torch.cuda.synchronize()
        hess = sum(a.elapsed_time(b) for a, b in self.events) * 1e-3
        t = dict(self.t)
        t["preproc"] -= t["operator_sampling"]                  # sampling runs inside preproc
        t["hessian"] = hess
        t["forwards_and_moves"] = wall - sum(t.values())
        out = {k: round(v, 3) for k, v in t.items()}
        out["operator_sampling_on_prefetch_thread_not_in_wall"] = round(self.offpath, 3)
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-2-7b", choices=sorted(MODELS))
    ap.add_argument("--nsamples", type=int, default=128)
    ap.add_argument("--seqlen", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=0, help="override the block count (0 = the architecture's)")
    ap.add_argument("--wbits", type=int, default=None)
    ap.add_argument("--quant", default="ldlq")
    ap.add_argument("--no-incoh", action="store_true")
    ap.add_argument("--extra", type=int, default=0, help="pre_proj_extra: 0 = what --incoh_processing yields (blocked butterfly), 1 = Kronecker")
    ap.add_argument("--restatement", action="store_true")
    ap.add_argument("--fast-hessian", action="store_true", help="method.HESSIAN_FAST (opt-in, not the reference's arithmetic)")
    ap.add_argument("--device-rng", action="store_true", help="method.DEVICE_RNG (opt-in, not the reference's seeded operators)")
    ap.add_argument("--prefetch-operators", dest="prefetch_operators", action="store_true", default=True,
                    help="method.OPERATOR_PREFETCH: sample the next operators on a host thread (exact: the same draws from the same streams; "
                         "the DEFAULT of this driver since round 5 -- a whole-model run never reseeds between constructing a method and its preproc, "
                         "which is the one thing the library-level default, off, protects against)")
    ap.add_argument("--no-prefetch-operators", dest="prefetch_operators", action="store_false")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    out = run(a)
    line = json.dumps(out)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(line + "\n")


def run(a):
    """a: namespace with the fields of main()'s parser (bench.py builds one for its `quantise_model` leg)"""
    spec = dict(MODELS[a.model])
    if a.layers:
        spec["layers"] = a.layers
    incoh = not a.no_incoh and a.model != "opt-125m"            # configs[0] is plain `--wbits 4 --quant ldlq`
    wbits = a.wbits if a.wbits is not None else (4 if a.model == "opt-125m" else 2)
    dev = torch.device("cuda:0")
    import run_reference_driver as R
    import quip_amd.method as M
    M.HESSIAN_FAST, M.DEVICE_RNG = bool(a.fast_hessian), bool(a.device_rng)
    if a.prefetch_operators:
        M.OPERATOR_PREFETCH = True
    if spec["arch"] == "llama":
        drv, is_ref = R.load_llama_driver(restatement=a.restatement)
    else:
        drv, is_ref = R.load_driver(restatement=a.restatement)
    t0 = time.perf_counter()
    torch.manual_seed(0)
    model = build(spec, a.seqlen, dev)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    rs = np.random.RandomState(1)
    batches = [(torch.from_numpy(rs.randint(0, spec["vocab"], size=(1, a.seqlen))).long(), None) for _ in range(a.nsamples)]
    args = types.SimpleNamespace(nsamples=a.nsamples, quant=a.quant, wbits=wbits, qbits=wbits, qfn=("b" if incoh and a.quant != "gptq" else "a"),
                                 npasses=0, unbiased=False, lazy_batch=False, percdamp=0.01, pre_gptqH=True, pre_rescale=incoh, pre_proj=incoh,
                                 pre_proj_extra=a.extra, groupsize=-1)
    np.random.seed(0)
    torch.manual_seed(0)
    torch.cuda.reset_peak_memory_stats()
    with Phases() as ph:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, errors = drv(model, batches, dev, args)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        phases = ph.report(wall)
    pf_stats = dict(M.operator_prefetcher().stats) if a.prefetch_operators else None
    if hasattr(M, "operator_prefetch_stop"):
        M.operator_prefetch_stop()
    errs = [float(e) for e in errors]
    out = {"model": a.model, "geometry": spec, "driver": ("reference " + ("llama.py:36-171" if spec["arch"] == "llama" else "opt.py:29-190"))
           if is_ref else "scripts/quantize_opt.py (restatement)", "nsamples": a.nsamples, "seqlen": a.seqlen,
           "args": {k: v for k, v in vars(args).items()}, "opt_ins": {"HESSIAN_FAST": M.HESSIAN_FAST, "DEVICE_RNG": M.DEVICE_RNG,
                                                                        "OPERATOR_PREFETCH": bool(getattr(M, "OPERATOR_PREFETCH", False))},
           "operator_prefetch_stats": pf_stats, "wall_s": round(wall, 2), "model_build_s": round(t_build, 2), "phases_s": phases, "add_batch_calls": ph.calls["add_batch"],
           "linears": ph.calls["linears"], "peak_device_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
           "errors_finite": bool(np.all(np.isfinite(errs))), "error_sum": float(np.sum(errs)),
           "per_block_error_sum": [float(np.sum(errs[i:i + len(errs) // spec["layers"]])) for i in range(0, len(errs), len(errs) // spec["layers"])],
           "per_linear_first_block": ph.per_linear[:len(errs) // spec["layers"]], "per_linear_last_block": ph.per_linear[-(len(errs) // spec["layers"]):]}
    if a.prefetch_operators:
        M.OPERATOR_PREFETCH = False
    M.HESSIAN_FAST = M.DEVICE_RNG = False
    del model
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
