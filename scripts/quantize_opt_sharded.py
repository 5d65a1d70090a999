#!/usr/bin/env python3
"""BASELINE configs[4] as a runnable driver: LDLQ quantisation of an OPT-shaped model with the rows of every Linear sharded
over the GPUs of one node (quip_amd/shard.py; one process per GPU, RCCL over xGMI).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        scripts/quantize_opt_sharded.py --hidden 7168 --ffn 28672 --heads 56 --layers 2 --nsamples 16 --seqlen 512 --incoh

Rank 0 is the owner: it holds the model, runs the block forwards and the Hessian pass (K7), preproc (K3) and the Cholesky
factors (K8) -- everything that couples rows -- then announces one rounding job per Linear; the other ranks sit in
shard.serve().  Per transformer block (opt.py:97-181, same call order as scripts/quantize_opt.py):
    hooks -> add_batch over the calibration samples -> post_batch + preproc for every Linear of the block
    -> the block's LT factors are queued (shard.ShardedLDLQ.queue_LTs): LT of Linear k+1 is broadcast while all ranks round k
    -> fasterquant per Linear: broadcast LT (first Linear only) / scatter grid rows / K4 on every rank / gather packed codes
Blocks stay sequential (their Hessians depend on the quantised predecessors).  With one process (plain `python`) the
collectives still run when --force-exchange is given: the single-GPU way to exercise the RCCL + HIP pack/unpack path.
Prints one JSON line from rank 0: wall time, per-phase bytes, proxy errors."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--ffn", type=int, default=3072)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--vocab", type=int, default=4096)
    ap.add_argument("--nsamples", type=int, default=8)
    ap.add_argument("--seqlen", type=int, default=128)
    ap.add_argument("--wbits", type=int, default=2)
    ap.add_argument("--incoh", action="store_true")
    ap.add_argument("--force-exchange", action="store_true", help="run the collectives even with one rank")
    ap.add_argument("--backend", default="nccl")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group(args.backend, rank=rank, world_size=world, device_id=dev if args.backend == "nccl" else None)
    from quip_amd import bal, quant, shard, vector_balance
    from quip_amd.modelutils import find_layers
    try:
        if rank != 0:
            jobs = shard.serve()
            return {"rank": rank, "jobs": jobs}
        from transformers import OPTConfig, OPTForCausalLM
        cfg = OPTConfig(hidden_size=args.hidden, ffn_dim=args.ffn, num_hidden_layers=args.layers, num_attention_heads=args.heads,
                        word_embed_proj_dim=args.hidden, vocab_size=args.vocab, max_position_embeddings=args.seqlen)
        torch.manual_seed(0)
        np.random.seed(0)
        model = OPTForCausalLM(cfg).half().to(dev).eval()
        model.config.use_cache = False
        g = torch.Generator().manual_seed(1)
        batches = [torch.randint(0, args.vocab, (1, args.seqlen), generator=g) for _ in range(args.nsamples)]
        layers = model.model.decoder.layers
        inps = torch.zeros((args.nsamples, args.seqlen, args.hidden), dtype=torch.float16, device=dev)
        cache = {"i": 0, "kwargs": None}

        class Catcher(torch.nn.Module):
            def __init__(self, module):
                super().__init__()
                self.module = module

            def forward(self, inp, **kwargs):
                inps[cache["i"]] = inp
                cache["i"] += 1
                cache["kwargs"] = kwargs
                raise ValueError
        layers[0] = Catcher(layers[0])
        with torch.no_grad():
            for b in batches:
                try:
                    model(b.to(dev))
                except ValueError:
                    pass
        layers[0] = layers[0].module
        kwargs = {k: v for k, v in cache["kwargs"].items() if "past" not in k and "cache" not in k}
        outs = torch.zeros_like(inps)

        handle = shard.ShardedLDLQ(force_exchange=args.force_exchange)
        shard.activate(handle)
        qfn = "b" if args.incoh else "a"
        report, totals = [], {"bytes_broadcast_LT": 0, "bytes_broadcast_next_LT": 0, "bytes_scatter": 0, "bytes_gather": 0}
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        with torch.no_grad():
            for i, layer in enumerate(layers):
                subset = find_layers(layer)
                methods = {}
                for name, lin in subset.items():
                    m = bal.Balance(lin)
                    m.configure("ldlq", args.wbits, 0, unbiased=False)
                    m.quantizer = quant.Quantizer()
                    m.quantizer.configure(args.wbits, perchannel=True, sym=False, qfn=qfn, mse=False)
                    methods[name] = m
                hooks = [subset[n].register_forward_hook(lambda _, inp, out, n=n: methods[n].add_batch(inp[0].data, out.data)) for n in subset]
                for j in range(args.nsamples):
                    o = layer(inps[j].unsqueeze(0), **kwargs)
                    outs[j] = o[0] if isinstance(o, (tuple, list)) else o
                for h in hooks:
                    h.remove()
                for m in methods.values():                         # everything that couples rows: owner only
                    m.post_batch()
                    m.preproc(preproc_gptqH=True, percdamp=0.01, preproc_rescale=args.incoh, preproc_proj=args.incoh, preproc_proj_extra=0)
                handle.queue_LTs([(m.H, vector_balance._ldl_transposed(m.H)) for m in methods.values()])
                for name, m in methods.items():
                    m.fasterquant(lazy_batch=False)
                    for k in totals:
                        totals[k] += shard.last_stats.get(k, 0)
                    report.append({"layer": i, "name": name, "error": float(m.error)})
                    m.free()
                for j in range(args.nsamples):
                    o = layer(inps[j].unsqueeze(0), **kwargs)
                    outs[j] = o[0] if isinstance(o, (tuple, list)) else o
                inps, outs = outs, inps
        torch.cuda.synchronize()
        wall = time.perf_counter() - t_start
        handle.shutdown()
        shard.activate(None)
        out = {"world": world, "backend": args.backend, "wall_s": round(wall, 3), "linears": len(report),
               "mean_proxy_error": float(np.mean([r["error"] for r in report])), **totals,
               "config": {k: v for k, v in vars(args).items()}}
        print(json.dumps(out), flush=True)
        return out
    finally:
        if own_group:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
