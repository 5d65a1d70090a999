#!/usr/bin/env python3
"""BASELINE configs[4] as a runnable driver: LDLQ quantisation of an OPT-shaped model on the GPUs of one node (quip_amd/shard.py;
one process per GPU, RCCL over xGMI).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        scripts/quantize_opt_sharded.py --hidden 7168 --ffn 28672 --heads 56 --layers 2 --nsamples 16 --seqlen 512 --incoh

--calibration sharded (default; SPMD, every rank runs this loop -- the reference spends its wall time in the per-sample hook loop,
opt.py:131-145, method.py:98-120, so THAT is what has to scale):
    every rank holds the model and forwards ITS share of the calibration samples (shard.sample_partition) through the current block
    with the add_batch hooks on (K7 partial Hessians)  ->  ONE fp64 SUM all-reduce per Linear (shard.all_reduce_hessians)  ->  the
    owner (rank 0) runs what couples rows: post_batch, preproc (K3), the LDL factors (K8); the block's LT factors are queued
    (ShardedLDLQ.queue_LTs: LT of Linear k+1 travels while all ranks round k)  ->  per Linear: rows scattered, K4 on every rank, packed
    codes gathered (shard.ldlq_round_sharded; the other ranks join with shard.worker_round)  ->  the owner's quantised fp16 weights
    are broadcast (shard.broadcast_weights)  ->  every rank re-forwards its own samples through the quantised block (opt.py:172-174).
--owners per-linear (default, round 4): what couples the rows of a Linear (post_batch, preproc, the LDL factor) runs on ONE rank PER LINEAR,
    the six Linears of a block on up to six ranks at once (shard.assign_owners / shard.block_owner_per_linear); every LT is broadcast from
    its owner, the rows of Linear j are scattered from / gathered at owner(j), and owner(j) broadcasts its quantised weights.  Ranks that do
    not own a Linear consume its random draws (QuantMethod.skip_operators), so the operators are the ones a single-owner run draws:
    per-Linear errors are identical.  --owners rank0 is round 3's single owner (the baseline of the phase split).
--calibration owner (round 2): rank 0 does forwards, Hessians, preproc and factors alone, the others sit in shard.serve() and only
    round -- kept as the baseline the phase split is compared with (Amdahl: <= 1.15x at 8 GPUs for OPT-1.3B).
Blocks stay sequential (their Hessians depend on the quantised predecessors).  With one process (plain `python`) the collectives
still run when --force-exchange is given: the single-GPU way to exercise the RCCL + HIP pack/unpack path.
Prints one JSON line from rank 0: wall time, per-phase seconds and bytes, proxy errors."""
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
    ap.add_argument("--quiet", action="store_true", help="do not print the JSON line (bench.py calls main() and reads the dict)")
    ap.add_argument("--calibration", default="sharded", choices=["sharded", "owner"], help="who forwards the calibration samples (see the module docstring)")
    ap.add_argument("--owners", default="per-linear", choices=["per-linear", "rank0"], help="who runs preproc + the LDL factor of a Linear (see the module docstring)")
    ap.add_argument("--backend", default="nccl")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    own_group = not dist.is_initialized()
    if (args.backend if own_group else dist.get_backend()) != "nccl":
        local %= max(torch.cuda.device_count(), 1)       # a host-memory backend: ranks may share a GPU (tests, bench.py's dry run)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if own_group:
        dist.init_process_group(args.backend, rank=rank, world_size=world, device_id=dev if args.backend == "nccl" else None)
    from quip_amd import bal, quant, shard, vector_balance
    from quip_amd.modelutils import find_layers
    try:
        spmd = args.calibration == "sharded"
        if rank != 0 and not spmd:
            jobs = shard.serve()
            return {"rank": rank, "jobs": jobs}
        from transformers import OPTConfig, OPTForCausalLM
        cfg = OPTConfig(hidden_size=args.hidden, ffn_dim=args.ffn, num_hidden_layers=args.layers, num_attention_heads=args.heads,
                        word_embed_proj_dim=args.hidden, vocab_size=args.vocab, max_position_embeddings=args.seqlen)
        torch.manual_seed(0)                                   # every rank builds the SAME model and calibration set
        np.random.seed(0)
        # random init straight on the GPU in fp16 (HF's own initialisers; a 24-block model takes ~20 s to initialise on the host): the device
        # generator is seeded by manual_seed too, and the same seed gives the same Philox stream on every rank's GPU
        old_dtype = torch.get_default_dtype()
        torch.set_default_dtype(torch.float16)
        try:
            with torch.device(dev):
                model = OPTForCausalLM(cfg)
        finally:
            torch.set_default_dtype(old_dtype)
        model = model.half().to(dev).eval()
        model.config.use_cache = False
        g = torch.Generator().manual_seed(1)
        batches = [torch.randint(0, args.vocab, (1, args.seqlen), generator=g) for _ in range(args.nsamples)]
        lo, hi = shard.sample_partition(args.nsamples, world)[rank] if spmd else (0, args.nsamples)
        mine = hi - lo                                         # this rank's calibration samples
        layers = model.model.decoder.layers
        inps = torch.zeros((max(mine, 1), args.seqlen, args.hidden), dtype=torch.float16, device=dev)
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
            for b in batches[lo:hi] if mine else batches[:1]:     # (a rank without samples still needs the layer's keyword arguments)
                try:
                    model(b.to(dev))
                except ValueError:
                    pass
        layers[0] = layers[0].module
        kwargs = {k: v for k, v in cache["kwargs"].items() if "past" not in k and "cache" not in k}
        outs = torch.zeros_like(inps)

        handle = shard.ShardedLDLQ(force_exchange=args.force_exchange, spmd=spmd)
        if rank == 0:
            shard.activate(handle)
        qfn = "b" if args.incoh else "a"
        report, totals = [], {"bytes_broadcast_LT": 0, "bytes_broadcast_next_LT": 0, "bytes_scatter": 0, "bytes_gather": 0}
        phases = {"forward_hessian_s": 0.0, "allreduce_s": 0.0, "owner_preproc_factor_s": 0.0, "round_s": 0.0, "broadcast_weights_s": 0.0,
                  "reforward_s": 0.0}
        per_linear = spmd and args.owners == "per-linear"
        if per_linear:
            phases["broadcast_LT_s"] = 0.0
        owner_table = None
        bytes_weights = 0
        bytes_lt_explicit = 0                                  # per-linear owners: the LT slabs broadcast outside the rounding jobs

        def tick():
            torch.cuda.synchronize()
            return time.perf_counter()

        def run_block(layer, src, dst):
            for j in range(mine):
                o = layer(src[j].unsqueeze(0), **kwargs)
                dst[j] = o[0] if isinstance(o, (tuple, list)) else o
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
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
                t0 = tick()
                hooks = [subset[n].register_forward_hook(lambda _, inp, out, n=n: methods[n].add_batch(inp[0].data, out.data)) for n in subset]
                run_block(layer, inps, outs)
                for h in hooks:
                    h.remove()
                t1 = tick()
                if spmd:                                           # ONE exchange step per block: fp64 partial Hessians, summed
                    shard.all_reduce_hessians(list(methods.values()))
                t2 = tick()
                if per_linear:
                    mlist = list(methods.values())
                    owners = shard.assign_owners([(m.rows, m.columns) for m in mlist], world)
                    owner_table = owners
                    blk = {}

                    def prepare(m):
                        m.post_batch()
                        m.preproc(preproc_gptqH=True, percdamp=0.01, preproc_rescale=args.incoh, preproc_proj=args.incoh, preproc_proj_extra=0)
                        return m.H, vector_balance._ldl_transposed(m.H)

                    def finish(m):
                        m.fasterquant(lazy_batch=False)
                        for k in totals:
                            totals[k] += shard.last_stats.get(k, 0)
                        return m.error
                    errs = shard.block_owner_per_linear(mlist, list(subset.values()), owners, prepare,
                                                        lambda m: m.skip_operators(args.incoh, 0), finish,
                                                        force_exchange=args.force_exchange, timers=blk)
                    for name, e in zip(methods, errs):
                        report.append({"layer": i, "name": name, "error": float(e)})
                    for m in mlist:
                        m.free()
                    t6a = tick()
                    run_block(layer, inps, outs)
                    t6 = tick()
                    inps, outs = outs, inps
                    phases["forward_hessian_s"] += t1 - t0
                    phases["allreduce_s"] += t2 - t1
                    for k in ("owner_preproc_factor_s", "broadcast_LT_s", "round_s", "broadcast_weights_s"):
                        phases[k] += blk.get(k, 0.0)
                    bytes_weights += blk.get("bytes_broadcast_weights", 0)
                    bytes_lt_explicit += blk.get("bytes_broadcast_LT", 0)
                    phases["reforward_s"] += t6 - t6a
                    continue
                if rank == 0:
                    for m in methods.values():                     # everything that couples rows: owner only
                        m.post_batch()
                        m.preproc(preproc_gptqH=True, percdamp=0.01, preproc_rescale=args.incoh, preproc_proj=args.incoh, preproc_proj_extra=0)
                    handle.queue_LTs([(m.H, vector_balance._ldl_transposed(m.H)) for m in methods.values()])
                    t3 = tick()
                    for name, m in methods.items():
                        m.fasterquant(lazy_batch=False)
                        for k in totals:
                            totals[k] += shard.last_stats.get(k, 0)
                        report.append({"layer": i, "name": name, "error": float(m.error)})
                        m.free()
                else:
                    t3 = tick()
                    ready = None
                    for name in subset:                            # the same Linears in the same order: one rounding job each
                        ready = shard.worker_round(ready)
                    for m in methods.values():
                        m.free()
                t4 = tick()
                if spmd:
                    bytes_weights += shard.broadcast_weights(list(subset.values()))
                t5 = tick()
                run_block(layer, inps, outs)                       # opt.py:172-174: the next block sees the quantised one
                t6 = tick()
                inps, outs = outs, inps
                for k, dt_ in zip(phases, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
                    phases[k] += dt_
        torch.cuda.synchronize()
        if world > 1 and spmd:
            dist.barrier()
        wall = time.perf_counter() - t_start
        handle.shutdown()
        shard.activate(None)
        if rank != 0:
            return {"rank": rank, "samples": mine}
        out = {"world": world, "backend": args.backend, "calibration": args.calibration, "wall_s": round(wall, 3), "linears": len(report),
               "mean_proxy_error": float(np.mean([r["error"] for r in report])), "errors": [r["error"] for r in report], **totals,
               "bytes_broadcast_weights": bytes_weights, "bytes_broadcast_LT_explicit": bytes_lt_explicit, "samples_rank0": mine, "owners": args.owners if spmd else "rank0",
               "owner_of_each_linear_last_block": owner_table,
               "phase_seconds_rank0": {k: round(v, 4) for k, v in phases.items()},
               "config": {k: v for k, v in vars(args).items()}}
        if not args.quiet:
            print(json.dumps(out), flush=True)
        return out
    finally:
        if own_group:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
