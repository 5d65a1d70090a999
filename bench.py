#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on configs[1]: fused 2-bit dequant-GEMM, single 4096x4096 Linear,
w2 qfn-b, bs=16, synthetic data, on N MI355X GPUs of one node (one rank per GPU, independent replicas:
the path is data-parallel, there is no collective on it -> "scaling": "weak").

A step = ONE launch of quipamd_dequant_gemm (C ABI) on one batch x[16,4096] (bf16) already resident in HBM.
Two residency regimes are timed, each as K back-to-back launches captured in a hipGraph (the loop is
launch-bound: a launch is ~2 us of GPU work):
  cold : every launch streams a DIFFERENT packed weight copy out of a 96-copy ring (384 MiB > 256 MiB
         Infinity Cache), i.e. the weights really come from HBM as in a decode step of a large model.
         This is `value` and the `roofline` (bound "hbm").
  warm : the same 4 MiB weight every launch (L2 / Infinity-Cache resident) -> `warm` / `roofline_warm`
         (bound "mfma"), never presented as an HBM fraction.
Besides `value` (bf16 y, SURVEY.md 8(d) byte formula) the line carries `accumulate_contract`: the same layer under
the reference operator's own contract (fp32 y pre-filled by the caller, accumulated in place -- quant.py:226-230),
where K2 may split K over workgroups with fp32 atomics; `decode`: the other half of BASELINE.json's metric, OPT-1.3B w2
decode tok/s at batch 1 (scripts/decode_opt.py, N=1 only); and `sharded_ldlq`: one LDLQ rounding of an OPT-30B-fc1-sized
Linear (28672x7168) with its rows scattered over the N ranks (quip_amd/shard.py; N=1: the kernel alone); `hessian`: one
add_batch call of the calibration pass (K7, fp64 X^T X at the OPT-1.3B fc2 input shape) next to the reference's op.
`roofline.traffic` is the PMC-measured HBM traffic per launch of the last committed rocprofv3 pass
(profiles/k2_pmc_latest.json, FETCH_SIZE corrected x2 as MI355X_MICROARCH.md prescribes), or null.
`cpu_baseline` = what the reference actually runs at inference (dense fake-quant nn.Linear: torch CPU
F.linear, fp32) on the host cores, a bounded sample, rank 0 / N=1 only (kind "reference": the reference has no packed
CPU GEMM, this library call IS its inference op).  `ldlq_cpu_reference`: the reference's own bal.py / vector_balance.py
(staged copies, oracle/ref_ldlq_time.py in a subprocess) timed on the same host cores; `ldlq_cpu_port`: the oracle's C
restatement beside it.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

M = D = 4096
BS = 16
BITS = 2
MAXQ = 3
FLOPS = 2.0 * BS * M * D                                   # 536 870 912
BYTES = M * D * BITS // 8 + 2 * BS * D + 2 * BS * M        # 4 456 448 (bf16 in, bf16 out)
HBM_PEAK_GBS = 8000.0                                      # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TF = 2500.0                                      # dense bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--ring", type=int, default=96, help="number of distinct weight copies for the cold regime")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ldlq", action="store_true", help="skip the sharded-LDLQ side measurement")
    ap.add_argument("--no-decode", action="store_true", help="skip the OPT-1.3B w2 decode tok/s side measurement")
    ap.add_argument("--no-llama", action="store_true", help="skip the Llama-2-7B-architecture decode side measurement (about 40 s)")
    ap.add_argument("--profile-cold-only", action="store_true",
                    help="for rocprofv3 passes: run only the cold bf16 regime (so per-kernel averages are the headline kernel's)")
    ap.add_argument("--eager", action="store_true", help="time eager launches queued behind a spin kernel (default for --steps <= 256)")
    ap.add_argument("--graph", action="store_true", help="time one hipGraph of K launches (default for --steps > 256)")
    ap.add_argument("--preheat-ms", type=float, default=50.0,
                    help="milliseconds of the measured launch itself, on the far end of the weight ring, in front of the W warm-up steps of a timed "
                         "region: the clocks of a GPU that has been serving for a while (0: time from whatever state the set-up left; the line "
                         "carries that figure too, as `from_idle`)")
    ap.add_argument("--no-spin", action="store_true",
                    help="with --eager: launch without the spin kernel in front (each kernel then starts on an idle GPU: the form the "
                         "rocprofv3 passes use, whose per-kernel durations are the kernel alone)")
    return ap.parse_args()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def respawn_if_needed(args, argv=None):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: start the N ranks ourselves -- the same command line the driver
    uses, `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` -- pass
    rank 0's JSON line through, and leave with the launcher's exit code.  Returns None when this process IS a rank (or N == 1)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    sys.stderr.write("bench.py: no launcher environment, starting the ranks: " + " ".join(cmd) + "\n")
    return subprocess.call(cmd, env=env)


def launcher_selftest(args):
    """QUIP_BENCH_SELFTEST=1 (tests/test_bench_launcher.py; never set by the driver): the launch path alone, on CPU -- rendezvous on gloo,
    the barrier + MAX-over-ranks timing reduction of `timed`, and rank 0's single line -- with every GPU leg left out.  The line says so."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"selftest": "launcher only: no GPU leg ran", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "max_over_ranks": float(t.item())}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    rc = respawn_if_needed(args)
    if rc is not None:
        sys.exit(rc)
    if os.environ.get("QUIP_BENCH_SELFTEST") == "1":
        return launcher_selftest(args)
    # The contract is ONE JSON line on stdout.  RCCL writes a version banner to stdout through C stdio whenever a communicator is created
    # (every rank of an N > 1 run; the one-rank group of the sharded_block leg), flushed at process exit, i.e. after python's line.  So
    # file descriptor 1 is pointed at stderr for the whole run and the JSON line goes out through a private copy of the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def emit(line):
        real_stdout.write(line + "\n")
        real_stdout.flush()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # QUIP_BENCH_BACKEND=gloo (never set by the driver): a DRY RUN of the N > 1 flow on a box with fewer GPUs than ranks -- the ranks share
    # the GPUs that exist and the collectives go through host memory.  It validates the control flow (every rank in every collective, the
    # barriers, the max-over-ranks reduction, rank 0's single line); its numbers mean nothing and the line says so.
    backend = os.environ.get("QUIP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from quip_amd import ops, _lib

    # ---- synthetic layer: SURVEY.md 8(d) config A --------------------------------------------------------
    torch.manual_seed(0)
    W = (0.02 * torch.randn(M, D)).to(dev)
    scale = ops.qfnb_scale(W)                                               # K5
    _, codes = ops.quantize(W, "b", scale, None, MAXQ, want_codes=True)     # qfn-b grid, quant.py:10-15
    qs = ops.pack(codes, BITS, ops.LAYOUT_STREAM)                           # K1
    x = torch.randn(BS, D).to(torch.bfloat16).to(dev)
    y = torch.empty(BS, M, dtype=torch.bfloat16, device=dev)

    # parity of the thing being timed (not timed): fp32 dense matmul of the dequantised weights
    What = ops.codes_to_weight(codes, "b", scale, None, MAXQ, out_dtype=torch.float32)
    y32 = ops.dequant_gemm(x, qs, BITS, "b", scale, None, None, out_dtype=torch.float32)
    ref = x.float().double() @ What.double().T
    rel = float((y32.double() - ref).norm() / ref.norm())
    assert rel < 1e-3, f"parity check failed: rel err {rel}"
    del W, What, y32, ref

    ring = [qs] + [qs.clone() for _ in range(max(args.ring, 1) - 1)]
    lib = _lib.load()
    fn = lib.quipamd_dequant_gemm
    vp = ctypes.c_void_p

    # lab switch (A/B runs of kernel configurations on one box, never set by the driver): QUIP_K2_CFG="family,p1,p2" routes the timed
    # launches through quipamd_dequant_gemm_cfg with that configuration forced
    k2cfg = os.environ.get("QUIP_K2_CFG")
    cfg_arr = (ctypes.c_int32 * 4)(*([int(v) for v in k2cfg.split(",")] + [0])[:4]) if k2cfg else None
    fn_cfg = lib.quipamd_dequant_gemm_cfg

    def launch(qw, stream):
        if cfg_arr is not None:
            rc = fn_cfg(vp(x.data_ptr()), 2, vp(qw.data_ptr()), BITS, 1, 1, vp(scale.data_ptr()), vp(0), vp(0),
                        vp(y.data_ptr()), 2, 0, BS, M, D, cfg_arr, stream)
        else:
            rc = fn(vp(x.data_ptr()), 2, vp(qw.data_ptr()), BITS, 1, 1, vp(scale.data_ptr()), vp(0), vp(0),
                    vp(y.data_ptr()), 2, 0, BS, M, D, stream)
        if rc:
            raise RuntimeError(lib.quipamd_last_error())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(weights, steps, warmup, preheat_ms=0.0):
        """returns seconds for exactly `steps` launches (max over ranks).  `launch` is looked up at call time.
        preheat_ms: that many milliseconds of the same launches BEFORE the warm-up steps (over the far end of the ring), so that the
        clocks are where a GPU that has been serving for a while has them."""
        side = torch.cuda.Stream()
        nw = len(weights)
        if nw > 1:
            # One untimed walk over the whole ring, in order: every copy's pages are mapped and translated before the timed
            # region (a K = 20 run would otherwise pay first-touch page-table walks on every launch, which a model whose layers
            # are read once per token does not), while the DATA of the copies the timed launches stream (0 .. K-1, the oldest)
            # has been pushed out of the 256 MiB Infinity Cache by the 384 MiB that followed it.
            with torch.cuda.stream(side):
                st = vp(side.cuda_stream)
                for i in range(nw):
                    launch(weights[i], st)
                side.synchronize()
        # How the K launches reach the GPU.  Up to 256 steps: eager launches enqueued while a spin kernel holds the stream, so they run
        # back to back from the queue.  More: one hipGraph (the host cannot enqueue thousands of launches ahead of the GPU).  A SHORT
        # graph pays its own start-up inside the timed region (K = 20: 5.3-5.4 us per launch as a graph, 4.96-4.98 queued eagerly,
        # 4.77 at K = 2000 either way; profiles/r02s_bench_ring_walk.txt, r02u_bench_launch_modes.txt) -- the kernel is the same.
        use_eager = args.eager or (not args.graph and steps <= 256)
        if preheat_ms > 0:
            with torch.cuda.stream(side):
                st = vp(side.cuda_stream)
                far = max(1, nw - nw // 3) if nw > 1 else 1              # the last two thirds of the ring: the timed copies stay untouched
                t_end, i = time.perf_counter() + preheat_ms * 1e-3, 0
                while time.perf_counter() < t_end:
                    for _ in range(64):
                        launch(weights[nw - 1 - (i % far)], st)
                        i += 1
                    side.synchronize()
        if use_eager:
            with torch.cuda.stream(side):
                st = vp(side.cuda_stream)
                for i in range(warmup):
                    launch(weights[(nw - 1 - i) % nw], st)          # warm-up copies come from the END of the ring: the timed
                barrier()                                             # launches (copies 0 .. K-1) never see a pre-touched copy
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                # the launches are enqueued while a spin kernel holds the stream, so they run back to back from the queue (host launch
                # latency is not step time): ~5 us of host time per launch, the spin (~14 us per launch) covers the first ~400
                if not args.no_spin:
                    torch.cuda._sleep(int(min(steps, 400) * 30000 + 300000))    # generous: a loaded host still stays ahead of the GPU
                for i in range(min(warmup, 8)):
                    launch(weights[(nw - 1 - i) % nw], st)          # pre-roll right in front of the start event
                e0.record(side)
                for i in range(steps):
                    launch(weights[i % nw], st)
                e1.record(side)
                barrier()
            t = e0.elapsed_time(e1) * 1e-3
        else:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                st = vp(side.cuda_stream)
                for i in range(min(warmup, 32)):
                    launch(weights[(nw - 1 - i) % nw], st)
                side.synchronize()
                with torch.cuda.graph(graph, stream=side):
                    cst = vp(torch.cuda.current_stream().cuda_stream)
                    for i in range(steps):
                        launch(weights[i % nw], cst)
            wgraph = None
            if warmup > 0:                                   # W untimed warm-up steps through the same path
                wgraph = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(wgraph, stream=side):
                        cst = vp(torch.cuda.current_stream().cuda_stream)
                        for i in range(warmup):
                            launch(weights[(nw - 1 - i) % nw], cst)   # distinct from the timed copies while K + W <= ring
                wgraph.replay()
            barrier()
            # The K launches are one hipGraph; its ~10 us host-side launch latency is not step time.  A ~100 us spin kernel
            # goes first, the start event and the graph are enqueued behind it while it spins, so the events bracket
            # exactly the K kernels back to back (what rocprofv3's per-kernel durations add up to); the W warm-up launches are
            # replayed once more right in front of the start event.
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(side):
                torch.cuda._sleep(250000)
                if wgraph is not None:
                    wgraph.replay()                          # pre-roll: the timed launches start from a busy, warm GPU
                e0.record(side)
                graph.replay()
                e1.record(side)
            barrier()
            t = e0.elapsed_time(e1) * 1e-3
        if dist is not None:
            tt = torch.tensor([t], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t

    t_idle = None
    if args.preheat_ms > 0 and not args.profile_cold_only:
        t_idle = timed(ring, args.steps, args.warmup)                  # first: the GPU as the set-up left it (a few short launches, then idle)
    t_cold = timed(ring, args.steps, args.warmup, args.preheat_ms)
    if args.profile_cold_only:
        if rank == 0:
            emit(json.dumps({"profile_cold_only": True, "us_per_launch": t_cold / args.steps * 1e6}))
        if dist is not None:
            dist.destroy_process_group()
        return
    t_warm = timed([qs], args.steps, args.warmup, args.preheat_ms)

    # ---- the reference operator's own contract: y fp32, accumulated in place (quant.py:226-230) ---------------
    yacc = torch.zeros(BS, M, dtype=torch.float32, device=dev)
    launch_bf16 = launch

    def launch_acc(qw, stream):
        rc = fn(vp(x.data_ptr()), 2, vp(qw.data_ptr()), BITS, 1, 1, vp(scale.data_ptr()), vp(0), vp(0),
                vp(yacc.data_ptr()), 0, 1, BS, M, D, stream)
        if rc:
            raise RuntimeError(lib.quipamd_last_error())
    launch = launch_acc
    t_acc_cold = timed(ring, args.steps, args.warmup, args.preheat_ms)
    t_acc_warm = timed([qs], args.steps, args.warmup, args.preheat_ms)
    # ---- the timed launch with y written as fp32 (no accumulate): the variant the north_star's 1e-3 tolerance is asserted on above ----
    yf32 = torch.empty(BS, M, dtype=torch.float32, device=dev)

    def launch_f32(qw, stream):
        rc = fn(vp(x.data_ptr()), 2, vp(qw.data_ptr()), BITS, 1, 1, vp(scale.data_ptr()), vp(0), vp(0),
                vp(yf32.data_ptr()), 0, 0, BS, M, D, stream)
        if rc:
            raise RuntimeError(lib.quipamd_last_error())
    launch = launch_f32
    t_f32_cold = timed(ring, args.steps, args.warmup, args.preheat_ms)
    launch = launch_bf16
    BYTES_F32 = M * D * BITS // 8 + 2 * BS * D + 4 * BS * M          # y written as fp32
    BYTES_ACC = M * D * BITS // 8 + 2 * BS * D + 2 * 4 * BS * M      # y read + written as fp32

    traffic, traffic_source = None, None
    pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "k2_pmc_latest.json")
    if os.path.exists(pmc_path):
        with open(pmc_path) as f:
            pmc = json.load(f)
        traffic = pmc.get("hbm_bytes_per_launch")
        traffic_source = ("profiles/k2_pmc_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command "
                          f"(scripts/gpu_round.sh), kernel {str(pmc.get('kernel', '?'))[:60]} -- counters cannot be collected inside the timed run")

    us_cold = t_cold / args.steps * 1e6
    us_warm = t_warm / args.steps * 1e6
    tf_cold = FLOPS * world / (t_cold / args.steps) / 1e12
    tf_warm = FLOPS * world / (t_warm / args.steps) / 1e12
    gbs_cold = BYTES / (t_cold / args.steps) / 1e9         # per GPU
    out = {
        "metric": "2-bit dequant-GEMM TFLOP/s (4096x4096, bs=16)",
        "value": round(tf_cold, 3),
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": us_cold * 1e-3,
        "higher_is_better": True,
        **({"dry_run": f"QUIP_BENCH_BACKEND={backend}: ranks share GPUs, collectives through host memory -- control flow only, the numbers mean nothing"}
           if backend != "nccl" else {}),
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: single 4096x4096 Linear, w2 qfn-b, fused dequant-GEMM, bs=16, "
                               f"cold weights (ring of {len(ring)} packed copies, {len(ring) * qs.numel() * 4 // 2**20} MiB; the timed launches stream copies 0..{min(len(ring), args.steps) - 1}, the warm-up ones come from the other end of the ring"
                               + (f"; {args.preheat_ms:g} ms of the same launch on the far two thirds of the ring precede the warm-up steps (clocks at their serving state; "
                                  "`from_idle` is the same measurement without them)" if args.preheat_ms > 0 else "") + ")", "m": M, "d": D, "bs": BS, "bits": BITS,
                   "launch": ("eager launches queued behind a spin kernel" if (args.eager or (not args.graph and args.steps <= 256))
                              else "one hipGraph of K launches"), "parallelism": f"dp{world} (replicas)"},
        "parity_rel_err": rel,
        "roofline": {"bound": "hbm", "achieved": round(gbs_cold, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(gbs_cold / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": BYTES, "us_per_launch": round(us_cold, 3),
                     "pct_mfma_peak": round(100 * tf_cold / world / MFMA_PEAK_TF, 2),
                     "regime": "achieved / frac / us_per_launch: HIP events around the K launches of THIS run, back to back from the queue"
                               + (f", after {args.preheat_ms:g} ms of the same launch (serving clocks)" if args.preheat_ms > 0 else "")
                               + "; traffic and the committed rocprofv3 kernel average (profiles/*_rocprof_summary.txt) come from separate passes of "
                                 "this command with --eager --no-spin --preheat-ms 0, where every kernel starts on an idle GPU -- the regime of "
                                 "`from_idle`, 2-5 % slower per launch than `us_per_launch`"},
        "fp32_y": {"what": "the same K launches with y written as fp32 (no accumulate): the output the parity check above holds to the north_star's "
                           "1e-3 (the bf16 y of `value` adds its own output rounding, <= 3e-3 in tests/test_gpu_dqgemm.py)",
                   "parity_rel_err": rel, "tolerance": 1e-3, "us_per_launch": round(t_f32_cold / args.steps * 1e6, 3),
                   "value": round(FLOPS * world / (t_f32_cold / args.steps) / 1e12, 3), "unit": "TFLOP/s",
                   "algorithmic_bytes_per_launch": BYTES_F32, "hbm_frac": round(BYTES_F32 / (t_f32_cold / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
        **({"from_idle": {"what": f"the same K = {args.steps} launches after W = {args.warmup} warm-up steps on a GPU that was idle before them (no pre-heat): "
                                  "2-3 % slower, the clocks are still ramping (profiles/r05v_headline_preheat.txt)",
                          "us_per_launch": round(t_idle / args.steps * 1e6, 3), "value": round(FLOPS * world / (t_idle / args.steps) / 1e12, 3), "unit": "TFLOP/s"}}
           if t_idle is not None else {}),
        "warm": {"value": round(tf_warm, 3), "unit": "TFLOP/s", "us_per_launch": round(us_warm, 3)},
        "roofline_warm": {"bound": "mfma", "achieved": round(tf_warm / world, 2), "peak": MFMA_PEAK_TF, "unit": "TFLOP/s",
                          "frac": round(tf_warm / world / MFMA_PEAK_TF, 4)},
        "accumulate_contract": {
            "what": "same layer, y fp32 accumulated in place (the reference operator's contract, quant.py:226-230); "
                    "K2 splits K over workgroups with fp32 atomics",
            "us_per_launch_cold": round(t_acc_cold / args.steps * 1e6, 3), "us_per_launch_warm": round(t_acc_warm / args.steps * 1e6, 3),
            "value": round(FLOPS * world / (t_acc_cold / args.steps) / 1e12, 3), "unit": "TFLOP/s",
            "algorithmic_bytes_per_launch": BYTES_ACC,
            "hbm_GBs": round(BYTES_ACC / (t_acc_cold / args.steps) / 1e9, 1),
            "hbm_frac": round(BYTES_ACC / (t_acc_cold / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
    }

    # ---- the same kernel family on the shapes where it is a stream / a GEMM, and fp16 activations ----------
    def k2_leg(m_, d_, bs_, dt, steps):
        g = torch.Generator().manual_seed(1)
        codes_ = torch.randint(0, 4, (m_, d_), generator=g, dtype=torch.uint8).to(dev)
        q_ = ops.pack(codes_, BITS, ops.LAYOUT_STREAM)
        del codes_
        wb = m_ * d_ * BITS // 8
        nr = max(2, min(96, (400 << 20) // wb + 1))
        ring_ = [q_] + [q_.clone() for _ in range(nr - 1)]
        x_ = torch.randn(bs_, d_, generator=g).to(dt).to(dev)
        y_ = torch.empty(bs_, m_, dtype=dt, device=dev)
        sc_ = torch.tensor([0.05], device=dev)

        def l_(qw, stream):
            rc = fn(vp(x_.data_ptr()), ops._DT[dt], vp(qw.data_ptr()), BITS, 1, 1, vp(sc_.data_ptr()), vp(0), vp(0),
                    vp(y_.data_ptr()), ops._DT[dt], 0, bs_, m_, d_, stream)
            if rc:
                raise RuntimeError(lib.quipamd_last_error())
        nonlocal launch
        keep = launch
        launch = l_
        try:
            tt = timed(ring_, steps, min(steps, 20), args.preheat_ms) / steps
        finally:
            launch = keep
        by = wb + 2 * bs_ * d_ + 2 * bs_ * m_
        return {"us_per_launch": round(tt * 1e6, 3), "GBs": round(by / tt / 1e9, 1), "hbm_frac": round(by / tt / 1e9 / HBM_PEAK_GBS, 4),
                "TFLOPs": round(2.0 * bs_ * m_ * d_ / tt / 1e12, 1), "mfma_frac": round(2.0 * bs_ * m_ * d_ / tt / 1e12 / MFMA_PEAK_TF, 4)}
    if rank == 0 and world == 1 and not args.no_ldlq:
        try:
            del ring
            ring = [qs]
            torch.cuda.empty_cache()
            out["k2_shapes"] = {
                "what": "quipamd_dequant_gemm, w2 qfn b, cold weights, default kernel choice; bytes/flops as SURVEY.md 8(d); every leg behind --preheat-ms of its "
                        "own launch (the bs > 16 legs are 15 % slower on a GPU that was idle a moment ago: profiles/r05u_k2_mb_modes.jsonl)",
                "4096x4096 bs16 fp16": k2_leg(4096, 4096, 16, torch.float16, 200),
                "28672x7168 bs16 bf16 (weight stream)": k2_leg(28672, 7168, 16, torch.bfloat16, 100),
                "28672x7168 bs256 bf16 (MFMA)": k2_leg(28672, 7168, 256, torch.bfloat16, 50),
                "4096x4096 bs2048 bf16 (prefill)": k2_leg(4096, 4096, 2048, torch.bfloat16, 50)}
        except Exception as ex:
            out["k2_shapes"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    # The side measurements below must never cost the headline line: exceptions are caught per leg, and a watchdog on
    # every rank covers a hang (a collective that never completes cannot be caught): when it fires, rank 0 prints the
    # line it has -- the headline is complete at this point -- and every rank leaves.
    import threading
    printed = threading.Lock()

    def emit_and_exit():
        if printed.acquire(blocking=False):
            if rank == 0:
                out.setdefault("side_measurements", "watchdog fired: a side leg did not finish within 540 s")
                emit(json.dumps(out))
            os._exit(0)
    watchdog = threading.Timer(540.0, emit_and_exit)
    watchdog.daemon = True
    watchdog.start()

    # ---- sharded LDLQ: rows of one OPT-1.3B-fc2-sized Linear over the N ranks -----------------
    if not args.no_ldlq:
        try:
            from quip_amd import shard
            lm, ld = 28672, 7168          # OPT-30B fc1 (BASELINE configs[4]); 1792 row groups: enough to fill 8 GPUs
            if rank == 0:
                g = torch.Generator().manual_seed(0)
                Xc = torch.randn(ld + 256, ld, generator=g).to(dev)
                Hh = Xc.T @ Xc / (ld + 256)
                Hh += 0.01 * Hh.diag().mean() * torch.eye(ld, device=dev)
                LT = ops.unit_lower_t(torch.linalg.cholesky(Hh))
                wg = (torch.rand(lm, ld, generator=g) * 3.6 - 0.3).clamp(0, 3).to(dev)
                del Xc, Hh
            else:
                LT = wg = None
            reps = 3
            ts = []
            for it in range(reps + 1):
                barrier()
                t0 = time.perf_counter()
                if world > 1:
                    codes_l = shard.ldlq_round_sharded(wg, LT, BITS)
                else:
                    codes_l = ops.ldlq_round(wg, LT, BITS)
                barrier()
                if it:
                    ts.append(time.perf_counter() - t0)
            tl = float(np.median(ts))
            if dist is not None:
                tt = torch.tensor([tl], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tl = float(tt.item())
            out["sharded_ldlq"] = {"what": f"LDLQ codes of one {lm}x{ld} Linear (OPT-30B fc1 shape), w{BITS}, rows over {world} rank(s); "
                                           "wall time incl. LT broadcast, row scatter, code gather",
                                   "ms": round(tl * 1e3, 3), "far_field_TFLOPs": round(lm * ld * ld / tl / 1e12, 2), "scaling": "strong"}
            if world > 1:                          # one more pass with per-phase synchronisation: where the time and the bytes go
                shard.TIMING = True
                barrier()
                shard.ldlq_round_sharded(wg, LT, BITS)
                barrier()
                shard.TIMING = False
                if rank == 0:
                    st = dict(shard.last_stats)
                    out["sharded_ldlq"]["exchange"] = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in st.items()}
        except Exception as ex:                       # a side measurement must never take the headline line down
            import traceback
            sys.stderr.write(f"[bench.py rank {rank}] sharded_ldlq leg failed:\n{traceback.format_exc()}\n")
            out["sharded_ldlq"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    # ---- the sharded driver on WHOLE models (configs[4] path, north_star "OPT-1.3B full-model LDLQ at 1 / 2 / 4 / 8 GPUs"):
    # `sharded_model`: all 24 blocks of the OPT-1.3B geometry, the reference's calibration size (128 x 2048 tokens) split over the N ranks
    # (strong scaling: the work is fixed); `sharded_block_opt30b`: one block at the OPT-30B geometry (7168 / 28672 / 56 heads), same
    # calibration size.  Every rank runs the SPMD loop of scripts/quantize_opt_sharded.py; rank 0 reports wall, phase split, bytes moved.
    if not args.no_ldlq:
        try:
            import importlib.util as _ilu
            spec = _ilu.spec_from_file_location("quantize_opt_sharded", os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "quantize_opt_sharded.py"))
            smod = _ilu.module_from_spec(spec)
            spec.loader.exec_module(smod)
            common = ["--nsamples", "128", "--seqlen", "2048", "--vocab", "4096", "--incoh", "--quiet"]
            smod.main(["--hidden", "2048", "--ffn", "8192", "--heads", "32", "--layers", "1"] + common)     # warm-up (kernels, allocator, RCCL channels)

            def sharded_leg(geom, what):
                barrier()
                sres = smod.main(geom + common)
                barrier()
                if rank != 0:
                    return None
                ph = sres["phase_seconds_rank0"]
                serial = ph["owner_preproc_factor_s"]
                moved = {k: sres[k] for k in ("bytes_broadcast_LT", "bytes_broadcast_next_LT", "bytes_broadcast_LT_explicit", "bytes_scatter", "bytes_gather", "bytes_broadcast_weights") if k in sres}
                return {"what": what, "wall_s": sres["wall_s"], "phase_seconds_rank0": ph, "bytes_moved": moved, "linears": sres["linears"],
                        "owners": sres["owners"], "samples_rank0": sres["samples_rank0"], "scaling": "strong",
                        "errors_finite": bool(np.all(np.isfinite(sres["errors"]))), "mean_proxy_error": sres["mean_proxy_error"],
                        "amdahl_note": f"phases that run on a Linear's owner only: {serial:.3f} s of {sres['wall_s']:.3f} s on rank 0"}
            r_ = sharded_leg(["--hidden", "2048", "--ffn", "8192", "--heads", "32", "--layers", "24"],
                             f"OPT-1.3B geometry, ALL 24 blocks, LDLQ w{BITS} + incoherence processing, 128 x 2048 calibration tokens over {world} rank(s): per block "
                             "own samples -> K7 partial Hessians -> all-reduce -> per-Linear owner preproc + LDL factors -> row-sharded K4 -> weight broadcast -> "
                             "re-forward (scripts/quantize_opt_sharded.py --calibration sharded --owners per-linear)")
            if rank == 0:
                out["sharded_model"] = r_
            torch.cuda.empty_cache()
            r_ = sharded_leg(["--hidden", "7168", "--ffn", "28672", "--heads", "56", "--layers", "1"],
                             f"ONE block at the OPT-30B geometry (7168 / 28672 / 56 heads; the model has 48), LDLQ w{BITS} + incoherence processing, 128 x 2048 "
                             f"calibration tokens over {world} rank(s), same driver")
            if rank == 0:
                out["sharded_block_opt30b"] = r_
            torch.cuda.empty_cache()
        except Exception as ex:
            import traceback
            sys.stderr.write(f"[bench.py rank {rank}] sharded_model / sharded_block_opt30b leg failed:\n{traceback.format_exc()}\n")
            if rank == 0:
                out["sharded_model"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    # ---- K7 Hessian accumulation: one add_batch call at the OPT-1.3B fc2 input shape -----------------
    if rank == 0 and world == 1 and not args.no_ldlq:
        try:
            ht, hd = 2048, 8192
            xh = torch.randn(ht, hd, device=dev).half()
            Hacc = torch.zeros(hd, hd, dtype=torch.float64, device=dev)

            def ev_time(fn, reps):
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                e1.synchronize()
                return e0.elapsed_time(e1) / reps

            def ref_add():                                           # the reference's op sequence (method.py:115-120)
                x64 = xh.t().to(torch.float64)
                Hacc.add_(x64.matmul(x64.t()))
            t_k7 = ev_time(lambda: ops.hessian_accum(Hacc, xh), 5)
            t_ref = ev_time(ref_add, 2)
            # the opt-in kernel (method.HESSIAN_FAST, off by default: exact products on the 16-bit matrix pipe, fp32 runs of 128 tokens, fp64 across
            # runs -- NOT the reference's arithmetic): its time, and how far one call lands from the exact accumulator
            t_fast = ev_time(lambda: ops.hessian_accum(Hacc, xh, fast=True), 5)
            Ha, Hb = torch.zeros_like(Hacc), torch.zeros_like(Hacc)
            ops.hessian_accum(Ha, xh)
            ops.hessian_accum(Hb, xh, fast=True)
            dg = Ha.diagonal().clamp_min(1e-300).sqrt()
            fast_dev = float(((Ha - Hb).abs().tril() / (dg[:, None] * dg[None, :])).max())
            del Ha, Hb, dg
            tiles = (hd // 128) * (hd // 128 + 1) // 2
            out["hessian"] = {"what": f"H += X^T X in fp64, X = [{ht} tokens, {hd}] fp16 (one add_batch call, OPT-1.3B fc2 input)",
                              "ms": round(t_k7, 3), "fp64_mfma_TFLOPs": round(2.0 * ht * tiles * 128 * 128 / t_k7 / 1e9, 1),
                              "fp64_mfma_peak_TFLOPs": 78.6, "dense_equiv_TFLOPs": round(2.0 * ht * hd * hd / t_k7 / 1e9, 1),
                              "reference_op_fp64_gemm_ms": round(t_ref, 3),
                              "opt_in_fast_kernel": {"what": "method.HESSIAN_FAST = True (library default False): fp16 x fp16 products exact on the 16-bit matrix pipe, fp32 "
                                                             "partial sums over 128 tokens, fp64 across runs; fails the exact kernel's parity gate (tests/test_gpu_hessian.py: "
                                                             "<= 1 ulp of the fp32 H on < 1e-3 of the entries) by construction, so no default path uses it",
                                                     "ms": round(t_fast, 3), "max_abs_dev_over_sqrt_HiiHjj_one_call": fast_dev}}
            del xh, Hacc
        except Exception as ex:
            out["hessian"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    # ---- the quantisation side of a Linear: LDL factor (K8), LDLQ rounding (K4), OPTQ with the qfn-b quantiser -------
    if rank == 0 and world == 1 and not args.no_ldlq:
        try:
            def ev_time2(fn, reps):
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                e1.synchronize()
                return e0.elapsed_time(e1) / reps
            ql = {}
            for dq in (4096, 8192):
                torch.manual_seed(dq)
                Xq = torch.randn(dq + 256, dq, device=dev)
                Hq = Xq.T @ Xq / dq + 0.01 * torch.eye(dq, device=dev)
                del Xq
                t8 = ev_time2(lambda: ops.cholesky_lt(Hq, check=False), 3)
                LTq = ops.cholesky_lt(Hq)
                Wq = torch.rand(dq, dq, device=dev) * 3
                t4 = ev_time2(lambda: ops.ldlq_round(Wq, LTq, BITS), 3)
                ql[f"{dq}x{dq}"] = {"cholesky_lt_ms": round(t8, 3), "cholesky_TFLOPs": round(dq ** 3 / 3 / t8 / 1e9, 1),
                                    "ldlq_round_ms": round(t4, 3), "ldlq_far_field_TFLOPs": round(dq ** 3 / t4 / 1e9, 1)}
                if dq == 4096:
                    FTq = ops.gptq_feedback(Hq)
                    tg = ev_time2(lambda: ops.gptq_round_qfnb(Wq - 1.5, FTq, BITS), 2)
                    ql[f"{dq}x{dq}"]["gptq_qfnb_ms"] = round(tg, 3)
                    del FTq
                del Hq, LTq, Wq
            out["quantise_linear"] = {"what": "K8 LDL factor (fp32 flops d^3 / 3 against the 157 TFLOP/s fp32 matrix pipe), K4 LDLQ rounding w2 (far field m d^2), "
                                              "OPTQ with the per-column qfn-b scale (csrc/gptq_qfnb.hip)", **ql}
            torch.cuda.empty_cache()
        except Exception as ex:
            out["quantise_linear"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    # ---- the other half of BASELINE.json's metric: OPT-1.3B w2 decode tok/s on one GPU (configs[2]) -----------------
    if rank == 0 and world == 1 and not args.no_decode:
        import importlib.util
        ring = None
        torch.cuda.empty_cache()
        spec = importlib.util.spec_from_file_location(
            "decode_opt", os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "decode_opt.py"))
        dmod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(dmod)
        try:
            dres = dmod.run(layers=24, bits=BITS, bs=1, prompt=64, tokens=64)
        except Exception as ex:
            dres = None
            out["decode"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    if rank == 0 and world == 1 and not args.no_decode and dres is not None:
        cands = [k for k in ("packed_w2_v3_head", "packed_w2_v3", "packed_w2_vfused_split_handover", "packed_w2_vfused", "packed_w2_tiled", "packed_w2_chained") if k in dres]
        best = max(cands, key=lambda k: dres[k]["tok_per_s"])
        out["decode"] = {"metric": "OPT-1.3B w2 (incoherence-processed, packed) decode tok/s, batch 1, one hipGraph per token",
                         "value": round(dres[best]["tok_per_s"], 1), "unit": "tok/s",
                         "ms_per_token": round(dres[best]["ms_per_token_median"], 3),
                         "variant": best,
                         "what": "v3 = csrc/decode_fused.hip: U^T(prev) + residual -> LayerNorm -> V -> 2-bit GEMM in ONE launch per packed "
                                 "layer group (fp16 operator pass in the GEMM prologue), + decode attention with the U^T of q/k/v in "
                                 "its prologue: 5 launches per block; v3_head = + embedding / [U^T + final LayerNorm + lm_head + argmax partials] as one launch "
                                 "each (csrc/decode_head.hip) instead of ~11 torch / rocBLAS launches per token; the round-2 variants (9-13 launches) timed beside it: " +
                                 ", ".join(f"{k[10:]} {dres[k]['tok_per_s']:.0f}" for k in cands),
                         "chained_10_launch_tok_per_s": round(dres["packed_w2_chained"]["tok_per_s"], 1),
                         "unchained_tok_per_s": round(dres["packed_w2_fused_attn"]["tok_per_s"], 1),
                         "with_eager_torch_attention_tok_per_s": round(dres["packed_w2_fused"]["tok_per_s"], 1),
                         "unfused_tok_per_s": round(dres["packed_w2"]["tok_per_s"], 1),
                         "dense_fp16_same_harness_tok_per_s": round(dres["dense_fp16_fused_attn"]["tok_per_s"], 1),
                         "dense_fp16_eager_torch_attention_tok_per_s": round(dres["dense_fp16"]["tok_per_s"], 1),
                         "packed_weight_MB": round(dres["packed_w2"]["packed_weight_MB"], 1),
                         "hbm_bound_tok_per_s": round(dres["packed_w2"]["hbm_bound_tok_per_s"]),
                         "roofline": {"bound": "hbm", "what": "packed codes + fp16 head read once per token at 8 TB/s", "achieved": round(dres[best]["tok_per_s"], 1),
                                      "peak": round(dres["packed_w2"]["hbm_bound_tok_per_s"]), "unit": "tok/s",
                                      "frac": round(dres[best]["tok_per_s"] / dres["packed_w2"]["hbm_bound_tok_per_s"], 4)},
                         "data": "random-init OPT-1.3B architecture, nearest-rounded qfn-b codes (scripts/decode_opt.py)"}

    # ---- the Llama half of the decode row (llama.py:418-471): Llama-2-7B architecture, w2, batch 1 (scripts/decode_llama.py) ----
    if rank == 0 and world == 1 and not args.no_decode and not args.no_llama:
        try:
            import importlib.util
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts"))
            spec = importlib.util.spec_from_file_location(
                "decode_llama", os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "decode_llama.py"))
            lmod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(lmod)
            torch.cuda.empty_cache()
            lres = lmod.run(layers=32, bits=BITS, bs=1, prompt=32, tokens=48)
            lbest = max([k for k in ("packed_w2_v3_head", "packed_w2_v3", "packed_w2_fused") if k in lres], key=lambda k: lres[k]["tok_per_s"])
            out["decode_llama"] = {"metric": "Llama-2-7B-architecture w2 (incoherence-processed, packed) decode tok/s, batch 1, one hipGraph per token",
                                   "value": round(lres[lbest]["tok_per_s"], 1), "unit": "tok/s",
                                   "ms_per_token": round(lres[lbest]["ms_per_token_median"], 3),
                                   "variant": lbest,
                                   "what": "6 launches per block (csrc/decode_fused.hip, decode_attn.hip, decode_bigp.hip: the 11008-wide operators cut over "
                                           "p, V_down fused into a split-K 2-bit GEMM) + 2 per token (csrc/decode_head.hip); variants: " +
                                           ", ".join(f"{k[10:]} {lres[k]['tok_per_s']:.0f}" for k in ("packed_w2_v3_head", "packed_w2_v3", "packed_w2_fused") if k in lres),
                                   "round2_fused_13_launch_tok_per_s": round(lres["packed_w2_fused"]["tok_per_s"], 1),
                                   "unfused_tok_per_s": round(lres["packed_w2"]["tok_per_s"], 1),
                                   "dense_fp16_same_harness_tok_per_s": round(lres["dense_fp16"]["tok_per_s"], 1),
                                   "packed_weight_MB": round(lres["packed_w2"]["packed_weight_MB"], 1),
                                   "hbm_bound_tok_per_s": round(lres["packed_w2"]["hbm_bound_tok_per_s"]),
                                   "roofline": {"bound": "hbm", "what": "packed codes + fp16 head read once per token at 8 TB/s", "achieved": round(lres[lbest]["tok_per_s"], 1),
                                                "peak": round(lres["packed_w2"]["hbm_bound_tok_per_s"]), "unit": "tok/s",
                                                "frac": round(lres[lbest]["tok_per_s"] / lres["packed_w2"]["hbm_bound_tok_per_s"], 4)},
                                   "data": "random-init Llama-2-7B architecture, nearest-rounded qfn-b codes (scripts/decode_llama.py)"}
        except Exception as ex:
            out["decode_llama"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    # ---- what a model quantised by the SHIPPED flag decodes at: --incoh_processing leaves pre_proj_extra = 0 (opt.py:596), i.e. blocked butterfly
    # operators; the package engine then runs the packed layers on csrc/ortho_blk.hip + the grouped dequant-GEMM (mode "fused") ------------------
    if rank == 0 and world == 1 and not args.no_decode:
        try:
            import importlib.util, types as _types
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts"))
            spec = importlib.util.spec_from_file_location("decode_engine_bench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "decode_engine_bench.py"))
            emod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(emod)
            eres = emod.run(_types.SimpleNamespace(arch="opt", layers=0, bits=BITS, blocked=True, prompt=32, tokens=32, mode="auto"))
            out["decode_blocked"] = {"metric": "OPT-1.3B w2 decode tok/s with BLOCKED butterfly operators (what opt.py's --incoh_processing yields), batch 1, "
                                               "quip_amd.decode.DecodeEngine", "value": round(eres["tok_per_s"], 1), "unit": "tok/s",
                                     "engine_mode": eres["engine_mode"], "ms_per_token": round(eres["ms_per_token_median"], 3),
                                     "operator_factor_MB_fp16": round(eres["operator_factor_MB_fp16"], 1), "packed_weight_MB": round(eres["packed_weight_MB"], 1),
                                     "roofline": {"bound": "hbm", "what": "codes + fp16 operator factors + fp16 head once per token at 8 TB/s",
                                                  "achieved": round(eres["tok_per_s"], 1), "peak": round(eres["hbm_bound_tok_per_s"]), "unit": "tok/s",
                                                  "frac": round(eres["frac_of_byte_bound"], 4)},
                                     "what": "operator / grouped GEMM / operator launches per packed layer group; the blocked operators on csrc/ortho_blk.hip (one "
                                             "launch per operator at n = 2048, two at n = 8192; q / k / v share theirs); round 3 ran this model on the general K3 "
                                             "launches: 171 tok/s"}
        except Exception as ex:
            out["decode_blocked"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        # several sequences side by side on the same launches (the prologue of every fused launch runs once per sequence, the weights stream once)
        try:
            torch.cuda.empty_cache()
            ns = _types.SimpleNamespace(arch="opt", layers=0, bits=BITS, blocked=False, prompt=32, tokens=32, mode="auto", bs=1, blk_fused_n=-1)
            built, rows = None, {}
            for nb in (2, 4, 8, 16):
                ns.bs = nb
                r_, built = emod.run(ns, model=built, keep=True)
                rows[f"bs{nb}"] = {"tok_per_s": round(r_["tok_per_s"], 1), "ms_per_step": round(r_["ms_per_step_median"], 3), "engine_mode": r_["engine_mode"]}
            del built
            torch.cuda.empty_cache()
            out["decode_batch"] = {"metric": "OPT-1.3B w2 (Kronecker operators) decode, aggregate tok/s with 2 / 4 / 8 / 16 sequences per step (quip_amd.decode.DecodeEngine, "
                                             "one hipGraph per step); batch 1 is the `decode` leg; up to 2 rows (4 at hidden > 2048: quant.two_launch_from) ride in the single fused "
                                             "launch per layer group, more rows run [prologue-only launch, one workgroup per row] + [dequant-GEMM] per group", **rows}
        except Exception as ex:
            out["decode_batch"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    # ---- a whole model through the reference's own driver (opt.py:29-190, staged copy) on quip_amd: OPT-1.3B, 24 blocks, 128 x 2048 tokens ----
    if rank == 0 and world == 1 and not args.no_ldlq:
        try:
            import importlib.util, types as _types
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts"))
            spec = importlib.util.spec_from_file_location("run_full_model", os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "run_full_model.py"))
            fmod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(fmod)
            fres = fmod.run(_types.SimpleNamespace(model="opt-1.3b", nsamples=128, seqlen=2048, layers=0, wbits=None, quant="ldlq", no_incoh=False, extra=0,
                                                   restatement=False, fast_hessian=False, device_rng=False, prefetch_operators=True, out=None))
            out["quantise_model"] = {"what": "OPT-1.3B architecture (random init, fp16), LDLQ w2 + incoherence processing, 128 x 2048 calibration tokens, all 24 blocks "
                                             "through the block-sequential driver on ONE MI355X; BASELINE.md section 2 derives ~28 min of CPU LDLQ + ~1 h of CPU "
                                             "Hessian accumulation for the reference on 8 cores",
                                     "driver": fres["driver"], "wall_s": fres["wall_s"], "phases_s": fres["phases_s"], "linears": fres["linears"],
                                     "errors_finite": fres["errors_finite"], "opt_ins": fres["opt_ins"]}
            # the same run with the opt-in Hessian kernel (not the reference's arithmetic; reported so that the price of exactness is on the line)
            ffast = fmod.run(_types.SimpleNamespace(model="opt-1.3b", nsamples=128, seqlen=2048, layers=0, wbits=None, quant="ldlq", no_incoh=False, extra=0,
                                                    restatement=False, fast_hessian=True, device_rng=False, prefetch_operators=True, out=None))
            out["quantise_model"]["with_opt_in_fast_hessian"] = {"wall_s": ffast["wall_s"], "phases_s": ffast["phases_s"], "opt_ins": ffast["opt_ins"],
                                                                 "errors_finite": ffast["errors_finite"]}
        except Exception as ex:
            out["quantise_model"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        Wd = ops.codes_to_weight(codes, "b", scale, None, MAXQ, out_dtype=torch.float32).cpu()
        xc = x.float().cpu()
        cores = torch.get_num_threads()
        for _ in range(5):
            torch.nn.functional.linear(xc, Wd)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0:
            for _ in range(20):
                torch.nn.functional.linear(xc, Wd)
            n += 20
        dt = (time.perf_counter() - t0) / n
        try:        # the other CPU figure: the oracle's OpenMP restatement of round_ldl (vector_balance.py:155-199) on the host cores
            from oracle import quip_oracle as OR
            import numpy as _np
            rows_s, dl = 4096, 4096                                  # the whole layer, once (~13 s on the GPU box's host)
            try:
                import ctypes as _ct
                omp_threads = int(_ct.CDLL("libgomp.so.1").omp_get_max_threads())
            except Exception:
                omp_threads = os.cpu_count()
            rs = _np.random.default_rng(0)
            Xs = (rs.standard_normal((2 * dl, dl)) * _np.arange(1, dl + 1) ** -0.75).astype(_np.float32)
            Hs = Xs.T @ Xs / (2 * dl)
            Hs += 0.01 * _np.diag(Hs).mean() * _np.eye(dl, dtype=_np.float32)
            Ls = OR.ldl_factor(Hs)
            wg_s = _np.clip(rs.random((rows_s, dl), dtype=_np.float32) * 3.6 - 0.3, 0, 3)
            t0 = time.perf_counter()
            OR.round_ldl(wg_s, Hs, 2, L=Ls)
            t_l = time.perf_counter() - t0
            out["ldlq_cpu_port"] = {"what": f"oracle round_ldl (C, OpenMP over rows) on the full 4096 x 4096 layer, w2, timed once",
                                    "full_layer_s": round(t_l, 2), "cores": omp_threads,
                                    "cores_note": f"OpenMP threads of the oracle's row loop; torch's CPU pool (cpu_baseline) uses {cores} threads; the box has {os.cpu_count()} logical CPUs",
                                    "kind": "port"}
        except Exception as ex:
            out["ldlq_cpu_port"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
        try:        # the reference's OWN files (bal.py, vector_balance.py, method.py, quant.py staged into oracle/_ref/cpu) on the host cores
            import subprocess
            here = os.path.dirname(os.path.abspath(__file__))
            script = os.path.join(here, "oracle", "ref_ldlq_time.py")
            if os.path.exists(os.path.join(here, "oracle", "_ref", "cpu", "bal.py")):
                # thread count: torch's default on this box (one per physical core) is the WRONG setting for this code -- round_ldl is d
                # dependent steps of small mat-vecs, and 128 threads spend their time in the fork / join of each (profiles/r04b: 43.3 s
                # for 2048^2 against 1.5 s on 8 cores) -- so the reference is timed at 8 threads, BASELINE.md section 2's setting, and
                # a small default-thread probe records the ratio
                pr = subprocess.run([sys.executable, script, "--sizes", "2048", "--budget", "40", "--threads", "8"], capture_output=True, text=True, timeout=240)
                rows = [json.loads(l) for l in pr.stdout.splitlines() if l.startswith("{")]
                done = [r for r in rows if "fasterquant_time_attr_s" in r]
                # 4096^2 costs ~12.5 x 2048^2 (d^2 m; measured 75 s vs 6 s on a slow box, 20 s vs 1.5 s on a fast one): run it only where the
                # whole leg then stays under a minute, and say so otherwise
                t2048 = sum(r["fasterquant_time_attr_s"] for r in done) if done else 1e9
                if 12.5 * t2048 <= 45.0:
                    pr = subprocess.run([sys.executable, script, "--sizes", "4096", "--budget", "30", "--threads", "8"], capture_output=True, text=True, timeout=240)
                    rows += [json.loads(l) for l in pr.stdout.splitlines() if l.startswith("{")]
                else:
                    rows.append({"m": 4096, "d": 4096, "skipped": f"predicted {12.5 * t2048:.0f} s on this host (12.5 x the 2048^2 runs): over the leg's budget; "
                                                                  "profiles/r04Y_bench.json has one such run (75 s for lazy_batch False)"})
                probe = []
                try:
                    for th in ("8", "0"):
                        pp = subprocess.run([sys.executable, script, "--sizes", "1024", "--budget", "20", "--threads", th], capture_output=True, text=True, timeout=120)
                        probe += [json.loads(l) for l in pp.stdout.splitlines() if l.startswith("{")]
                except Exception as ex:
                    probe.append({"error": f"{type(ex).__name__}: {ex}"[:120]})
                out["ldlq_cpu_reference"] = {
                    "what": "the reference's Balance.fasterquant (bal.py:21-48 -> vector_balance.py:155-199 round_ldl / 218-291 round_ldl_block), its own "
                            "files run unmodified on this box's host cores: w2 qfn b, W = 0.02 randn fp16, H = X^T X / (d + 256), preproc(gptqH, rescale, "
                            "proj, extra 0) -- BASELINE.md section 2's recipe; seconds = the `.time` attribute fasterquant sets",
                    "kind": "reference", "cores": (done[0]["threads"] if done else None), "logical_cpus": os.cpu_count(),
                    "cores_note": "8 torch threads (BASELINE.md section 2's setting): the box's default, one thread per physical core, is far slower for "
                                  "this chain of small mat-vecs -- see thread_probe_1024 (8 threads vs the default)",
                    "runs": rows, "thread_probe_1024": probe,
                    "sample": "2048^2 (and 4096^2 where the host is fast enough for the leg to stay under a minute), lazy_batch False and True, each layer whole, once"}
            else:
                out["ldlq_cpu_reference"] = {"error": "oracle/_ref/cpu not staged (no reference checkout when build() ran)"}
        except Exception as ex:
            out["ldlq_cpu_reference"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
        out["cpu_baseline"] = {"value": round(FLOPS / dt / 1e12, 4), "unit": "TFLOP/s", "cores": cores,
                               "cores_note": f"torch.get_num_threads() = the threads F.linear used; {os.cpu_count()} logical CPUs on the box",
                               "kind": "reference",
                               "kind_note": "the reference has no packed CPU GEMM: at inference it runs Hugging Face's dense nn.Linear on the "
                                            "fake-quantised weights, i.e. this very torch F.linear call (opt.py:193-299); the reference's LDLQ is "
                                            "timed from its own files in `ldlq_cpu_reference`, the C restatement in `ldlq_cpu_port`.  On a GPU box the reference runs this op "
                                            "on the GPU in fp16: that comparison is `decode.kronecker_operators.dense_fp16_same_harness_tok_per_s` / "
                                            "`decode_llama.dense_fp16_same_harness_tok_per_s`; the CPU figure is the reported baseline the task asks for, not a speed-up claim",
                               "sample": f"{n} calls of torch CPU F.linear fp32 x[16,4096] @ What[4096,4096]^T "
                                         f"(dense fake-quant weights, what the reference runs at inference), "
                                         f"{dt * 1e3:.3f} ms/call"}
    # configs[2] is "OPT-1.3B w2 --incoh_processing": the shipped flag leaves pre_proj_extra = 0 (opt.py:596 sets an unused `proj_extra`), i.e.
    # BLOCKED butterfly operators -- so the `decode` object leads with that model, and the Kronecker model (pre_proj_extra = 1, what the
    # north_star's "--pre_proj" text describes and the fused launches serve) sits beside it
    if rank == 0 and isinstance(out.get("decode_blocked"), dict) and "value" in out["decode_blocked"] and isinstance(out.get("decode"), dict) and "value" in out["decode"]:
        kron = out.pop("decode")
        blk = out.pop("decode_blocked")
        blk["operators"] = "blocked butterfly, pre_proj_extra = 0: what opt.py --incoh_processing yields (configs[2] as shipped)"
        blk["kronecker_operators"] = kron
        blk["kronecker_operators"]["operators"] = "Kronecker, pre_proj_extra = 1 (fused launches: csrc/decode_fused.hip)"
        out["decode"] = blk
    watchdog.cancel()
    if not printed.acquire(blocking=False):
        return                                   # the watchdog is printing
    if rank == 0:
        emit(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
