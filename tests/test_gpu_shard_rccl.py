"""-m gpu (one GPU): the RCCL branch of quip_amd/shard.py for real -- a world-1 `nccl` process group with force_exchange:
broadcast / scatter / gather run through RCCL on HIP memory, the codes travel STREAM-packed (HIP pack kernel before the
gather, HIP unpack after it), K4 rounds the chunk.  Until now that branch had only been exercised with gloo and an injected
CPU kernel.  Then the torchrun-able sharded driver (scripts/quantize_opt_sharded.py) end to end with one
rank, against the unsharded driver sequence."""
import importlib.util
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_group():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield
    dist.destroy_process_group()


def _fixture(m, d, seed):
    from quip_amd import ops
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(d, d, generator=g) / d ** 0.5
    X = (torch.randn(2 * d, d, generator=g) * torch.arange(1, d + 1) ** -0.75) @ A
    H = (X.T @ X / (2 * d)).to(DEV)
    H = H + 0.01 * H.diag().mean() * torch.eye(d, device=DEV)
    LT = ops.cholesky_lt(H)
    W = (torch.rand(m, d, generator=g) * 3.6 - 0.3).clamp(0, 3).to(DEV)
    return W, LT


@pytest.mark.parametrize("m,d,bits", [(256, 512, 2), (1000, 1024, 2), (96, 256, 4)])
def test_rccl_exchange_with_hip_pack_unpack(nccl_group, m, d, bits):
    from quip_amd import ops, shard
    W, LT = _fixture(m, d, seed=m)
    want = ops.ldlq_round(W, LT, bits)
    h = shard.ShardedLDLQ(force_exchange=True)
    got = h.round(W, LT, bits)
    assert torch.equal(got, want)
    st = shard.last_stats
    assert st["world"] == 1 and st["bytes_broadcast_LT"] == shard.lt_bytes(d) and st["bytes_gather"] == 0 and st["bytes_scatter"] == 0
    # the queued form: LT of job 2 is broadcast under the rounding of job 1
    W2, LT2 = _fixture(m, d, seed=m + 1)
    Ha, Hb = torch.zeros(2, 2), torch.zeros(2, 2)                       # the queue is keyed by the identity of the H tensor
    h.queue_LTs([(Ha, LT), (Hb, LT2)])
    a = h.round(W, None, bits, key=shard.h_key(Ha))
    assert shard.last_stats["bytes_broadcast_next_LT"] == shard.lt_bytes(d)
    b = h.round(W2, None, bits, key=shard.h_key(Hb))
    assert shard.last_stats["bytes_broadcast_LT"] == 0
    assert torch.equal(a, want) and torch.equal(b, ops.ldlq_round(W2, LT2, bits))
    # unpacked gather (the branch RCCL takes when the shape does not pack) gives the same codes
    c = shard.ldlq_round_sharded(W, LT, bits, force_exchange=True, gather_packed=False)
    assert torch.equal(c, want)


@pytest.mark.parametrize("qfn,bits,dt", [('b', 2, torch.float16), ('a', 4, torch.float16), ('b', 2, torch.bfloat16)])
def test_rows_travel_in_their_own_dtype(nccl_group, qfn, bits, dt):
    """round 6: quantize_weight_vecbal under a row-sharding handle scatters a 16-bit layer as it is (2 B per weight) and maps each chunk
    onto the grid with K5 on the receiving rank: codes, weights and grid parameters bit for bit those of the one-process call; the LT factor
    travels as upper slabs only (0.5625 d^2 words) and arrives with exact zeros below"""
    from quip_amd import ops, shard, vector_balance as vb, quant
    m, d = 200, 1024
    g = torch.Generator().manual_seed(3)
    A = torch.randn(d, d, generator=g) / d ** 0.5
    X = (torch.randn(2 * d, d, generator=g) * torch.arange(1, d + 1) ** -0.75) @ A
    H = (X.T @ X / (2 * d)).to(DEV)
    H = H + 0.01 * H.diag().mean() * torch.eye(d, device=DEV)
    W = (0.02 * torch.randn(m, d, generator=g)).to(DEV).to(dt)
    q = quant.Quantizer()
    q.configure(bits, perchannel=True, sym=False, qfn=qfn, mse=False)
    q.find_params(W, weight=True)
    kw = dict(w=W, H=H, nbits=bits, npasses=0, scale=q.scale, zero=q.zero, maxq=q.maxq, qfn=qfn, qmethod='ldlq', return_codes=True)
    want = vb.quantize_weight_vecbal(**kw)
    shard.activate(shard.ShardedLDLQ(force_exchange=True))
    try:
        got = vb.quantize_weight_vecbal(**kw)
        st = dict(shard.last_stats)
        vb.SHARD_RAW16 = False
        old = vb.quantize_weight_vecbal(**kw)
        st_old = dict(shard.last_stats)
    finally:
        vb.SHARD_RAW16 = True
        shard.activate(None)
    assert st["scatter_form"] == "raw16" and st_old["scatter_form"] == "grid32" and st["bytes_broadcast_LT"] == shard.lt_bytes(d) < 0.6 * 4 * d * d
    for a, b, c in zip(want, got, old):
        if a is None:
            assert b is None and c is None
        else:
            assert torch.equal(a, b) and torch.equal(a, c)
    # the slab broadcast itself: upper part equal, exact zeros below the diagonal on a receiver-shaped buffer
    LT = ops.cholesky_lt(H)
    got_lt, h = shard.broadcast_LT(LT, d, 0, None, torch.device(DEV), async_op=True)
    h.wait()
    assert torch.equal(got_lt, LT)


def test_sharded_driver_script_one_rank(nccl_group):
    """scripts/quantize_opt_sharded.py with --force-exchange == the same block sequence without the exchange."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("quantize_opt_sharded", os.path.join(root, "scripts", "quantize_opt_sharded.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = ["--hidden", "256", "--ffn", "1024", "--heads", "4", "--layers", "2", "--nsamples", "4", "--seqlen", "64", "--vocab", "512", "--incoh"]
    argv = argv + ["--owners", "rank0"]                       # round 3's single owner: the LT prefetch queue and its byte counters
    a = mod.main(argv + ["--force-exchange"])
    b = mod.main(argv)
    assert a["linears"] == b["linears"] == 12
    assert a["bytes_broadcast_LT"] > 0 and a["bytes_broadcast_next_LT"] > 0 and b["bytes_broadcast_LT"] == 0
    assert abs(a["mean_proxy_error"] - b["mean_proxy_error"]) <= 1e-6 * abs(b["mean_proxy_error"])
    # sample-sharded calibration (SPMD loop: own samples -> partial Hessians -> all-reduce -> owner factors -> row-sharded rounding ->
    # weight broadcast -> re-forward) against the round-2 owner-only loop: with one rank the summation orders coincide, so every
    # per-Linear proxy error is the SAME number; the phase split is reported
    c = mod.main(argv + ["--force-exchange", "--calibration", "owner"])
    assert a["calibration"] == "sharded" and c["calibration"] == "owner" and a["errors"] == c["errors"]
    ph = a["phase_seconds_rank0"]
    assert set(ph) == {"forward_hessian_s", "allreduce_s", "owner_preproc_factor_s", "round_s", "broadcast_weights_s", "reforward_s"}
    assert ph["forward_hessian_s"] > 0 and ph["round_s"] > 0
    # round 4: one owner PER LINEAR (shard.block_owner_per_linear).  With one rank every Linear is rank 0's, the LTs go through the
    # explicit broadcasts and the preloaded jobs: the same per-Linear errors as the single-owner loop, number for number
    base = [x for x in argv if x not in ("--owners", "rank0")]
    p = mod.main(base + ["--force-exchange"])
    q = mod.main(base)
    assert p["owners"] == "per-linear" and p["errors"] == a["errors"] and q["errors"] == a["errors"]
    assert set(p["phase_seconds_rank0"]) == set(ph) | {"broadcast_LT_s"} and p["owner_of_each_linear_last_block"] == [0] * 6
    assert p["bytes_broadcast_LT"] == 0 and q["bytes_broadcast_LT"] == 0      # the LTs travel outside the rounding jobs (explicit broadcasts)
    assert p["phase_seconds_rank0"]["broadcast_LT_s"] >= 0 and p["phase_seconds_rank0"]["owner_preproc_factor_s"] > 0
