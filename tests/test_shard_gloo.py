"""CPU, world_size 2, gloo: the row-sharded LDLQ exchange (quip_amd/shard.py).
The per-chunk kernel is injected (the oracle's kernel-order restatement) because the HIP kernel needs a GPU; what is
under test is the partition, the broadcast / scatter / gather plumbing, the worker loop, and that the sharded result
is bit-identical to the unsharded one (rows are independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_compute(wgrid, LT, bits, eta):
    from oracle import quip_oracle as O
    codes = O.round_ldl_kernel_order(wgrid.numpy().astype(np.float32), LT.numpy().astype(np.float32), bits,
                                     eta=None if eta is None else eta.numpy().astype(np.float32))
    return torch.from_numpy(np.ascontiguousarray(codes).astype(np.uint8))


def _oracle_gridmap(w, qfn, scale, zero, maxq):
    """the grid map in the layer's own dtype (vector_balance.py:515, 522-524 as oracle.gridmap_qfnb / gridmap_qfna restate it), given the
    grid parameters -- what K5 computes on a rank's row chunk"""
    from oracle import quip_oracle as O
    wn = w.numpy()
    if qfn == 'b':
        dt = wn.dtype.type
        s = dt(float(scale.reshape(-1)[0]))
        return torch.from_numpy(np.clip(((wn / s + dt(1)) / dt(2)) * dt(maxq), 0, maxq).astype(np.float32))
    return torch.from_numpy(O.gridmap_qfna(wn, scale.numpy().reshape(-1, 1), zero.numpy().reshape(-1, 1), maxq))


def _fixture(m, d, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(d, d, generator=g) / d ** 0.5
    X = (torch.randn(2 * d, d, generator=g) * torch.arange(1, d + 1) ** -0.75) @ A     # correlated Hessian
    H = X.T @ X / (2 * d)
    H = H + 0.01 * H.diag().mean() * torch.eye(d)
    C = torch.linalg.cholesky(H.double()).float()
    LT = torch.zeros(d, d)
    Lunit = C / C.diag()[None, :]
    LT = torch.triu(Lunit.T.contiguous(), diagonal=1).contiguous()                      # LT[c][j] = L[j][c], j > c
    W = (torch.rand(m, d, generator=g) * 3.6 - 0.3).clamp(0, 3)
    eta = torch.rand(m, d, generator=g)
    return W, LT, eta


def _worker(rank, world, port, m, d, bits, use_eta, mode, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quip_amd import shard
    try:
        W, LT, eta = _fixture(m, d, seed=5)
        eta = eta if use_eta else None
        if mode == "queue":                          # a block's LT factors queued: LT k+1 travels under the rounding of k
            W2, LT2, _ = _fixture(m, d // 2, seed=6)
            if rank == 0:
                h = shard.ShardedLDLQ(compute=_oracle_compute)
                H1, H2, H3, Hx = (torch.zeros(2, 2) for _ in range(4))       # stand-ins: the queue is keyed by the H tensor's identity
                k1, k2, k3, kx = (shard.h_key(t) for t in (H1, H2, H3, Hx))
                h.queue_LTs([(H1, LT), (H2, LT2), (H3, LT)])
                assert h.queued(k1) and not h.queued(k2)
                got = h.round(W, None, bits, eta=eta, key=k1)
                assert shard.last_stats["bytes_broadcast_LT"] == shard.lt_bytes(d) and shard.last_stats["bytes_broadcast_next_LT"] == shard.lt_bytes(d // 2)
                got2 = h.round(W2, None, bits, key=k2)
                assert shard.last_stats["bytes_broadcast_LT"] == 0     # came with the previous job
                got3 = h.round(W[: m // 2], None, bits, key=k3)
                assert not h.queued() and "bytes_broadcast_next_LT" not in shard.last_stats
                got4 = h.round(W2, LT2, bits)                          # and a plain job after the queue has drained
                assert torch.equal(got2, _oracle_compute(W2, LT2, bits, None)) and torch.equal(got4, got2)
                assert torch.equal(got3, _oracle_compute(W[: m // 2], LT, bits, None))
                # desync: the queue says H1 -> LT, H2 -> LT2, but the caller rounds a Hessian the queue does not know
                # (ldlqRG's permuted copy, a skipped Linear).  The queued factor and the LT2 already prefetched under job 5 must
                # NOT be used: job 6 runs with the LT it was handed, the queue is dropped, job 7 is a plain job again.
                h.queue_LTs([(H1, LT), (H2, LT2)])
                got5 = h.round(W, None, bits, key=k1)
                assert shard.last_stats["bytes_broadcast_next_LT"] == shard.lt_bytes(d // 2)
                got6 = h.round(W, LT, bits, key=kx)                    # same shape as the queued job's, a different H
                assert h.desyncs == 1 and not h.queued() and shard.last_stats["bytes_broadcast_LT"] == shard.lt_bytes(d)
                got7 = h.round(W2, LT2, bits, key=k2)
                assert shard.last_stats["bytes_broadcast_LT"] == shard.lt_bytes(d // 2)
                assert torch.equal(got5, _oracle_compute(W, LT, bits, None)) and torch.equal(got6, got5) and torch.equal(got7, got2)
                h.shutdown()
            else:
                assert shard.serve(compute=_oracle_compute) == 7
                got = None
        elif mode == "spmd":
            # every rank runs the same driver loop (scripts/quantize_opt_sharded.py --calibration sharded): no job announcements; the
            # owner rounds through ShardedLDLQ(spmd=True) with its queued LTs, the other rank joins each job with worker_round; then the
            # owner's quantised weights are broadcast so that every rank can re-forward its own calibration samples
            W2, LT2, _ = _fixture(m, d // 2, seed=6)
            lins = [torch.nn.Linear(d, m, bias=False).half(), torch.nn.Linear(d // 2, m, bias=False).half()]
            if rank == 0:
                h = shard.ShardedLDLQ(compute=_oracle_compute, spmd=True)
                H1, H2 = torch.zeros(2, 2), torch.zeros(2, 2)
                h.queue_LTs([(H1, LT), (H2, LT2)])
                got = h.round(W, None, bits, eta=eta, key=shard.h_key(H1))
                got2 = h.round(W2, None, bits, key=shard.h_key(H2))
                assert shard.last_stats["bytes_broadcast_LT"] == 0     # LT2 travelled under job 1
                assert torch.equal(got2, _oracle_compute(W2, LT2, bits, None))
                lins[0].weight.data = got.to(torch.float16)             # what fasterquant leaves behind: a NEW tensor on the owner
                lins[1].weight.data = got2.to(torch.float16)
                h.shutdown()                                            # spmd: nothing is announced
            else:
                ready = shard.worker_round(None, compute=_oracle_compute)
                assert ready is not None                                # the owner prefetched LT2
                assert shard.worker_round(ready, compute=_oracle_compute) is None
                got = None
            nbytes = shard.broadcast_weights(lins)
            assert nbytes == 2 * (m * d + m * (d // 2))
            want_w = _oracle_compute(W, LT, bits, eta).to(torch.float16)
            assert torch.equal(lins[0].weight.data, want_w)            # every rank now holds the owner's quantised weights
            assert torch.equal(lins[1].weight.data, _oracle_compute(W2, LT2, bits, None).to(torch.float16))
        elif mode in ("raw_b", "raw_a"):
            # round 6: the rows travel as the 16-bit tensor they are + the grid parameters; every rank maps its chunk onto the grid
            g = torch.Generator().manual_seed(11)
            W16 = (0.02 * torch.randn(m, d, generator=g)).half()
            maxq = 2 ** bits - 1
            if mode == "raw_b":
                qfn, sc, zr = 'b', torch.tensor([float((2.4 * W16.float().square().mean().sqrt()).half())]), None
            else:
                qfn = 'a'
                lo_, hi_ = W16.float().min(1).values.clamp(max=0), W16.float().max(1).values.clamp(min=0)
                sc = (hi_ - lo_) / maxq
                zr = torch.round(-lo_ / sc)
            raw = (W16, qfn, sc, zr, maxq) if rank == 0 else None
            got = shard.ldlq_round_sharded(None, LT if rank == 0 else None, bits, eta=eta if rank == 0 else None, compute=_oracle_compute,
                                           raw=raw, gridmap=_oracle_gridmap)
            assert (got is None) == (rank != 0)
            if rank == 0:
                assert shard.last_stats["scatter_form"] == "raw16"
                assert shard.last_stats["bytes_scatter"] == (2 + (4 if use_eta else 0)) * shard.row_chunk(m, world) * d * (world - 1)
            W = _oracle_gridmap(W16, qfn, sc, zr, maxq)                   # what the unsharded run rounds
        elif mode == "collective":
            got = shard.ldlq_round_sharded(W if rank == 0 else None, LT if rank == 0 else None, bits,
                                           eta=eta if rank == 0 else None, compute=_oracle_compute)
            assert (got is None) == (rank != 0)
        else:                                      # owner drives, the other rank sits in serve()
            if rank == 0:
                h = shard.ShardedLDLQ(compute=_oracle_compute)
                got = h.round(W, LT, bits, eta=eta)
                got2 = h.round(W[: m // 2], LT, bits, eta=None)       # a second job through the same loop
                h.shutdown()
                assert torch.equal(got2, _oracle_compute(W[: m // 2], LT, bits, None))
            else:
                assert shard.serve(compute=_oracle_compute) == 2
                got = None
        if rank == 0:
            want = _oracle_compute(W, LT, bits, eta)
            torch.save({"equal": bool(torch.equal(got, want)), "shape": tuple(got.shape)}, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("m,d,bits,use_eta,mode", [(96, 128, 2, False, "collective"), (40, 128, 4, True, "collective"),
                                                    (16, 64, 2, False, "collective"), (70, 128, 2, True, "serve"),
                                                    (48, 128, 2, False, "queue"), (64, 128, 2, True, "spmd"),
                                                    (70, 128, 2, False, "raw_b"), (40, 128, 4, True, "raw_a")])
def test_sharded_ldlq_matches_unsharded(tmp_path, m, d, bits, use_eta, mode):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), m, d, bits, use_eta, mode, out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["equal"] and res["shape"] == (m, d)


def test_row_partition_covers_rows_once():
    from quip_amd import shard
    for m in [1, 15, 16, 17, 96, 100, 4096, 11008]:
        for world in [1, 2, 3, 4, 8]:
            parts = shard.row_partition(m, world)
            assert len(parts) == world
            covered = [r for (a, b) in parts for r in range(a, b)]
            assert covered == list(range(m))
            c = shard.row_chunk(m, world)
            assert c % shard.ROW_ALIGN == 0 and c * world >= m
            assert all(b - a <= c for a, b in parts)


def test_world_size_one_is_a_plain_call():
    """no process group needed for the degenerate case through ShardedLDLQ? -- it needs one; the collective with
    world 1 must reduce to compute()."""
    port = _free_port()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from quip_amd import shard
        W, LT, _ = _fixture(32, 64, seed=1)
        got = shard.ldlq_round_sharded(W, LT, 2, compute=_oracle_compute)
        assert torch.equal(got, _oracle_compute(W, LT, 2, None))
        h = shard.ShardedLDLQ(compute=_oracle_compute)
        assert torch.equal(h.round(W, LT, 2), got)
        h.shutdown()
        # force_exchange: the whole broadcast / scatter / gather path with one rank (what a single-GPU box uses to run RCCL)
        hx = shard.ShardedLDLQ(compute=_oracle_compute, force_exchange=True)
        assert torch.equal(hx.round(W, LT, 2), got)
        assert shard.last_stats["world"] == 1 and shard.last_stats["bytes_broadcast_LT"] == shard.lt_bytes(64)
    finally:
        dist.destroy_process_group()


def _hessian_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quip_amd import shard
    from quip_amd.method import QuantMethod
    try:
        torch.manual_seed(0)
        d_in, nsamp = (48, 80), 5
        layers = [torch.nn.Linear(d, 8) for d in d_in]
        X = [torch.randn(nsamp, 12, d).half() for d in d_in]
        qms = [QuantMethod(l) for l in layers]
        a, b = shard.sample_partition(nsamp, world)[rank]
        for qm, x in zip(qms, X):
            for j in range(a, b):
                qm.add_batch(x[j].unsqueeze(0), None)
        follower = QuantMethod(torch.nn.Linear(d_in[0], 4))          # shares the first Linear's input (q/k/v): H is None
        follower.share_hessian_from(qms[0])
        shard.all_reduce_hessians(qms + [follower])
        assert follower.H is None or follower.H is qms[0].H
        for qm in qms:
            qm.post_batch()
        if rank == 0:
            ok = True
            for qm, x, l in zip(qms, X, layers):
                ref = QuantMethod(l)
                for j in range(nsamp):
                    ref.add_batch(x[j].unsqueeze(0), None)
                ref.post_batch()
                ok = ok and qm.nsamples == nsamp and torch.allclose(qm.H, ref.H, rtol=1e-6, atol=0)
            torch.save({"ok": bool(ok)}, out_path)
    finally:
        dist.destroy_process_group()


def test_calibration_samples_split_over_ranks(tmp_path):
    """DP over the calibration samples: partial Hessians all-reduced once per block == the single-process Hessian."""
    from quip_amd import shard
    assert shard.sample_partition(128, 8) == [(16 * r, 16 * r + 16) for r in range(8)]
    assert shard.sample_partition(5, 2) == [(0, 3), (3, 5)] and shard.sample_partition(1, 3) == [(0, 1), (1, 1), (1, 1)]
    out = str(tmp_path / "h.pt")
    mp.spawn(_hessian_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert torch.load(out)["ok"]


def _empty_rank_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quip_amd import shard, method
    from quip_amd.method import QuantMethod
    try:
        assert method.SHARE_IDENTICAL_INPUTS
        torch.manual_seed(0)
        d, nsamp = 48, 2                                               # fewer samples than ranks: the last rank forwards nothing
        q, k, v, o = (torch.nn.Linear(d, 8) for _ in range(4))
        X = torch.randn(nsamp, 12, d).half()
        Xo = torch.randn(nsamp, 12, d).half()
        qms = [QuantMethod(l) for l in (q, k, v, o)]
        a, b = shard.sample_partition(nsamp, world)[rank]
        for j in range(a, b):
            xin = X[j].unsqueeze(0)
            for qm in qms[:3]:                                         # q / k / v are handed THE SAME tensor (HF attention)
                qm.add_batch(xin, None)
            qms[3].add_batch(Xo[j].unsqueeze(0), None)
        if b > a:
            assert qms[1].H is None and qms[2].H is None               # followers of q on the ranks that saw a sample
        else:
            assert all(qm.H is not None for qm in qms)                 # the empty rank knows nothing of the sharing yet
        shard.all_reduce_hessians(qms)
        assert qms[1].H is None and qms[2].H is None                   # ... and has adopted it: 2 collectives everywhere, not 4 vs 2
        for qm in qms:
            qm.post_batch()
        ok = True
        for qm, l, x in zip(qms, (q, k, v, o), (X, X, X, Xo)):
            ref = QuantMethod(l)
            for j in range(nsamp):
                ref.add_batch(x[j].unsqueeze(0).clone(), None)         # (a clone: no sharing in the reference run)
            ref.post_batch()
            ok = ok and qm.nsamples == nsamp and torch.allclose(qm.H, ref.H, rtol=1e-6, atol=0)
        torch.save({"ok": bool(ok)}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_rank_without_samples_joins_the_same_collectives(tmp_path):
    """nsamples < world leaves a rank that never forwarded, hence never learned that q / k / v share one accumulator
    (method.SHARE_IDENTICAL_INPUTS); all_reduce_hessians must still issue the SAME collectives on every rank and give every rank the
    full Hessians."""
    mp.spawn(_empty_rank_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        assert torch.load(str(tmp_path / f"r{r}.pt"))["ok"], f"rank {r}"


# ---- round 4: one owner PER LINEAR (shard.assign_owners / shard.block_owner_per_linear) ------------------------------------------------
_PL_SHAPES = [(32, 64), (32, 64), (48, 64), (32, 128), (64, 32)]          # (rows, columns) of a block's Linears in call order


class _FakeMethod:
    """what block_owner_per_linear needs of a QuantMethod: the layer, rows / columns, an H to key the prepared factor by"""

    def __init__(self, lin):
        self.layer, self.rows, self.columns = lin, lin.weight.shape[0], lin.weight.shape[1]
        self.H = torch.zeros(2, 2)


def _pl_draw(m):
    """the random draws `prepare` makes (numpy's and torch's global streams, like gen_rand_orthos + randperm): returns a per-Linear
    perturbation of the fixture's factor, so a rank that skipped a draw or made it out of order rounds with a different LT"""
    a = float(np.random.normal())
    p = torch.randperm(m.columns)
    return a, p


def _pl_factor(m, j, draw):
    _, LT, _ = _fixture(m.rows, m.columns, seed=20 + j)
    a, p = draw
    return (LT * (1.0 + 0.05 * a) + 1e-3 * torch.triu(p.float()[None, :].expand(m.columns, -1) / m.columns, diagonal=1)).contiguous()


def _pl_reference():
    """the block in ONE process: every Linear prepared in call order on the same streams, rounded by the oracle"""
    np.random.seed(11)
    torch.manual_seed(11)
    out = []
    lins = [torch.nn.Linear(c, r, bias=False).half() for r, c in _PL_SHAPES]      # (their initialisers draw from torch's stream: as in the workers)
    for j, (r, c) in enumerate(_PL_SHAPES):
        m = _FakeMethod(lins[j])
        W, _, _ = _fixture(r, c, seed=20 + j)
        codes = _oracle_compute(W, _pl_factor(m, j, _pl_draw(m)), 2, None)
        out.append(codes)
    return out


def _pl_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quip_amd import shard
    try:
        np.random.seed(11)                                               # every rank seeds alike (the drivers do)
        torch.manual_seed(11)
        lins = [torch.nn.Linear(c, r, bias=False).half() for r, c in _PL_SHAPES]
        methods = [_FakeMethod(l) for l in lins]
        idx = {id(m): j for j, m in enumerate(methods)}
        owners = shard.assign_owners(_PL_SHAPES, world)
        prepared_by, srcs = [], []

        def prepare(m):
            j = idx[id(m)]
            prepared_by.append(j)
            return m.H, _pl_factor(m, j, _pl_draw(m))

        def skip(m):
            _pl_draw(m)

        def finish(m):
            j = idx[id(m)]
            W, _, _ = _fixture(m.rows, m.columns, seed=20 + j)
            codes = shard.active().round(W, None, 2, key=shard.h_key(m.H))      # what quantize_weight_vecbal does with a queued H
            srcs.append(shard.last_stats["world"])
            m.layer.weight.data = codes.to(torch.float16)                # fasterquant leaves a NEW tensor on the owner
            return float(codes.double().sum())
        timers = {}
        errs = shard.block_owner_per_linear(methods, lins, owners, prepare, skip, finish, compute=_oracle_compute, timers=timers)
        assert prepared_by == [j for j in range(len(methods)) if owners[j] == rank]            # own Linears only, in call order
        assert shard.active() is None and set(timers) >= {"owner_preproc_factor_s", "broadcast_LT_s", "round_s", "broadcast_weights_s"}
        torch.save({"owners": owners, "errs": errs, "weights": [l.weight.data.clone() for l in lins]}, f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_one_owner_per_linear_reproduces_the_single_owner_block(tmp_path, world):
    """every Linear prepared on its own owner in parallel, LTs broadcast from their owners, rows scattered from / gathered at owner(j), weights
    and errors to everyone: on EVERY rank the block's weights and per-Linear errors equal the one-process run -- which also proves that
    the ranks consumed the random streams in step (a skipped or reordered draw changes a factor)"""
    from quip_amd import shard
    out = str(tmp_path / "pl.pt")
    mp.spawn(_pl_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    want = _pl_reference()
    owners = shard.assign_owners(_PL_SHAPES, world)
    assert len(set(owners)) == min(world, len(_PL_SHAPES))               # the work really is spread
    for rank in range(world):
        res = torch.load(f"{out}.{rank}")
        assert res["owners"] == owners
        for j, codes in enumerate(want):
            assert torch.equal(res["weights"][j], codes.to(torch.float16)), (rank, j)
            assert res["errs"][j] == float(codes.double().sum())


def test_assign_owners_balances_by_cost():
    from quip_amd import shard
    opt = [(2048, 2048)] * 4 + [(8192, 2048), (2048, 8192)]             # OPT-1.3B block: fc2's factor (d = 8192) dwarfs the rest
    o2 = shard.assign_owners(opt, 2)
    assert o2[5] != o2[4] and all(o == o2[4] for o in o2[:5])           # fc2 alone, everything else on the other rank
    o8 = shard.assign_owners(opt, 8)
    assert sorted(o8) == list(range(6))                                 # six Linears, six different owners
    assert shard.assign_owners(opt, 1) == [0] * 6
    llama = [(4096, 4096)] * 4 + [(11008, 4096)] * 2 + [(4096, 11008)]
    o3 = shard.assign_owners(llama, 3)
    assert len(set(o3)) == 3 and o3 == shard.assign_owners(llama, 3)    # deterministic: every rank computes the same table
