"""Row-sharded LDLQ across the GPUs of one node (BASELINE.json north_star: "per-layer LDLQ is
embarrassingly parallel and is sharded across the 8 GPUs of one node with RCCL over xGMI for the layer
scatter only").

What shards and why
  * inside one Linear the rows of W are independent given L (vector_balance.py:179-180 is row-wise), and the
    LDLQ cost is proportional to m*d^2, so the unit of work is a ROW CHUNK of one Linear -- whole-Linear
    assignment would cap the speed-up at 1.5x because fc2 is 2/3 of an OPT block;
  * everything that couples rows (the qfn-b scalar scale = global RMS, vector_balance.py:522; U*W row mixing,
    method.py:175; the Cholesky of H) is done by the owner BEFORE the scatter, so the exchange is
        broadcast  LT      strict upper triangle of [d, d] fp32, as 8 row slabs cut at their first column: 0.56 d^2 words
                                          (owner -> all; round 6: the lower half is zero by construction and no longer travels)
        scatter    W       [m/k, d]      the layer's own 16-bit rows + the grid parameters; every rank runs the grid map (K5) itself
                                          (round 6; fp32 grid coordinates, 4 B per weight, only when W is not a 16-bit tensor)
        (scatter   eta     [m/k, d]      fp32   only with --unbiased)
        gather     codes   [m/k, d]      2/4-bit STREAM-packed words (uint8 on the gloo test path)  (rank r -> owner)
    and no all-reduce exists on the path.  xGMI is point-to-point, so scatter/gather from the owner run over
    all 7 links at once; only the LT broadcast is ring/tree-shaped.
  * blocks of the transformer stay sequential (opt.py:172-181), so the owner keeps the model and the block
    forward; the other ranks sit in `serve()`.

Process model: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" in the CPU
tests, where the per-chunk kernel is injected because the HIP kernels need a GPU).

Collective entry points
  ldlq_round_sharded(wgrid, LT, bits, eta)   called by EVERY rank (non-owners pass None tensors)
  serve() / shutdown()                        worker loop for ranks that do not run the driver: the owner's
                                              ShardedLDLQ.round(...) announces each job with a header broadcast
"""
import torch
import torch.distributed as dist

ROW_ALIGN = 16          # the K4 workgroup owns 16 rows; STREAM tiles are 16 rows
_OP_STOP, _OP_LDLQ = 0, 1


def row_chunk(m, world, align=ROW_ALIGN):
    """rows per rank: equal chunks (RCCL scatter/gather want equal counts), multiple of `align`, covering m."""
    per = -(-m // world)
    return -(-per // align) * align


def row_partition(m, world, align=ROW_ALIGN):
    """[(start, stop)] of the real (unpadded) rows each rank owns; ranks past the end own nothing."""
    c = row_chunk(m, world, align)
    return [(min(r * c, m), min((r + 1) * c, m)) for r in range(world)]


def _comm_device(group=None):
    """tensors handed to the collectives must live where the backend works: HIP memory for RCCL, host for gloo."""
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


LT_SLABS = 8            # row slabs the LT factor travels in (each cut at its first column)


def lt_slabs(d):
    """[(r0, r1)] row slabs of LT = strict-upper(L^T) [d, d]; slab rows r0..r1 are zero left of column r0, so only columns r0..d travel"""
    if d < 16 * LT_SLABS:
        return [(0, d)]
    b = -(-d // LT_SLABS)
    return [(r0, min(r0 + b, d)) for r0 in range(0, d, b)]


def lt_wire_numel(d):
    return sum((r1 - r0) * (d - r0) for r0, r1 in lt_slabs(d))


def lt_bytes(d):
    """bytes one LT broadcast moves per receiving rank (0.5625 d^2 fp32 words with 8 slabs instead of d^2)"""
    return 4 * lt_wire_numel(d)


class _TriBroadcast:
    """work handle of an LT broadcast in slab form: wait() lets the flat buffer land and, on the receiving ranks, cuts it back into the
    zero-initialised [d, d] tensor (stream-ordered on RCCL: the unpack is queued behind the collective)."""

    def __init__(self, flat, work, LT, unpack):
        self.flat, self.work, self.LT, self.unpack = flat, work, LT, unpack

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.unpack:
            d, off = self.LT.shape[0], 0
            for r0, r1 in lt_slabs(d):
                n = (r1 - r0) * (d - r0)
                self.LT[r0:r1, r0:] = self.flat[off:off + n].view(r1 - r0, d - r0)
                off += n
            self.unpack = False
        self.flat = None


def broadcast_LT(LT, d, src, group, dev, async_op=False):
    """LT [d, d] fp32 from `src` to every rank of `group`, upper slabs only.  Returns (LT on the comm device, handle | None): the tensor is
    valid after handle.wait() (already waited when async_op is False).  On `src` LT is the factor; elsewhere it is ignored."""
    rank = dist.get_rank(group)
    slabs = lt_slabs(d)
    if rank == src:
        LT = LT.to(dev, torch.float32).contiguous()
        flat = LT.reshape(-1) if len(slabs) == 1 else torch.cat([LT[r0:r1, r0:].reshape(-1) for r0, r1 in slabs])
    else:
        LT = torch.zeros(d, d, dtype=torch.float32, device=dev) if len(slabs) > 1 else torch.empty(d, d, dtype=torch.float32, device=dev)
        flat = LT.reshape(-1) if len(slabs) == 1 else torch.empty(lt_wire_numel(d), dtype=torch.float32, device=dev)
    work = dist.broadcast(flat, src=src, group=group, async_op=async_op)
    h = _TriBroadcast(flat, work if async_op else None, LT, unpack=(rank != src and len(slabs) > 1))
    if not async_op:
        h.wait()
        return LT, None
    return LT, h


def _host_backend(group=None):
    return dist.is_initialized() and dist.get_backend(group) != "nccl"


def _default_compute(wgrid, LT, bits, eta, group=None):
    from . import ops                          # HIP kernel K4; raises on CPU tensors -- no fallback
    if not wgrid.is_cuda and torch.cuda.is_available() and _host_backend(group):
        # a host-memory backend (gloo) delivered the chunk on the CPU: stage it to this rank's GPU for the kernel and hand the codes back
        # where the exchange expects them (tests/test_gpu_shard_two_ranks.py: several ranks sharing one GPU).  Still the HIP kernel.
        dev = torch.device("cuda", torch.cuda.current_device())
        out = ops.ldlq_round(wgrid.to(dev), LT.to(dev), bits, eta=None if eta is None else eta.to(dev))
        return out.cpu()
    return ops.ldlq_round(wgrid, LT, bits, eta=eta)


def _default_gridmap(w, qfn, scale, zero, maxq, group=None):
    """grid coordinates of a row chunk on the rank that rounds it (K5, csrc/gridmap.hip; vector_balance.py:515, 522-524)"""
    from . import ops
    if not w.is_cuda and torch.cuda.is_available() and _host_backend(group):        # a host-memory backend delivered the rows on the CPU
        dev = torch.device("cuda", torch.cuda.current_device())
        return ops.gridmap(w.to(dev), qfn, scale.to(dev), None if zero is None else zero.to(dev), maxq).cpu()
    return ops.gridmap(w, qfn, scale, zero, maxq)


_RAW_DTYPES = {torch.float16: 1, torch.bfloat16: 2}
_RAW_CODES = {v: k for k, v in _RAW_DTYPES.items()}


def _pack_chunk(codes, bits):
    from . import ops
    return ops.pack(codes, bits, ops.LAYOUT_STREAM)


def _unpack_all(words, bits, m, d):
    from . import ops
    return ops.unpack(words, bits, ops.LAYOUT_STREAM, m, d)


last_stats = {}      # owner side, filled by every ldlq_round_sharded call: bytes per phase (and seconds when timing is on)
TIMING = False       # True: synchronise around every phase and record wall-clock seconds in last_stats (benchmarks only)


def _tick(dev):
    if TIMING:
        if dev.type == "cuda":
            torch.cuda.synchronize()
        import time
        return time.perf_counter()
    return 0.0


def ldlq_round_sharded(wgrid, LT, bits, eta=None, src=0, group=None, compute=None, gather_packed=None, force_exchange=False,
                       lt_ready=None, next_LT=None, raw=None, gridmap=None):
    """LDLQ codes of one Linear with its rows split over the ranks of `group`.
    gather_packed: return the codes to the owner as STREAM-packed words (bits/8 bytes per code instead of 1; the
    STREAM layout is row-tile-major, so the per-rank chunks concatenate into the whole matrix's packing).  Default:
    on for the RCCL backend when the shape packs (bits in {2,4}, d a multiple of 512/bits), off otherwise.

    Collective: every rank calls it.  On `src`: wgrid float32 [m,d] grid coordinates, LT float32 [d,d]
    (ops.unit_lower_t of the Cholesky factor), eta float32 [m,d] or None; returns codes uint8 [m,d].
    On the other ranks the tensor arguments are ignored (pass None) and None is returned.
    raw (round 6, instead of wgrid): (W [m,d] fp16 | bf16 -- the layer's own rows as vector_balance.py:513-524 sees them --, qfn 'a' | 'b',
    scale fp32 [1] | [m], zero fp32 [m] | None, maxq): the rows travel in their 16-bit dtype (half the bytes of the fp32 grid coordinates)
    and EVERY rank runs the grid map on its chunk (`gridmap(w, qfn, scale, zero, maxq) -> fp32`, default K5): the map is elementwise
    given the grid parameters, so the chunk's coordinates are bit for bit the rows of the owner's full map.
    `compute(wgrid_chunk, LT, bits, eta_chunk) -> uint8 codes` defaults to the HIP kernel.
    force_exchange: run broadcast / scatter / gather (and the HIP pack / unpack around the gather) also with ONE rank -- how a
    single-GPU box exercises the RCCL path.
    lt_ready: (LT tensor on the comm device, work handle | None) prefetched by a previous call's `next_LT` on every rank: the
    LT broadcast of this job is skipped.  next_LT (owner: float32 [d2, d2], others: anything non-None the header says): the
    NEXT job's LT is broadcast between this job's scatter and its compute, i.e. it travels while every rank rounds
    (SURVEY.md 8(e)); returned as the second element of the result tuple (codes | None, (LT_next, work))."""
    if compute is None:
        def compute(wg, lt, b, e):
            return _default_compute(wg, lt, b, e, group=group)
    if gridmap is None:
        def gridmap(w, qfn, sc, zr, mq):
            return _default_gridmap(w, qfn, sc, zr, mq, group=group)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _comm_device(group)
    hdr = torch.zeros(9, dtype=torch.int64, device=dev)
    if rank == src:
        shape = raw[0].shape if raw is not None else wgrid.shape
        assert len(shape) == 2 and LT.shape == (shape[1], shape[1])
        rawdt, qfnc, maxq = 0, 0, 0
        if raw is not None:
            assert wgrid is None and raw[0].dtype in _RAW_DTYPES and raw[1] in ('a', 'b')
            rawdt, qfnc, maxq = _RAW_DTYPES[raw[0].dtype], int(raw[1] == 'b'), int(raw[4])
        hdr = torch.tensor([shape[0], shape[1], int(bits), int(eta is not None), int(lt_ready is not None),
                            0 if next_LT is None else next_LT.shape[0], rawdt, qfnc, maxq], dtype=torch.int64, device=dev)
    if world > 1 or force_exchange:
        dist.broadcast(hdr, src=src, group=group)
    m, d, bits, has_eta, lt_pref, d_next, rawdt, qfnc, maxq = (int(v) for v in hdr.tolist())
    qfn = 'b' if qfnc else 'a'
    if world == 1 and not force_exchange:
        if raw is not None:
            wgrid = gridmap(raw[0], raw[1], raw[2], raw[3], raw[4])
        out = compute(wgrid, LT, bits, eta)
        return (out, None) if next_LT is not None else out
    if gather_packed is None:
        gather_packed = dist.get_backend(group) == "nccl" and bits in (2, 4) and d % (512 // bits) == 0

    c = row_chunk(m, world)
    pad = c * world - m

    def padded_chunks(t, dtype=torch.float32):
        t = t.to(dev, dtype)
        if pad:
            t = torch.cat([t, torch.zeros((pad,) + tuple(t.shape[1:]), dtype=dtype, device=dev)], 0)
        return list(t.split(c, 0))

    stats = {"world": world, "m": m, "d": d, "bytes_broadcast_LT": 0, "bytes_scatter": 0, "bytes_gather": 0, "scatter_form": "raw16" if rawdt else "grid32"}
    t0 = _tick(dev)
    if lt_pref:                                                       # prefetched under the previous job's rounding
        assert lt_ready is not None, "the owner announced a prefetched LT this rank does not hold"
        LT, work = lt_ready
        if work is not None:
            work.wait()
        assert LT.shape == (d, d)
    else:
        if lt_ready is not None and lt_ready[1] is not None:          # the owner dropped its queue (desync): drain, forget
            lt_ready[1].wait()
        LT, _ = broadcast_LT(LT, d, src, group, dev)                   # the one tree/ring-shaped transfer
        stats["bytes_broadcast_LT"] = lt_bytes(d)
    t1 = _tick(dev)

    sc_mine = zr_mine = None
    if rawdt:                                                         # 16-bit rows (as bytes: gloo and RCCL both move uint8) + the grid parameters
        mine8 = torch.empty(c, 2 * d, dtype=torch.uint8, device=dev)
        dist.scatter(mine8, padded_chunks(raw[0].contiguous().view(torch.uint8), torch.uint8) if rank == src else None, src=src, group=group)
        mine = mine8.view(_RAW_CODES[rawdt])
        if qfn == 'b':                                                # one scalar for the whole matrix (vector_balance.py:522)
            sc_mine = raw[2].to(dev, torch.float32).reshape(1).clone() if rank == src else torch.empty(1, dtype=torch.float32, device=dev)
            dist.broadcast(sc_mine, src=src, group=group)
        else:                                                         # per-row (scale, zero): [c, 2]
            sz = torch.empty(c, 2, dtype=torch.float32, device=dev)
            parts = None
            if rank == src:
                parts = padded_chunks(torch.stack([raw[2].reshape(-1).float().expand(m) if raw[2].numel() == 1 else raw[2].reshape(-1).float(),
                                                   raw[3].reshape(-1).float().expand(m) if raw[3].numel() == 1 else raw[3].reshape(-1).float()], 1))
                parts = [p_.contiguous() for p_ in parts]
                if pad:
                    parts[-1][c - pad:, 0] = 1.0                      # padded rows: a finite grid (never rounded, never returned)
            dist.scatter(sz, parts, src=src, group=group)
            sc_mine, zr_mine = sz[:, 0].contiguous(), sz[:, 1].contiguous()
        stats["bytes_scatter"] = 2 * c * d * (world - 1)
    else:
        mine = torch.empty(c, d, dtype=torch.float32, device=dev)
        dist.scatter(mine, padded_chunks(wgrid) if rank == src else None, src=src, group=group)
        stats["bytes_scatter"] = 4 * c * d * (world - 1)
    eta_mine = None
    if has_eta:
        eta_mine = torch.empty(c, d, dtype=torch.float32, device=dev)
        dist.scatter(eta_mine, padded_chunks(eta) if rank == src else None, src=src, group=group)
        stats["bytes_scatter"] += 4 * c * d * (world - 1)
    t2 = _tick(dev)

    nxt = None
    if d_next:                                                        # the next job's LT rides under this job's rounding
        nxt = broadcast_LT(next_LT if rank == src else None, d_next, src, group, dev, async_op=True)
        stats["bytes_broadcast_next_LT"] = lt_bytes(d_next)

    lo, hi = row_partition(m, world)[rank]
    n_real = hi - lo
    codes = torch.zeros(c, d, dtype=torch.uint8, device=dev)
    if n_real > 0:                                                    # padded rows are never rounded
        if rawdt:
            wg = gridmap(mine[:n_real], qfn, sc_mine if qfn == 'b' else sc_mine[:n_real], None if zr_mine is None else zr_mine[:n_real], maxq)
        else:
            wg = mine[:n_real]
        codes[:n_real] = compute(wg, LT, bits, None if eta_mine is None else eta_mine[:n_real])
    t3 = _tick(dev)

    ref = raw[0] if raw is not None else wgrid

    def done(out):
        if rank == src:
            t4 = _tick(dev)
            stats["bytes_gather"] = (c * d * bits // 8 if gather_packed else c * d) * (world - 1)
            if TIMING:
                stats.update({"s_broadcast_LT": t1 - t0, "s_scatter": t2 - t1, "s_round": t3 - t2, "s_gather": t4 - t3})
            last_stats.clear()
            last_stats.update(stats)
        if out is not None and ref is not None and out.device != ref.device:
            out = out.to(ref.device)                                   # a host-memory backend gathered on the CPU: back to where the caller's tensors live
        return (out, nxt) if d_next else out

    if gather_packed:
        words = _pack_chunk(codes, bits)                               # c*d*bits/32 int32 words, whole 16-row tiles
        parts = [torch.empty_like(words) for _ in range(world)] if rank == src else None
        dist.gather(words, parts, dst=src, group=group)
        if rank != src:
            return done(None)
        return done(_unpack_all(torch.cat(parts, 0), bits, c * world, d)[:m].contiguous())
    parts = [torch.empty(c, d, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == src else None
    dist.gather(codes, parts, dst=src, group=group)
    if rank != src:
        return done(None)
    return done(torch.cat(parts, 0)[:m].contiguous())


class ShardedLDLQ:
    """Owner-side handle used by vector_balance.quantize_weight_vecbal when a process group with more than one
    rank is active: announces a job to the ranks parked in serve(), then joins the collective itself."""

    def __init__(self, group=None, src=0, compute=None, force_exchange=False, spmd=False, gridmap=None):
        """spmd=True: every rank runs the same driver loop and joins each collective by itself (worker_round): no job announcements"""
        self.group, self.src, self.compute, self.force_exchange, self.spmd = group, src, compute, force_exchange, spmd
        self.gridmap = gridmap     # tests on CPU ranks inject the oracle's grid map next to its rounding kernel
        self._lt_ready = None      # (key, (LT, work)): the coming job's LT, already broadcast under the previous job's rounding
        self._queue = []           # [(key, LT, H)] of the coming round() calls, in call order (queue_LTs); H held so that its address stays its own
        self.desyncs = 0           # how often a round() call did not match the head of the queue (the queue is dropped then)

    def _announce(self, op):
        dev = _comm_device(self.group)
        dist.broadcast(torch.tensor([op], dtype=torch.int64, device=dev), src=self.src, group=self.group)

    def queue_LTs(self, items):
        """The driver knows the LT factors of the next Linears it will round (all Hessians of a transformer block exist
        before its first Linear is rounded, opt.py:141-150): queue them in call order as (H, LT) pairs -- H the very tensor
        the rounding will be called with (QuantMethod.H after preproc), LT = its transposed unit LDL factor.  round() takes
        its LT from the queue only when the H it is asked to round IS the queued one (h_key) and broadcasts the FOLLOWING
        entry's LT while every rank rounds the current one.  A call that does not match the head of the queue
        (ldlqRG's permuted copy of H, a Linear that took the greedy-pass path, a skipped layer) drops the queue and the
        prefetched LT and factors / broadcasts its own H: slower, never wrong."""
        assert not self._queue and self._lt_ready is None, "queue_LTs: the previous block's queue was not consumed"
        # the entry KEEPS H alive: as long as it is queued its storage cannot be freed and handed to another tensor of the same shape
        # (ldlqRG's permuted copy ...), so the (address, shape) key cannot match a different Hessian
        self._queue = [(h_key(H), LT, H) for H, LT in items]

    def preload(self, H, LT, ready):
        """ONE coming round() call, for the Hessian `H`, whose LT every rank already holds: `ready` = (LT on the comm device, work | None)
        from a broadcast the caller issued itself (block_owner_per_linear: all LTs of a block travel as soon as their owners have
        factored them).  The job's header then says "prefetched" and no LT is broadcast inside it."""
        assert not self._queue and self._lt_ready is None, "preload: the previous queue was not consumed"
        k = h_key(H)
        self._queue = [(k, LT, H)]
        self._lt_ready = (k, ready)

    def queued(self, key=None):
        """is the next round() call served from the queue?  key = h_key(H) of the H about to be rounded"""
        return bool(self._queue) and (key is None or self._queue[0][0] == key)

    def round(self, wgrid, LT, bits, eta=None, key=None, raw=None):
        """raw: see ldlq_round_sharded -- the rows in their 16-bit dtype plus the grid parameters instead of fp32 grid coordinates"""
        if dist.get_world_size(self.group) > 1 and not self.spmd:
            self._announce(_OP_LDLQ)
        ready = None
        if self._queue and self._queue[0][0] == key and key is not None:
            _, LT, _ = self._queue.pop(0)
            if self._lt_ready is not None and self._lt_ready[0] == key:
                ready = self._lt_ready[1]
        elif self._queue or self._lt_ready is not None:              # not what the queue describes: never round with its factor
            self.desyncs += 1
            self._queue = []
        stale, self._lt_ready = (self._lt_ready if ready is None else None), None
        if stale is not None and stale[1][1] is not None:
            stale[1][1].wait()                                        # a prefetch nobody will use: let it land, then forget it
        assert LT is not None, "round(): no LT given and none queued for this H"
        nxt_key, nxt, _ = self._queue[0] if self._queue else (None, None, None)
        out = ldlq_round_sharded(wgrid, LT, bits, eta=eta, src=self.src, group=self.group, compute=self.compute,
                                 force_exchange=self.force_exchange, lt_ready=ready, next_LT=nxt, raw=raw, gridmap=self.gridmap)
        if nxt is not None:
            out, pre = out
            self._lt_ready = (nxt_key, pre) if pre is not None else None
        return out

    def shutdown(self):
        if dist.get_world_size(self.group) > 1 and not self.spmd:
            self._announce(_OP_STOP)


def serve(group=None, src=0, compute=None, gridmap=None):
    """Worker loop for every rank except the owner: wait for a job header, join the collective, repeat until the
    owner calls ShardedLDLQ.shutdown().  Returns the number of jobs served."""
    dev = _comm_device(group)
    jobs, ready = 0, None
    while True:
        op = torch.zeros(1, dtype=torch.int64, device=dev)
        dist.broadcast(op, src=src, group=group)
        if int(op.item()) == _OP_STOP:
            return jobs
        out = ldlq_round_sharded(None, None, 0, src=src, group=group, compute=compute, lt_ready=ready, next_LT=True, gridmap=gridmap)
        ready = out[1] if isinstance(out, tuple) else None           # a prefetched LT for the next job, if the owner sent one
        jobs += 1


def worker_round(ready=None, group=None, src=0, compute=None, gridmap=None):
    """SPMD counterpart of ShardedLDLQ.round on every rank but the owner: join ONE rounding job (the header broadcast by the owner says
    what it is).  `ready`: what the previous call returned -- the LT the owner prefetched for this job, or None.  Returns the `ready`
    for the next call."""
    out = ldlq_round_sharded(None, None, 0, src=src, group=group, compute=compute, lt_ready=ready, next_LT=True, gridmap=gridmap)
    return out[1] if isinstance(out, tuple) else None


def broadcast_weights(layers, group=None, src=0):
    """the quantised weights of a block's Linears from the owner to every rank (each rank re-forwards its own calibration samples
    through the quantised block: opt.py:172-174 on N GPUs).  `layers`: nn.Linear modules in the same order on every rank; on the
    owner `weight.data` is what fasterquant left there (a NEW fp16 tensor on the Balance path), elsewhere the stale copy it replaces.
    Returns the bytes sent per receiving rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    dev = _comm_device(group)
    nbytes = 0
    for lin in layers:
        w = lin.weight.data
        buf = w if (w.device == dev and w.is_contiguous()) else w.to(dev).contiguous()
        dist.broadcast(buf, src=src, group=group)
        if buf is not w:
            lin.weight.data.copy_(buf)
        nbytes += buf.numel() * buf.element_size()
    return nbytes


# ------------------------------------------------------------------------------------- one owner PER LINEAR
def assign_owners(shapes, world):
    """owner rank of every Linear of a block: what couples the rows of a Linear -- post_batch, preproc (rescale, the projection of W and
    of H), the LDL factor -- runs on ONE rank per Linear, different Linears on different ranks at the same time (round 3 ran all of
    them on rank 0: 26 % of a block, capping 8 GPUs at 2.3x).  shapes: [(rows, columns)] in call order.  Longest-processing-time-first
    on the cost model  columns^3 / 3 (factor) + 2 columns^2 (p + q) (V H V^T) + 2 rows columns (p + q) (U W V^T)  with p + q ~ 2 sqrt(n);
    ties go to the lowest rank, so every rank computes the same assignment from the shapes alone."""
    def cost(r, c):
        return c ** 3 / 3.0 + 4.0 * c * c * (c ** 0.5) + 2.0 * r * c * ((r ** 0.5) + (c ** 0.5))
    order = sorted(range(len(shapes)), key=lambda i: (-cost(*shapes[i]), i))
    load = [0.0] * world
    owners = [0] * len(shapes)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owners[i] = r
        load[r] += cost(*shapes[i])
    return owners


def block_owner_per_linear(methods, layers, owners, prepare, skip, finish, group=None, compute=None, force_exchange=False, timers=None,
                           gridmap=None):
    """One transformer block's Linears quantised with ONE OWNER PER LINEAR (SPMD: every rank calls this with the same lists).

      methods   the block's QuantMethod objects in the driver's call order (opt.py:147-170), Hessians already all-reduced
      layers    their nn.Linear modules (same order); on return every rank holds every owner's quantised weights
      owners    assign_owners(...) -- owner rank per Linear
      prepare(m) -> (H, LT)   on the OWNER: post_batch + preproc, then the transposed unit LDL factor of the preprocessed H
      skip(m)                 on every other rank: consume exactly the random draws prepare(m) makes (QuantMethod.skip_operators), so
                              that the operators of the Linears this rank DOES own are the ones a single-owner run would have drawn
      finish(m) -> float      on the OWNER: fasterquant (its rounding is routed through shard.active()); returns the proxy error

    Phases: (A) every rank walks the Linears in order, preparing its own and skipping the others -- the owners work in parallel;
    (B) every LT is broadcast from its owner (asynchronously, back to back: they travel while later owners still factor and while
    earlier Linears round); (C) per Linear, in order: rows scattered from ITS owner, K4 on every rank, codes gathered there, the owner
    finishes (postproc, error); (D) the owner's fp16 weights to everyone; (E) the errors to everyone.  No collective moves arithmetic:
    broadcast / scatter / gather only, as the north_star asks.  Returns [error per Linear]."""
    import time
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _comm_device(group)
    exchange = world > 1 or force_exchange
    n = len(methods)

    def tick():
        if timers is None:
            return 0.0
        if dev.type == "cuda":
            torch.cuda.synchronize()
        return time.perf_counter()
    t0 = tick()
    prepared = {}
    for j, m in enumerate(methods):                                    # (A)
        if owners[j] == rank:
            prepared[j] = prepare(m)
        else:
            skip(m)
    t1 = tick()
    ready = [None] * n
    if exchange:                                                       # (B)
        dims = torch.zeros(n, dtype=torch.int64, device=dev)
        for j, (H, LT) in prepared.items():
            dims[j] = LT.shape[0]
        if world > 1:
            dist.all_reduce(dims, group=group)                         # every rank learns every factor's size (one entry per owner)
        for j in range(n):
            ready[j] = broadcast_LT(prepared[j][1] if owners[j] == rank else None, int(dims[j]), owners[j], group, dev, async_op=True)
            if timers is not None:
                timers["bytes_broadcast_LT"] = timers.get("bytes_broadcast_LT", 0) + lt_bytes(int(dims[j]))
    t2 = tick()
    errors = torch.zeros(n, dtype=torch.float64, device=dev)
    prev = active()
    try:
        for j, m in enumerate(methods):                                # (C)
            if owners[j] == rank:
                handle = ShardedLDLQ(group=group, src=rank, compute=compute, force_exchange=force_exchange, spmd=True, gridmap=gridmap)
                H, LT = prepared.pop(j)
                if exchange:
                    handle.preload(H, LT, ready[j])
                else:
                    handle.queue_LTs([(H, LT)])
                activate(handle)
                errors[j] = float(finish(m))
                activate(None)
                assert handle.desyncs == 0 and not handle.queued(), "block_owner_per_linear: finish() did not round the prepared Hessian"
            elif exchange:
                worker_round(ready[j], group=group, src=owners[j], compute=compute, gridmap=gridmap)
            ready[j] = None
    finally:
        activate(prev)
    t3 = tick()
    nbytes = 0
    if world > 1:                                                      # (D), (E)
        for j, lin in enumerate(layers):
            nbytes += broadcast_weights([lin], group=group, src=owners[j])
        dist.all_reduce(errors, group=group)
    t4 = tick()
    if timers is not None:
        for k, v in (("owner_preproc_factor_s", t1 - t0), ("broadcast_LT_s", t2 - t1), ("round_s", t3 - t2), ("broadcast_weights_s", t4 - t3)):
            timers[k] = timers.get(k, 0.0) + v
        timers["bytes_broadcast_weights"] = timers.get("bytes_broadcast_weights", 0) + nbytes
    return [float(e) for e in errors.tolist()]


# ------------------------------------------------------------------------------------- calibration-sample sharding
def sample_partition(nsamples, world):
    """[(start, stop)] of the calibration samples each rank feeds through add_batch (contiguous, sizes differ by <= 1)."""
    base, rem = divmod(nsamples, world)
    out, a = [], 0
    for r in range(world):
        b = a + base + (1 if r < rem else 0)
        out.append((a, b))
        a = b
    return out


def all_reduce_hessians(methods, group=None):
    """SURVEY.md 8(e) "alternative/extra": the 128 calibration samples of the H pass (opt.py:141-143) split over the
    ranks -- every rank runs the block forward on ITS samples (sample_partition) and accumulates its partial
    fp64 X^T X per Linear (QuantMethod.add_batch -> K7), then ONE exchange step per block: a SUM all-reduce of each
    Linear's accumulator (fp64, d x d; direct reduce-scatter + all-gather over the 7 xGMI links inside RCCL) and of
    the sample counts.  Call between the last add_batch and post_batch, with the same list order on every rank.
    The K7 accumulator holds the block-lower triangle only (the rest is zero on every rank), so the sum of the
    partials is again a valid accumulator; summing partials changes the fp64 summation order only."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dev = _comm_device(group)
    counts = torch.tensor([float(m.nsamples) for m in methods], dtype=torch.float64, device=dev)
    dist.all_reduce(counts, group=group)
    tri = torch.tensor([1.0 if getattr(m, "_tri", False) else 0.0 for m in methods], dtype=torch.float64, device=dev)
    dist.all_reduce(tri, op=dist.ReduceOp.MAX, group=group)
    # Who skips its collective is decided BY ALL RANKS TOGETHER.  method.SHARE_IDENTICAL_INPUTS turns q / k / v into followers inside
    # add_batch -- on the ranks that forwarded a sample.  A rank whose share of the calibration set is empty (nsamples < world) never
    # gets there and would issue one all-reduce per Linear while the others issue one per leader: the calls would pair different
    # Linears' Hessians.  So: every rank reports the leader (as a position in `methods`) of each method, MAX over ranks, and a rank
    # that has not seen the sharing adopts it before the Hessians move.
    # Every failure below is decided COLLECTIVELY: a rank that raised alone would leave its peers inside the H all-reduces.
    # Each rank computes its error code, the codes are MAX-reduced, and all ranks raise the same exception together.
    def _leader_of(m):
        lead = getattr(m, "_leader", None) or getattr(m, "_auto_leader", None)
        if lead is None:
            return -1.0
        for k, o in enumerate(methods):
            if o is lead:
                return float(k)
        return -2.0                                                     # a leader outside the list: reported below, by every rank
    mine = [_leader_of(m) for m in methods]
    lead_ix = torch.tensor([max(v, -1.0) for v in mine], dtype=torch.float64, device=dev)
    dist.all_reduce(lead_ix, op=dist.ReduceOp.MAX, group=group)
    err = 1.0 if any(v == -2.0 for v in mine) else 0.0
    for m, k in zip(methods, lead_ix.tolist()):
        if k >= 0 and m.H is not None and (m.nsamples != 0 or getattr(m, "_followers", 0) > 0):
            err = max(err, 2.0)
    errt = torch.tensor([err], dtype=torch.float64, device=dev)
    dist.all_reduce(errt, op=dist.ReduceOp.MAX, group=group)
    if float(errt.item()) == 1.0:
        raise RuntimeError("all_reduce_hessians: on some rank a method shares the Hessian of a leader that is not in the list")
    if float(errt.item()) == 2.0:
        raise RuntimeError("all_reduce_hessians: the ranks disagree on which Linears share a Hessian, and a rank has already accumulated "
                           "samples into its own (set method.SHARE_IDENTICAL_INPUTS = False)")
    for m, k in zip(methods, lead_ix.tolist()):
        if k >= 0 and m.H is not None:
            m.share_hessian_from(methods[int(k)])
    for m, n, t in zip(methods, counts.tolist(), tri.tolist()):
        if m.H is None:                     # a follower of QuantMethod.share_hessian_from (q/k/v share one accumulator): its
            m.nsamples = int(round(n))      # leader's H is reduced once; only the sample count is per method
            continue
        assert m.H.dtype == torch.float64, "all_reduce_hessians: call before post_batch"
        H = m.H if m.H.device == dev else m.H.to(dev)
        dist.all_reduce(H, group=group)
        if H is not m.H:
            m.H.copy_(H)
        m.nsamples = int(round(n))
        if t:                         # a rank that saw no sample still has to mirror the triangle in post_batch
            m._tri = True


def h_key(H):
    """identity of the Hessian a queued LT was factored from: the tensor's storage address and shape (unique while the tensor is
    alive -- ShardedLDLQ's queue holds a reference to every H it describes)"""
    return (int(H.data_ptr()), tuple(H.shape))


_active = None


def activate(handle):
    """Install (or clear, with None) the ShardedLDLQ that quantize_weight_vecbal routes its rounding through."""
    global _active
    _active = handle


def active():
    return _active
