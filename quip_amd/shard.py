"""Row-sharded LDLQ across the GPUs of one node (BASELINE.json north_star: "per-layer LDLQ is
embarrassingly parallel and is sharded across the 8 GPUs of one node with RCCL over xGMI for the layer
scatter only").

What shards and why
  * inside one Linear the rows of W are independent given L (vector_balance.py:179-180 is row-wise), and the
    LDLQ cost is proportional to m*d^2, so the unit of work is a ROW CHUNK of one Linear -- whole-Linear
    assignment would cap the speed-up at 1.5x because fc2 is 2/3 of an OPT block;
  * everything that couples rows (the qfn-b scalar scale = global RMS, vector_balance.py:522; U*W row mixing,
    method.py:175; the Cholesky of H) is done by the owner BEFORE the scatter, so the exchange is
        broadcast  LT      [d, d]        fp32   (owner -> all)
        scatter    Wgrid   [m/k, d]      fp32   (owner -> rank r)
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


def _default_compute(wgrid, LT, bits, eta):
    from . import ops                          # HIP kernel K4; raises on CPU tensors -- no fallback
    return ops.ldlq_round(wgrid, LT, bits, eta=eta)


def _pack_chunk(codes, bits):
    from . import ops
    return ops.pack(codes, bits, ops.LAYOUT_STREAM)


def _unpack_all(words, bits, m, d):
    from . import ops
    return ops.unpack(words, bits, ops.LAYOUT_STREAM, m, d)


def ldlq_round_sharded(wgrid, LT, bits, eta=None, src=0, group=None, compute=None, gather_packed=None):
    """LDLQ codes of one Linear with its rows split over the ranks of `group`.
    gather_packed: return the codes to the owner as STREAM-packed words (bits/8 bytes per code instead of 1; the
    STREAM layout is row-tile-major, so the per-rank chunks concatenate into the whole matrix's packing).  Default:
    on for the RCCL backend when the shape packs (bits in {2,4}, d a multiple of 512/bits), off otherwise.

    Collective: every rank calls it.  On `src`: wgrid float32 [m,d] grid coordinates, LT float32 [d,d]
    (ops.unit_lower_t of the Cholesky factor), eta float32 [m,d] or None; returns codes uint8 [m,d].
    On the other ranks the tensor arguments are ignored (pass None) and None is returned.
    `compute(wgrid_chunk, LT, bits, eta_chunk) -> uint8 codes` defaults to the HIP kernel."""
    compute = compute or _default_compute
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _comm_device(group)
    hdr = torch.zeros(4, dtype=torch.int64, device=dev)
    if rank == src:
        assert wgrid.dim() == 2 and LT.shape == (wgrid.shape[1], wgrid.shape[1])
        hdr = torch.tensor([wgrid.shape[0], wgrid.shape[1], int(bits), int(eta is not None)], dtype=torch.int64,
                           device=dev)
    if world > 1:
        dist.broadcast(hdr, src=src, group=group)
    m, d, bits, has_eta = (int(v) for v in hdr.tolist())
    if world == 1:
        return compute(wgrid, LT, bits, eta)
    if gather_packed is None:
        gather_packed = dist.get_backend(group) == "nccl" and bits in (2, 4) and d % (512 // bits) == 0

    c = row_chunk(m, world)
    pad = c * world - m

    def padded_chunks(t):
        t = t.to(dev, torch.float32)
        if pad:
            t = torch.cat([t, torch.zeros(pad, d, dtype=torch.float32, device=dev)], 0)
        return list(t.split(c, 0))

    if rank == src:
        LT = LT.to(dev, torch.float32).contiguous()
    else:
        LT = torch.empty(d, d, dtype=torch.float32, device=dev)
    dist.broadcast(LT, src=src, group=group)                          # the one tree/ring-shaped transfer

    mine = torch.empty(c, d, dtype=torch.float32, device=dev)
    dist.scatter(mine, padded_chunks(wgrid) if rank == src else None, src=src, group=group)
    eta_mine = None
    if has_eta:
        eta_mine = torch.empty(c, d, dtype=torch.float32, device=dev)
        dist.scatter(eta_mine, padded_chunks(eta) if rank == src else None, src=src, group=group)

    lo, hi = row_partition(m, world)[rank]
    n_real = hi - lo
    codes = torch.zeros(c, d, dtype=torch.uint8, device=dev)
    if n_real > 0:                                                    # padded rows are never rounded
        codes[:n_real] = compute(mine[:n_real], LT, bits, None if eta_mine is None else eta_mine[:n_real])

    if gather_packed:
        words = _pack_chunk(codes, bits)                               # c*d*bits/32 int32 words, whole 16-row tiles
        parts = [torch.empty_like(words) for _ in range(world)] if rank == src else None
        dist.gather(words, parts, dst=src, group=group)
        if rank != src:
            return None
        return _unpack_all(torch.cat(parts, 0), bits, c * world, d)[:m].contiguous()
    parts = [torch.empty(c, d, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == src else None
    dist.gather(codes, parts, dst=src, group=group)
    if rank != src:
        return None
    return torch.cat(parts, 0)[:m].contiguous()


class ShardedLDLQ:
    """Owner-side handle used by vector_balance.quantize_weight_vecbal when a process group with more than one
    rank is active: announces a job to the ranks parked in serve(), then joins the collective itself."""

    def __init__(self, group=None, src=0, compute=None):
        self.group, self.src, self.compute = group, src, compute

    def _announce(self, op):
        dev = _comm_device(self.group)
        dist.broadcast(torch.tensor([op], dtype=torch.int64, device=dev), src=self.src, group=self.group)

    def round(self, wgrid, LT, bits, eta=None):
        if dist.get_world_size(self.group) > 1:
            self._announce(_OP_LDLQ)
        return ldlq_round_sharded(wgrid, LT, bits, eta=eta, src=self.src, group=self.group, compute=self.compute)

    def shutdown(self):
        if dist.get_world_size(self.group) > 1:
            self._announce(_OP_STOP)


def serve(group=None, src=0, compute=None):
    """Worker loop for every rank except the owner: wait for a job header, join the collective, repeat until the
    owner calls ShardedLDLQ.shutdown().  Returns the number of jobs served."""
    dev = _comm_device(group)
    jobs = 0
    while True:
        op = torch.zeros(1, dtype=torch.int64, device=dev)
        dist.broadcast(op, src=src, group=group)
        if int(op.item()) == _OP_STOP:
            return jobs
        ldlq_round_sharded(None, None, 0, src=src, group=group, compute=compute)
        jobs += 1


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
    for m, n, t in zip(methods, counts.tolist(), tri.tolist()):
        assert m.H.dtype == torch.float64, "all_reduce_hessians: call before post_batch"
        H = m.H if m.H.device == dev else m.H.to(dev)
        dist.all_reduce(H, group=group)
        if H is not m.H:
            m.H.copy_(H)
        m.nsamples = int(round(n))
        if t:                         # a rank that saw no sample still has to mirror the triangle in post_batch
            m._tri = True


_active = None


def activate(handle):
    """Install (or clear, with None) the ShardedLDLQ that quantize_weight_vecbal routes its rounding through."""
    global _active
    _active = handle


def active():
    return _active
