"""LDLQ rounding entry points -- the surface of the reference's vector_balance.py on the hot path
(round_ldl :155-199, round_ldl_block :218-291, dispatcher quantize_weight_vecbal :500-532), backed by
quip_amd/csrc/ldlq.hip (K4) and gridmap.hip (K5).

n_greedy_passes > 0 (the LDLQ-RG post-processing, :186-196 / :263-288) runs as one fp32 GEMM s @ H plus one launch of
K4's third mode per pass; `ldlqRG` adds the diag(H) sort of :139-153 / :202-217.
Out of scope (research variants): allbal, ADMM.
"""
import torch

from . import ops
from . import shard


SHARD_RAW16 = True      # row-sharded runs scatter 16-bit layers in their own dtype (False: fp32 grid coordinates, rounds 1-5)


def check_nbits(wr, nbits):
    """vector_balance.py:8-11."""
    vals, counts = torch.unique(wr, sorted=True, return_counts=True)
    assert len(vals) <= 2 ** nbits
    return counts


def hessian_loss(dw, H):
    """vector_balance.py:14-15."""
    return ((dw @ H) * dw).sum()


def _draw_eta(shape, device):
    """the uniform draw of `unbiased` rounding, vector_balance.py:174-175: torch.rand on the CPU global generator, then moved.  With the
    operator prefetch thread on (method.OPERATOR_PREFETCH) that generator may already have been advanced by draws for LATER Linears: drain
    the prefetcher first, which rewinds numpy and torch to where the reference's streams stand at this point."""
    from . import method
    if method.OPERATOR_PREFETCH:
        method.operator_prefetcher().drain()
    return torch.rand(shape).to(device)


def _ldl_transposed(H):
    """unit-lower LDL factor of H as LT = L^T - strict (vector_balance.py:171-173): blocked fp32 Cholesky of the upper
    triangle + row scaling in quip_amd/csrc/cholesky.hip (K8); raises LinAlgError like torch.linalg.cholesky."""
    return ops.cholesky_lt(H.to(torch.float32))


def _greedy_passes(w, codes, H, nbits, n_greedy_passes):
    """vector_balance.py:182-196 (and the block form :259-288, identical up to fp summation order): coordinate descent
    on tr((wr - w) H (wr - w)^T) over the integer grid, right to left, clamp after every pass, stop at a fixed point."""
    import sys
    w_hat = codes.to(torch.float32)
    wr = w_hat.clone()
    s = w_hat - w
    Hn = (H / H.diag().max()).to(torch.float32).contiguous()
    negU = (-torch.triu(Hn, diagonal=1)).contiguous()
    hd = Hn.diag().contiguous()
    maxq = float(2 ** nbits - 1)
    for igp in range(n_greedy_passes):
        wr_new, eps = ops.ldlq_greedy_pass(wr, s @ Hn, negU, hd)
        s -= eps
        wr = torch.clamp(wr_new, min=0, max=maxq)
        if bool((w_hat == wr).all()):
            sys.stderr.write(f"breaking after {igp+1} greedy passes found fixed point")
            break
        w_hat.copy_(wr)
    return wr.to(torch.uint8)


def _round_ldl_codes(w, H, nbits, n_greedy_passes, unbiased, raw=None):
    """raw (row-sharded runs only): (W 16-bit, qfn, scale, zero, maxq) instead of the grid coordinates `w` -- the rows travel in their own
    dtype and every rank maps its chunk onto the grid itself (shard.ldlq_round_sharded)"""
    assert (not unbiased) or (n_greedy_passes == 0), "greedy passes are incompatible with unbiased LDL rounding"
    if raw is not None:
        assert w is None and n_greedy_passes == 0 and shard.active() is not None
        eta = _draw_eta(raw[0].shape, raw[0].device) if unbiased else None
        sharded = shard.active()
        key = shard.h_key(H)
        LT = None if sharded.queued(key) else _ldl_transposed(H)
        return sharded.round(None, LT, nbits, eta=eta, key=key, raw=raw)
    w = w.to(torch.float32)
    if n_greedy_passes != 0:
        assert shard.active() is None, "greedy passes are not row-sharded (they need s @ H on the owner)"
        codes = ops.ldlq_round(w, _ldl_transposed(H), nbits, eta=None)
        return _greedy_passes(w, codes, H.to(torch.float32), nbits, n_greedy_passes)
    eta = _draw_eta(w.shape, w.device) if unbiased else None
    sharded = shard.active()
    if sharded is not None:                                           # rows split over the ranks of the node (shard.py)
        key = shard.h_key(H)                                          # a queued LT is used only for the H it was factored from
        LT = None if sharded.queued(key) else _ldl_transposed(H)
        return sharded.round(w, LT, nbits, eta=eta, key=key)
    return ops.ldlq_round(w, _ldl_transposed(H), nbits, eta=eta)


def round_ldl(w, H, nbits, n_greedy_passes=9, unbiased=False):
    """integer-valued fp32 codes, w in R^{m,d} grid coordinates (vector_balance.py:155-199)."""
    codes = _round_ldl_codes(w, H, nbits, n_greedy_passes, unbiased).to(torch.float32)
    check_nbits(codes, nbits)
    return codes


def round_ldl_block(w, H, nbits, blocksize=128, n_greedy_passes=9, unbiased=False):
    """`--lazy_batch` variant (vector_balance.py:218-291).  In the reference `blocksize` only regroups the SAME sum: column i is rounded at
    w_i + (W1 - WHat1) @ L1[i1:i2, i] + W2Hdiff @ L1[i2:, i] (:254) -- the in-block part plus the far field of every finished block -- which
    is round_ldl's single mat-vec (:180) split at the block edge; the codes of two block sizes differ only where the fp32 summation order
    moves a value across a rounding boundary (0 of 2.1 M codes at 2 bits between the block and the plain form).  K4's
    lazy block is a compile-time 128 columns (csrc/ldlq.hip BS: LDS image of the diagonal block, MFMA tiling of the far field), with its own
    summation order inside and across blocks, so every `blocksize` runs that kernel: the caller's value is validated like the reference's
    loop would use it (a positive integer, :243) and does not change the launch."""
    if not (isinstance(blocksize, int) and blocksize >= 1):
        raise ValueError(f"round_ldl_block: blocksize must be a positive integer (vector_balance.py:243 steps range(d, 0, -blocksize)); got {blocksize!r}")
    return round_ldl(w, H, nbits, n_greedy_passes=n_greedy_passes, unbiased=unbiased)


def round_ldl_gptqequiv(w, H, nbits, unbiased=False):
    """LDLQ in OPTQ's column order (vector_balance.py:381-422, used by optq_ldlq_equiv.py): Cholesky of the flipped H,
    factor flipped back, columns rounded left to right -- i.e. round_ldl on the column-reversed problem."""
    w = w.to(torch.float32)
    eta = _draw_eta(w.shape, w.device).flip(1).contiguous() if unbiased else None      # eta[:, i] belongs to column i
    LT = _ldl_transposed(torch.flip(H.to(torch.float32), [0, 1]).contiguous())
    codes = ops.ldlq_round(w.flip(1).contiguous(), LT, nbits, eta=eta).flip(1).contiguous().to(torch.float32)
    check_nbits(codes, nbits)
    return codes


def round_sorted_ldlqRG(w, H, nbits, n_greedy_passes=9, unbiased=False, pivot=None):
    """LDLQ-RG: columns sorted by diag(H) ascending, then round_ldl (vector_balance.py:139-153)."""
    p = torch.argsort(torch.diag(H))
    wr = torch.zeros(w.shape, device=w.device)
    wr[:, p] = round_ldl(w[:, p].contiguous(), H[p, :][:, p].contiguous(), nbits, n_greedy_passes, unbiased)
    return wr


def round_sorted_ldlqRG_block(w, H, nbits, n_greedy_passes=9, unbiased=False, pivot=None):
    """vector_balance.py:202-217; shares the kernel with round_sorted_ldlqRG like round_ldl_block does with round_ldl."""
    return round_sorted_ldlqRG(w, H, nbits, n_greedy_passes, unbiased, pivot)


@torch.no_grad()
def quantize_weight_vecbal(w, H, nbits, npasses, scale, zero, maxq, unbiased=False, qfn='a', qmethod='bitbal',
                           lazy_batch=False, return_codes=False):
    """grid map -> LDLQ -> weights, returned as fp16 like the reference (vector_balance.py:500-532).
    return_codes=True additionally returns (codes uint8 [m,d], scale fp32, zero fp32|None): the integer state
    the reference throws away and a packed layer needs (SURVEY.md section 7 "hard parts")."""
    if qmethod not in ('ldlq', 'ldlqRG', 'ldl_gptqequiv'):
        raise NotImplementedError(f"qmethod {qmethod!r} is outside the quip_amd hot path (only 'ldlq' / 'ldlqRG' / 'ldl_gptqequiv')")
    mq = int(maxq.item()) if torch.is_tensor(maxq) else int(maxq)
    if w.dtype == torch.float64:                      # optq_ldlq_equiv.py hands over a float64 FakeLayer: the kernels round in fp32
        w, H = w.float(), H.float()
        scale = None if scale is None else scale.float()
        zero = None if zero is None else zero.float()
    if qmethod == 'ldl_gptqequiv':                # optq_ldlq_equiv.py: LDLQ in OPTQ's column order (vector_balance.py:381-422, :508)
        def rounder(wgrid):
            return round_ldl_gptqequiv(wgrid, H, nbits, unbiased=unbiased).to(torch.uint8)
    elif qmethod == 'ldlqRG':                       # sort the columns by diag(H), round, undo the sort (:139-153)
        perm = torch.argsort(torch.diag(H))
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), device=perm.device)
        Hs = H[perm, :][:, perm].contiguous()

        def rounder(wgrid):
            return _round_ldl_codes(wgrid[:, perm].contiguous(), Hs, nbits, npasses, unbiased)[:, inv].contiguous()
    else:
        def rounder(wgrid):
            return _round_ldl_codes(wgrid, H, nbits, npasses, unbiased)
    # row-sharded plain LDLQ on a 16-bit layer (every HF checkpoint: the Balance path hands over layer.weight.data itself, bal.py:29):
    # the rows are scattered as they are, 2 bytes per weight, and each rank runs the grid map on its chunk (shard.py, round 6)
    raw16 = (shard.active() is not None and SHARD_RAW16 and qmethod == 'ldlq' and npasses == 0 and w.dtype in (torch.float16, torch.bfloat16)
             and qfn in ('a', 'b'))
    if qfn == 'a':
        if raw16:
            codes = _round_ldl_codes(None, H, nbits, npasses, unbiased, raw=(w.contiguous(), 'a', scale, zero, mq))
        else:
            codes = rounder(ops.gridmap(w, 'a', scale, zero, mq))
        out = ops.codes_to_weight(codes, 'a', scale, zero, mq, out_dtype=torch.float16)
        s_out, z_out = scale.reshape(-1).float(), zero.reshape(-1).float()
    elif qfn == 'b':
        s = ops.qfnb_scale(w)                                    # 2.4*rms(w)+1e-16 in w's dtype (:522)
        if raw16:
            codes = _round_ldl_codes(None, H, nbits, npasses, unbiased, raw=(w.contiguous(), 'b', s, None, mq))
        else:
            codes = rounder(ops.gridmap(w, 'b', s, None, mq))
        out = ops.codes_to_weight(codes, 'b', s, None, mq, out_dtype=torch.float16)
        s_out, z_out = s, None
    else:
        return NotImplementedError()                              # sic (vector_balance.py:532)
    return (out, codes, s_out, z_out) if return_codes else out
