"""LDLQ rounding entry points -- the surface of the reference's vector_balance.py on the hot path
(round_ldl :155-199, round_ldl_block :218-291, dispatcher quantize_weight_vecbal :500-532), backed by
quip_amd/csrc/ldlq.hip (K4) and gridmap.hip (K5).

Out of scope (research variants): allbal, ldlqRG greedy passes, ADMM.
"""
import torch

from . import ops
from . import shard


def check_nbits(wr, nbits):
    """vector_balance.py:8-11."""
    vals, counts = torch.unique(wr, sorted=True, return_counts=True)
    assert len(vals) <= 2 ** nbits
    return counts


def hessian_loss(dw, H):
    """vector_balance.py:14-15."""
    return ((dw @ H) * dw).sum()


def _ldl_transposed(H):
    """unit-lower LDL factor of H as LT = L^T - strict (vector_balance.py:171-173): blocked fp32 Cholesky of the upper
    triangle + row scaling in quip_amd/csrc/cholesky.hip (K8); raises LinAlgError like torch.linalg.cholesky."""
    return ops.cholesky_lt(H.to(torch.float32))


def _round_ldl_codes(w, H, nbits, n_greedy_passes, unbiased):
    if n_greedy_passes != 0:
        raise NotImplementedError("greedy post-passes (LDLQ-RG) are outside the quip_amd hot path; use npasses=0")
    w = w.to(torch.float32)
    eta = torch.rand(w.shape).to(w.device) if unbiased else None     # same CPU draw as vector_balance.py:174-175
    LT = _ldl_transposed(H)
    sharded = shard.active()
    if sharded is not None:                                           # rows split over the ranks of the node (shard.py)
        return sharded.round(w, LT, nbits, eta=eta)
    return ops.ldlq_round(w, LT, nbits, eta=eta)


def round_ldl(w, H, nbits, n_greedy_passes=9, unbiased=False):
    """integer-valued fp32 codes, w in R^{m,d} grid coordinates (vector_balance.py:155-199)."""
    codes = _round_ldl_codes(w, H, nbits, n_greedy_passes, unbiased).to(torch.float32)
    check_nbits(codes, nbits)
    return codes


def round_ldl_block(w, H, nbits, blocksize=128, n_greedy_passes=9, unbiased=False):
    """`--lazy_batch` variant (vector_balance.py:218-291).  The HIP kernel always works in 128-column lazy
    blocks, so this and round_ldl share one implementation; they differ in the reference only by fp32
    summation order."""
    assert blocksize == 128, "the kernel's lazy block is 128 columns (vector_balance.py:222 default)"
    return round_ldl(w, H, nbits, n_greedy_passes=n_greedy_passes, unbiased=unbiased)


@torch.no_grad()
def quantize_weight_vecbal(w, H, nbits, npasses, scale, zero, maxq, unbiased=False, qfn='a', qmethod='bitbal',
                           lazy_batch=False, return_codes=False):
    """grid map -> LDLQ -> weights, returned as fp16 like the reference (vector_balance.py:500-532).
    return_codes=True additionally returns (codes uint8 [m,d], scale fp32, zero fp32|None): the integer state
    the reference throws away and a packed layer needs (SURVEY.md section 7 "hard parts")."""
    if qmethod != 'ldlq':
        raise NotImplementedError(f"qmethod {qmethod!r} is outside the quip_amd hot path (only 'ldlq')")
    mq = int(maxq.item()) if torch.is_tensor(maxq) else int(maxq)
    if qfn == 'a':
        wgrid = ops.gridmap(w, 'a', scale, zero, mq)
        codes = _round_ldl_codes(wgrid, H, nbits, npasses, unbiased)
        out = ops.codes_to_weight(codes, 'a', scale, zero, mq, out_dtype=torch.float16)
        s_out, z_out = scale.reshape(-1).float(), zero.reshape(-1).float()
    elif qfn == 'b':
        s = ops.qfnb_scale(w)                                    # 2.4*rms(w)+1e-16 in w's dtype (:522)
        wgrid = ops.gridmap(w, 'b', s, None, mq)
        codes = _round_ldl_codes(wgrid, H, nbits, npasses, unbiased)
        out = ops.codes_to_weight(codes, 'b', s, None, mq, out_dtype=torch.float16)
        s_out, z_out = s, None
    else:
        return NotImplementedError()                              # sic (vector_balance.py:532)
    return (out, codes, s_out, z_out) if return_codes else out
