"""`GPTQ` (OPTQ) behind the QuantMethod protocol -- reference gptq.py:17-115.

Surface row of SURVEY.md 8(a) a13: `--quant gptq` must keep working next to LDLQ.  The d-step column loop and the lazy
block update W[:, i2:] -= Err @ Hinv[i1:i2, i2:] (gptq.py:56-93) run as ONE launch of the K4 kernel in its updated-weight
feedback modes (include/quip_amd.h), fed by ops.gptq_feedback (flip + K8 Cholesky + one unit-triangular inverse on the fp32
matrix pipe instead of gptq.py:51-54's three rocSOLVER factorisations):
  * ops.gptq_round         qfn a, groupsize -1: grid coordinates, codes kept for packing;
  * ops.gptq_round_groups  groupsize 16/32/64/128 (the group quantisers are found inside the kernel from the block-lazy W,
                           like gptq.py:72-75) and qfn c, in weight units with the reference's quantiser formula;
(nn.Linear is the only layer kind the reference's own preproc / error_compute survive: method.py:187 and :232 index a
Conv1D / Conv2d weight as if it were [out, in].)
  * ops.gptq_round_qfnb    qfn b (`--incoh_processing`): Quantizer.quantize recomputes ONE scale from all rows of every updated column
                           (quant.py:158-160) -- d grid-wide reductions in series, which K4's 16-rows-per-workgroup sweep cannot hold:
                           csrc/gptq_qfnb.hip (co-resident workgroups, data-tagged granules, ~3 us per column).
`_column_walk` (the reference's loop, in torch) serves what is left: debug_equiv's float64, more than 32768 rows, in_features % 16 != 0.
"""
import time

import torch
import torch.nn as nn
import transformers

from .method import QuantMethod
from .quant import *  # noqa: F401,F403  (the reference star-imports quant here, gptq.py:9)

DEBUG = False
USE_KERNEL = True        # False: always take the reference-order column loop

torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False


class GPTQ(QuantMethod):

    def fasterquant(self, blocksize=128, groupsize=-1, copy_H=False, debug_equiv=False):
        W = self.layer.weight.data.clone()
        if isinstance(self.layer, nn.Conv2d):
            W = W.flatten(1)
        if isinstance(self.layer, transformers.Conv1D):
            W = W.t()
        if not debug_equiv:
            W = W.float()
        full_W = W.clone()
        tick = time.time()
        if not self.quantizer.ready():
            self.quantizer.find_params(W, weight=True)
        H = self.H.data.clone() if copy_H else self.H
        Q = self._kernel_round(W, H, groupsize, debug_equiv, blocksize)
        if Q is None:
            # upper Cholesky factor of H^-1 (gptq.py:51-54)
            Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
            Q = _column_walk(W, Hinv, self.quantizer, blocksize, groupsize)
        torch.cuda.synchronize()
        self.time = time.time() - tick
        if isinstance(self.layer, transformers.Conv1D):
            Q = Q.t()
        self.layer.weight.data = Q.reshape(self.layer.weight.shape).to(self.layer.weight.data.dtype)
        self.postproc()
        self.error_compute(full_W, self.layer.weight.data)
        if not copy_H:
            del self.H

    def _kernel_round(self, W, H, groupsize, debug_equiv, blocksize=128):
        """the kernels want in_features % 16 == 0.  A ragged width is padded on the RIGHT with zero columns whose Hessian block is a
        multiple of the identity: H' = diag(H, c I) factors block by block (gptq.py:51-54: Hinv' = diag(Hinv, I / sqrt(c))), so no error of
        a real column reaches a padded one or comes back from it -- the real columns see exactly the sweep of gptq.py:56-93; the padded
        columns, their codes and their column scales are cut off again (round 6; the column walk below now serves debug_equiv, blocksize != 128
        with groups, and a qfn-b Linear with more rows than co-resident workgroups)."""
        d = W.shape[1] if W.dim() == 2 else 0
        pad = (-d) % 16
        if not pad or not USE_KERNEL or debug_equiv or W.dim() != 2 or not W.is_cuda or groupsize != -1:
            return self._kernel_round_aligned(W, H, groupsize, debug_equiv, blocksize)
        Wp = torch.nn.functional.pad(W, (0, pad))
        Hp = torch.zeros((d + pad, d + pad), dtype=H.dtype, device=H.device)
        Hp[:d, :d] = H
        Hp[d:, d:] = torch.eye(pad, dtype=H.dtype, device=H.device) * H.diagonal().mean()
        Q = self._kernel_round_aligned(Wp, Hp, groupsize, debug_equiv, blocksize)
        if Q is None:
            return None
        if getattr(self, 'codes', None) is not None and self.codes.shape[-1] == d + pad:
            self.codes = self.codes[:, :d].contiguous()
        if getattr(self, 'column_scale', None) is not None and self.column_scale.shape[0] == d + pad:
            self.column_scale = self.column_scale[:d].contiguous()
            self.quantizer.scale = self.column_scale[-1].clone()         # the last REAL column's scale (quant.py:158-160)
        return Q[:, :d].contiguous()

    def _kernel_round_aligned(self, W, H, groupsize, debug_equiv, blocksize=128):
        """the K4 launch that covers this configuration, or None.  The feedback matrix comes from H through K8 and one
        triangular inverse (ops.gptq_feedback) -- the Cholesky / inverse / Cholesky of gptq.py:51-54 is never formed."""
        qz = self.quantizer
        mq = int(qz.maxq.item()) if torch.is_tensor(qz.maxq) else int(qz.maxq)
        ok = (USE_KERNEL and not debug_equiv and W.is_cuda and W.dim() == 2 and W.shape[1] % 16 == 0 and qz.qfn in ('a', 'b', 'c')
              and mq in (1, 3, 7, 15, 255))
        if not ok:
            return None
        from . import ops
        m, d = W.shape
        bits = (mq + 1).bit_length() - 1
        if qz.qfn == 'b':
            # every column on its own scale, recomputed from all rows of the updated column (quant.py:158-160): whatever find_params left
            # in the quantiser -- per group or not -- is overwritten before it is used, so one kernel serves every groupsize
            from ._lib import QuipAmdError
            try:
                Q, colscale = ops.gptq_round_qfnb(W.float().contiguous(), ops.gptq_feedback(H.float()), bits)
            except QuipAmdError as e:                                 # more rows than co-resident workgroups on this device: the column walk
                if "co-resident" not in str(e):
                    raise
                return None
            qz.scale = colscale[-1].clone()                           # what the reference's quantiser is left holding: the last column's
            self.column_scale = colscale
            return Q.to(W.dtype)
        if groupsize == -1:
            if qz.scale.numel() not in (1, m):
                return None
            FT = ops.gptq_feedback(H.float())
            if qz.qfn == 'a':
                # grid coordinates WITHOUT the clamp of the LDLQ grid map (vector_balance.py:515 clamps, quantize_qfna does
                # not: OPTQ feeds the unclamped residual back)
                wg = (W.float() / qz.scale.reshape(-1, 1).float() + qz.zero.reshape(-1, 1).float()).contiguous()
                codes = ops.gptq_round(wg, None, bits, FT=FT)
                self.codes, self.qscale, self.qzero = codes, qz.scale.reshape(-1).float(), qz.zero.reshape(-1).float()
                return ops.codes_to_weight(codes, 'a', qz.scale, qz.zero, mq, out_dtype=torch.float32).to(W.dtype)
            Q, _, _ = ops.gptq_round_groups(W.float().contiguous(), None, bits, -1, qz.sym, 'c', qz.scale, qz.zero, FT=FT)
            return Q.to(W.dtype)
        # the kernel's lazy block is 128 columns wide: with another blocksize the groups would see a different W (gptq.py:72-75)
        if groupsize not in (16, 32, 64, 128) or d % groupsize or not qz.perchannel or qz.mse or blocksize != 128:
            return None
        Q, scale, zero = ops.gptq_round_groups(W.float().contiguous(), None, bits, groupsize, qz.sym, qz.qfn, FT=ops.gptq_feedback(H.float()))
        # the reference leaves the LAST group's quantiser in self.quantizer (gptq.py:72-75)
        qz.scale, qz.zero = scale[:, -1:].clone(), zero[:, -1:].clone()
        self.group_scale, self.group_zero = scale, zero
        return Q.to(W.dtype)


def _column_walk(W, Hinv, quantizer, blocksize, groupsize):
    """gptq.py:56-93 for the configurations no kernel covers (module docstring): one column at a time, errors of a block
    applied to the later blocks once per block."""
    d = W.shape[1]
    Q = torch.empty_like(W)
    for lo in range(0, d, blocksize):
        hi = min(lo + blocksize, d)
        blk = W[:, lo:hi].clone()
        Hb = Hinv[lo:hi, lo:hi]
        resid = torch.empty_like(blk)
        for j in range(hi - lo):
            if groupsize != -1 and (lo + j) % groupsize == 0:
                quantizer.find_params(W[:, lo + j:lo + j + groupsize], weight=True)
            col = blk[:, j]
            q = quantizer.quantize(col.unsqueeze(1)).flatten().to(col.dtype)
            Q[:, lo + j] = q
            resid[:, j] = (col - q) / Hb[j, j]
            blk[:, j:].addr_(resid[:, j], Hb[j, j:], alpha=-1)
        W[:, hi:] -= resid @ Hinv[lo:hi, hi:]
    return Q
