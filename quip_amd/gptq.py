"""`GPTQ` (OPTQ) behind the QuantMethod protocol -- reference gptq.py:17-115.

Surface row of SURVEY.md 8(a) a13: `--quant gptq` must keep working next to LDLQ.  The column quantiser is
the HIP grid kernel (ops.quantize via Quantizer.quantize); the Cholesky-inverse stays on rocSOLVER through torch.
For the common case (nn.Linear, groupsize -1, qfn a, width a multiple of 16) the d-step column loop and the lazy
block update W[:, i2:] -= Err @ Hinv[i1:i2, i2:] (gptq.py:56-93) run as ONE launch of the K4 kernel in its
updated-weight feedback mode (ops.gptq_round, include/quip_amd.h): the loop below is ~8 launches per column
(0.5 s for d = 8192), the kernel a few ms.  Everything else (groupsize, Conv layers, qfn c, debug_equiv) takes the
reference-order loop.
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
        Q = torch.zeros_like(W)
        # upper Cholesky factor of H^-1 (gptq.py:51-54)
        Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
        qz = self.quantizer
        fast = (USE_KERNEL and groupsize == -1 and not debug_equiv and isinstance(self.layer, nn.Linear) and W.is_cuda
                and qz.qfn == 'a' and self.columns % 16 == 0 and float(qz.maxq) in (1.0, 3.0, 7.0, 15.0, 255.0))
        if fast:
            from . import ops
            mq = int(qz.maxq.item()) if torch.is_tensor(qz.maxq) else int(qz.maxq)
            bits = (mq + 1).bit_length() - 1
            # qfn b is excluded: Quantizer.quantize recomputes its scalar scale from every (updated) column it is handed
            # grid coordinates WITHOUT the clamp of the LDLQ grid map (vector_balance.py:515 clamps, quantize_qfna does
            # not: OPTQ feeds the unclamped residual back)
            wg = (W.float() / qz.scale.reshape(-1, 1).float() + qz.zero.reshape(-1, 1).float()).contiguous()
            codes = ops.gptq_round(wg, Hinv.float().contiguous(), bits)
            Q = ops.codes_to_weight(codes, 'a', qz.scale, qz.zero, mq, out_dtype=torch.float32).to(W.dtype)
            self.codes, self.qscale, self.qzero = codes, qz.scale.reshape(-1).float(), qz.zero.reshape(-1).float()
        for i1 in range(0, self.columns if not fast else 0, blocksize):
            i2 = min(i1 + blocksize, self.columns)
            Wb = W[:, i1:i2].clone()
            Qb = torch.zeros_like(Wb)
            Eb = torch.zeros_like(Wb)
            Hb = Hinv[i1:i2, i1:i2]
            for i in range(i2 - i1):
                col = Wb[:, i]
                if groupsize != -1 and (i1 + i) % groupsize == 0:
                    self.quantizer.find_params(W[:, (i1 + i):(i1 + i + groupsize)], weight=True)
                q = self.quantizer.quantize(col.unsqueeze(1)).flatten().to(col.dtype)
                Qb[:, i] = q
                e = (col - q) / Hb[i, i]
                Wb[:, i:] -= e.unsqueeze(1) * Hb[i, i:].unsqueeze(0)     # rank-1 update inside the block
                Eb[:, i] = e
            Q[:, i1:i2] = Qb
            W[:, i2:] -= Eb @ Hinv[i1:i2, i2:]                              # lazy batch update (gptq.py:90)
        torch.cuda.synchronize()
        self.time = time.time() - tick
        if isinstance(self.layer, transformers.Conv1D):
            Q = Q.t()
        self.layer.weight.data = Q.reshape(self.layer.weight.shape).to(self.layer.weight.data.dtype)
        self.postproc()
        self.error_compute(full_W, self.layer.weight.data)
        if not copy_H:
            del self.H
