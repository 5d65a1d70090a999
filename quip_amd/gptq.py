"""`GPTQ` (OPTQ) behind the QuantMethod protocol -- reference gptq.py:17-115.

Surface row of SURVEY.md 8(a) a13: `--quant gptq` must keep working next to LDLQ.  The column quantiser is
the HIP grid kernel (ops.quantize via Quantizer.quantize); the Cholesky-inverse and the lazy block update
W[:, i2:] -= Err @ Hinv[i1:i2, i2:] (gptq.py:90) are plain library calls on the device (rocSOLVER / rocBLAS
through torch), listed under "next" in DESIGN.md for a fused kernel on the K4 machinery.
"""
import time

import torch
import torch.nn as nn
import transformers

from .method import QuantMethod
from .quant import *  # noqa: F401,F403  (the reference star-imports quant here, gptq.py:9)

DEBUG = False

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
        for i1 in range(0, self.columns, blocksize):
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
