"""`Balance` -- LDLQ behind the QuantMethod protocol (reference bal.py:13-48)."""
import time

import torch.nn as nn
import transformers

from .method import QuantMethod
from .vector_balance import quantize_weight_vecbal


class Balance(QuantMethod):

    def configure(self, qmethod, nbits, npasses, unbiased):
        self.qmethod, self.nbits, self.npasses, self.unbiased = qmethod, nbits, npasses, unbiased

    def fasterquant(self, lazy_batch=False):
        """grid map + LDLQ + write-back + postproc + proxy error, in the reference's order (bal.py:21-48,
        including its quirk of calling error_compute after postproc with pre-postproc weights).
        Besides assigning the dense fp16 weights the callers expect, keeps `codes`, `qscale`, `qzero`
        (integer codes in the projected basis + grid parameters) for QuantLinear.pack."""
        w = self.layer.weight.data.clone()
        if isinstance(self.layer, (nn.Conv2d, transformers.Conv1D)):
            raise NotImplementedError()
        tick = time.time()
        if not self.quantizer.ready():
            self.quantizer.find_params(w, weight=True)
        quant_w, self.codes, self.qscale, self.qzero = quantize_weight_vecbal(
            w=w, H=self.H, nbits=self.nbits, npasses=self.npasses, scale=self.quantizer.scale,
            zero=self.quantizer.zero, maxq=self.quantizer.maxq, unbiased=self.unbiased, qfn=self.quantizer.qfn,
            qmethod=self.qmethod, lazy_batch=lazy_batch, return_codes=True)
        self.layer.weight.data = quant_w
        self.postproc()
        self.time = time.time() - tick
        self.error_compute(w, quant_w)
