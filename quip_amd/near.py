"""`Nearest` -- round-to-nearest through the Quantizer (reference near.py:5-20)."""
import time

from .method import QuantMethod


class Nearest(QuantMethod):

    def fasterquant(self):
        tick = time.time()
        full_W = self.layer.weight.data.clone()
        if not self.quantizer.ready():
            self.quantizer.find_params(full_W, weight=True)
        self.layer.weight.data = self.quantizer.quantize(full_W).to(full_W.dtype)
        self.postproc()
        self.time = time.time() - tick
        self.error_compute(full_W, self.layer.weight.data)
