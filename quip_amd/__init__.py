"""quip_amd -- MI355X-native low-bit linear hot path of QuIP.

Python surface mirrors the reference modules (quant / method / vector_balance / bal / gptq / near /
modelutils); the arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI of
include/quip_amd.h (quip_amd/csrc -> libquip_amd.so, loaded with ctypes in quip_amd/_lib.py).
"""
__version__ = "0.1.0"
