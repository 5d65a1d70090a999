#!/usr/bin/env python3
"""Launch quipamd_dequant_gemm at one shape a few times -- the command the round-end rocprofv3 counter passes wrap for the MFMA-bound
legs (bs 256 weight stream, prefill): `rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES ... -- python scripts/prof_k2_shape.py m d bs`."""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops  # noqa: E402

m, d, bs = (int(v) for v in sys.argv[1:4])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
codes = torch.randint(0, 4, (m, d), generator=g, dtype=torch.uint8).to(dev)
q = ops.pack(codes, 2, ops.LAYOUT_STREAM)
x = torch.randn(bs, d, generator=g).to(torch.bfloat16).to(dev)
sc = torch.tensor([0.05], device=dev)
y = torch.empty(bs, m, dtype=torch.bfloat16, device=dev)
for _ in range(steps):
    ops.dequant_gemm(x, q, 2, "b", sc, None, None, out=y)
torch.cuda.synchronize()
print("ok", m, d, bs, steps)
