#!/usr/bin/env python3
"""where do two chain forms of csrc/gptq_qfnb.hip differ? (round 6 debugging aid)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops
DEV = "cuda:0"
for (m, d) in [(8192, 208), (8192, 64), (8192, 128), (4100, 208), (11008, 128), (8192, 2048)]:
    g = torch.Generator().manual_seed(m + d)
    W = (0.02 * torch.randn(m, d, generator=g)).to(DEV)
    X = torch.randn(2 * d + 64, d, generator=g).to(DEV) * (0.5 + torch.rand(d, generator=g).to(DEV))
    H = X.T @ X / X.shape[0]
    H += 0.01 * H.diag().mean() * torch.eye(d, device=DEV)
    FT = ops.gptq_feedback(H)
    outs = {}
    for form in (0, 64):
        ops.gptq_qfnb_debug(0, 0, form)
        outs[form] = ops.gptq_round_qfnb(W.clone(), FT, 2)
    ops.gptq_qfnb_debug(0, 0, 0)
    (q0, c0), (q1, c1) = outs[0], outs[64]
    step = 2.0 * c1[None, :] / 3
    bad = (q0 - q1).abs() > 0.25 * step
    print(f"{m}x{d}: flipped {bad.float().mean().item():.5f}; scale rel diff max {((c0 - c1).abs() / c1).max().item():.2e}; "
          f"first bad column {int(bad.any(0).float().argmax()) if bad.any() else -1}; bad per column (first 12 with any): "
          f"{[(int(c), int(bad[:, c].sum())) for c in torch.nonzero(bad.any(0))[:12, 0]]}; rows with any bad: {int(bad.any(1).sum())}, "
          f"row range {int(torch.nonzero(bad.any(1))[0, 0]) if bad.any() else -1}..{int(torch.nonzero(bad.any(1))[-1, 0]) if bad.any() else -1}")
