#!/usr/bin/env python3
"""tests/test_gpu_edge_round2.py::test_gptq_feedback_ragged_widths[2080] returned NaN once (profiles/r05a): hunt for a read of
uninitialised memory by POISONING the caching allocator (NaN-filled blocks freed back to it, so every torch.empty hands out NaNs),
and bisect gptq_feedback = flip -> K8 in place -> unit-upper inverse -> finish."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quip_amd import ops, _lib

DEV = "cuda:0"


def poison(mb=3000):
    ts = [torch.full((s,), float("nan"), device=DEV) for s in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 22, 1 << 20, 4327680, 4326400, 8652800, 2080 * 2080, 2 * 2080 * 2080)]
    del ts


def main():
    res = {}
    for d in (2080, 1104, 2064, 48, 4128):
        g = torch.Generator().manual_seed(d)
        X = torch.randn(2 * d, d, generator=g)
        H = (X.T @ X / (2 * d) + 0.05 * torch.eye(d))
        Hinv = torch.linalg.cholesky(torch.linalg.inv(H.double()), upper=True)
        ref = ops.gptq_feedback_matrix(Hinv)
        Hd = H.to(DEV)
        bad_fb = bad_k8 = bad_inv = 0
        worst = 0.0
        for it in range(12):
            poison()
            FT = ops.gptq_feedback(Hd)
            e = (FT.cpu().double() - ref).abs().max().item()
            if not (e <= 2e-4):
                bad_fb += 1
            worst = max(worst, e if e == e else float("inf"))
            # K8 alone, in place and out of place, poisoned output
            poison()
            Hf = torch.flip(Hd, [0, 1]).contiguous()
            LT = ops.cholesky_lt(Hf)
            if not bool(torch.isfinite(LT).all()):
                bad_k8 += 1
            # inverse alone with a NaN-prefilled X (the C entry as gptq_feedback calls it)
            poison()
            Xo = torch.full((d, d), float("nan"), device=DEV)
            work = torch.full((d, d), float("nan"), device=DEV)
            _lib.call("quipamd_unit_upper_inverse", ops._p(LT), ops._p(Xo), ops._p(work), d, ops._stream())
            up = torch.triu(Xo)
            if not bool(torch.isfinite(up).all()):
                bad_inv += 1
                nanpos = torch.nonzero(~torch.isfinite(up))
                print(f"d={d} it={it}: inverse non-finite at {nanpos.shape[0]} places, first {nanpos[:5].tolist()} last {nanpos[-5:].tolist()}", flush=True)
        res[d] = dict(bad_feedback=bad_fb, bad_k8=bad_k8, bad_inverse=bad_inv, worst_err=worst)
        print(d, res[d], flush=True)


if __name__ == "__main__":
    main()
