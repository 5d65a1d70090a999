#!/usr/bin/env python3
"""tests/test_gpu_feedback_stress.py reproduced the round-5 NaN: K8 (ops.cholesky_lt) at d = 2080 returns non-finite entries when a second
stream keeps the GPU busy and the allocator is poisoned.  Bisect: which ingredient (load / poison), which K8 variant, and WHERE the bad
entries sit (first bad row / column tells the kernel and the pair)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quip_amd import ops

DEV = "cuda:0"


def fixture(d):
    g = torch.Generator().manual_seed(d)
    X = torch.randn(2 * d, d, generator=g)
    H = X.T @ X / (2 * d) + 0.05 * torch.eye(d)
    return torch.flip(H.to(DEV), [0, 1]).contiguous()


def poison(d):
    ts = [torch.full((n,), float("nan"), device=DEV) for n in (d * d, 2 * d * d, d * d, d * d)]
    del ts


def run(d, reps, load, pois, cfg):
    Hf = fixture(d)
    ops.cholesky_config(**cfg)
    try:
        ref = ops.cholesky_lt(Hf).clone()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        noise = torch.randn(4096, 4096, device=DEV)
        sink = torch.empty_like(noise)
        bad, first = 0, None
        for it in range(reps):
            if load:
                with torch.cuda.stream(side):
                    for _ in range(1 + it % 5):
                        torch.mm(noise, noise, out=sink)
                        sink.mul_(1e-3)
            if pois:
                poison(d)
            LT = ops.cholesky_lt(Hf, check=False)
            nf = ~torch.isfinite(LT)
            diff = (LT != ref) & ~nf
            if bool(nf.any()) or bool(diff.any()):
                bad += 1
                if first is None:
                    w = torch.nonzero(nf | diff)
                    first = dict(it=it, nonfinite=int(nf.sum()), differing=int(diff.sum()), rows=(int(w[:, 0].min()), int(w[:, 0].max())),
                                 cols=(int(w[:, 1].min()), int(w[:, 1].max())), head=w[:6].tolist())
            side.synchronize()
        return bad, first
    finally:
        ops.cholesky_config()


def main():
    for d in (2080, 2064, 4128, 2048):
        for name, load, pois, cfg in (("load+poison", True, True, {}), ("load only", True, False, {}), ("poison only", False, True, {}),
                                      ("neither", False, False, {}), ("load+poison, round-1 syrk", True, True, dict(old_syrk=True)),
                                      ("load+poison, unblocked diag", True, True, dict(unblocked_diag=True))):
            bad, first = run(d, 30, load, pois, cfg)
            print(f"d={d} {name:<30} bad {bad:2d}/30  first: {first}", flush=True)


if __name__ == "__main__":
    main()
