"""-m gpu: permanent stress of `ops.gptq_feedback` (flip -> K8 in place -> unit-triangular inverse -> finish; gptq.py:51-54) at ragged
widths.  Round 5 recorded ONE NaN from width 2080 (= 32 * 65: a 32-wide tail behind the last 64-row panel) in a full-suite run that
never reproduced.  Three things are exercised here, every repetition:

  * every workspace the call allocates (`work`, `FT`) and the inverse's output come out of the caching allocator NaN-FILLED, so a read
    of memory the call did not write itself reaches the result (NaN propagates through every product, also through x 0);
  * a second stream keeps the memory system and the CUs busy with unrelated work of varying length, so the kernels of the chain start
    at varying offsets from each other (the failure was timing dependent if it was anything);
  * the three stages are also run apart (K8 alone, the inverse alone on NaN-prefilled X / work), to name the stage if one fails.

QUIP_FEEDBACK_REPS raises the repetition count (scripts/gpu_round.sh's stress leg: 500)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPS = int(os.environ.get("QUIP_FEEDBACK_REPS", "12"))


def _fixture(d):
    from quip_amd import ops
    g = torch.Generator().manual_seed(d)
    X = torch.randn(2 * d, d, generator=g)
    H = X.T @ X / (2 * d) + 0.05 * torch.eye(d)
    Hinv = torch.linalg.cholesky(torch.linalg.inv(H.double()), upper=True)
    return H.to(DEV), ops.gptq_feedback_matrix(Hinv)


def _poison(d):
    ts = [torch.full((n,), float("nan"), device=DEV) for n in (d * d, 2 * d * d, d * d, d * d)]
    del ts


@pytest.mark.parametrize("d", [2080, 2064, 1104, 4128])
def test_feedback_under_poison_and_load(d):
    from quip_amd import ops, _lib
    Hd, ref = _fixture(d)
    tol = 2e-4 * max(1.0, float(ref.abs().max()))
    side = torch.cuda.Stream()
    noise = torch.randn(4096, 4096, device=DEV)
    sink = torch.empty_like(noise)
    for it in range(REPS):
        with torch.cuda.stream(side):                                   # unrelated work of varying length beside the chain
            for _ in range(1 + it % 5):
                torch.mm(noise, noise, out=sink)
                sink.mul_(1e-3)
        _poison(d)
        FT = ops.gptq_feedback(Hd)
        bad = ~torch.isfinite(FT)
        assert not bool(bad.any()), f"d={d} it={it}: non-finite FT at {torch.nonzero(bad)[:8].tolist()}"
        err = float((FT.cpu().double() - ref).abs().max())
        assert err <= tol, f"d={d} it={it}: {err}"
        # the stages apart: K8 on the flipped Hessian, then the inverse into NaN-prefilled X / work
        _poison(d)
        Hf = torch.flip(Hd, [0, 1]).contiguous()
        LT = ops.cholesky_lt(Hf)
        if not bool(torch.isfinite(LT).all()):                          # say where, and whether it repeats on the same input
            w = torch.nonzero(~torch.isfinite(LT))
            again = ops.cholesky_lt(Hf)
            raise AssertionError(f"d={d} it={it}: K8 produced {w.shape[0]} non-finite entries, rows {int(w[:, 0].min())}..{int(w[:, 0].max())}, columns "
                                 f"{int(w[:, 1].min())}..{int(w[:, 1].max())}, first {w[:6].tolist()}; input finite: {bool(torch.isfinite(Hf).all())}; the same "
                                 f"call again: {int((~torch.isfinite(again)).sum())} non-finite entries")
        Xo = torch.full((d, d), float("nan"), device=DEV)
        work = torch.full((d, d), float("nan"), device=DEV)
        _lib.call("quipamd_unit_upper_inverse", ops._p(LT), ops._p(Xo), ops._p(work), d, ops._stream())
        assert bool(torch.isfinite(torch.triu(Xo)).all()), f"d={d} it={it}: the inverse read something it did not write"
        side.synchronize()
