"""-m gpu: K4 with two groups of 16 rows per workgroup (round 3: the far-field LT slabs shared by twice the MFMAs; chosen from 8192 rows
on where it saves rounds of workgroups) against the one-group form that the kernel-order oracle pins bit for bit
(tests/test_gpu_ortho_ldlq.py): rows are independent and every row keeps its summation order, so the outputs must be IDENTICAL -- in every mode of
the kernel (LDLQ, OPTQ in grid units, OPTQ with group quantisers, the greedy pass), for ragged row counts too."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lt(d, seed):
    from quip_amd import ops
    torch.manual_seed(seed)
    X = torch.randn(d + 128, d, device=DEV)
    H = X.T @ X / d + 0.01 * torch.eye(d, device=DEV)
    return H, ops.cholesky_lt(H)


@pytest.mark.parametrize("m,d", [(32, 256), (40, 384), (100, 1040), (8192, 512), (8200, 272)])
def test_two_row_groups_change_no_bit(m, d):
    from quip_amd import ops
    H, LT = _lt(d, m + d)
    torch.manual_seed(1)
    W = torch.rand(m, d, device=DEV) * 3
    eta = torch.rand(m, d, device=DEV)
    FT = ops.gptq_feedback(H)
    outs = {}
    try:
        for rg in (1, 2):
            ops.ldlq_config(rg)
            a = ops.ldlq_round(W, LT, 2, return_err=True)
            b = ops.ldlq_round(W, LT, 4, eta=eta)
            c = ops.gptq_round(W * 5, None, 4, FT=FT)
            dq = ops.gptq_round_groups(W - 1.5, None, 3, 16, False, 'a', FT=FT, return_codes=True)
            e = ops.gptq_round_groups(W - 1.5, None, 4, -1, True, 'c', scale=torch.full((m,), 0.2, device=DEV), zero=torch.full((m,), 8.0, device=DEV), FT=FT)
            outs[rg] = (a[0], a[1], b, c, *dq, *e)
    finally:
        ops.ldlq_config(0)
    for x, y in zip(outs[1], outs[2]):
        assert torch.equal(x, y)
