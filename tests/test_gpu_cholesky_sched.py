"""-m gpu: K8's round-3 schedule against the round-1 form of the same factorisation.  The branch-free, software-pipelined trailing-update
kernel and the two-stream look-ahead perform, per tile, the same operations in the same order as the guarded single-stream form:
the factors must agree BIT FOR BIT -- a lost dependency between the two streams would show here."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("d", [128, 192, 1024, 1152, 2048, 4096, 2176, 1100, 11008, 12288])
def test_schedule_is_bit_identical_to_the_single_stream_guarded_form(d):
    from quip_amd import ops
    torch.manual_seed(d)
    X = torch.randn(d + 256, d, device=DEV)
    H = X.T @ X / d + 0.01 * torch.eye(d, device=DEV)
    try:
        ops.cholesky_config(old_syrk=True, lookahead=False)
        ref = ops.cholesky_lt(H)
        ops.cholesky_config(old_syrk=False, lookahead=False)
        a = ops.cholesky_lt(H)
        ops.cholesky_config(old_syrk=False, lookahead=True)
        outs = [ops.cholesky_lt(H) for _ in range(3)]             # repeated: a race would not lose every time
        side = torch.cuda.Stream()                                  # and from a non-default stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            outs.append(ops.cholesky_lt(H))
        torch.cuda.current_stream().wait_stream(side)
        ops.cholesky_config()
        outs.append(ops.cholesky_lt(H))                           # the default schedule
    finally:
        ops.cholesky_config()
    assert torch.equal(a, ref)
    for o in outs:
        assert torch.equal(o, ref)


@pytest.mark.parametrize("d", [64, 128, 1024, 1100, 4096])
def test_blocked_diagonal_factorisation(d):
    """round 3: the 64 x 64 diagonal block factored 4 x 16 rows with the updates below each block on the matrix pipe, against the 64-step
    form of rounds 1-2: the same factor up to the order of the 16 products per element, and no further from the fp64 factor"""
    from quip_amd import ops
    torch.manual_seed(d)
    X = torch.randn(d + 256, d, device=DEV)
    H = X.T @ X / d + 0.01 * torch.eye(d, device=DEV)
    try:
        ops.cholesky_config(unblocked_diag=True)
        old = ops.cholesky_lt(H)
        ops.cholesky_config()
        new = ops.cholesky_lt(H)
    finally:
        ops.cholesky_config()
    L64 = torch.linalg.cholesky(H.double())
    want = torch.triu((L64 @ torch.diag(1.0 / torch.diag(L64))).T.contiguous(), 1)
    e_new = float((new.double() - want).norm() / want.norm())
    e_old = float((old.double() - want).norm() / want.norm())
    assert e_new <= 1.5 * e_old + 1e-7, (e_new, e_old)
    assert float((new - old).norm() / old.norm()) <= 20 * e_old + 1e-6
    # a matrix that is not positive definite is reported at the same column
    Hbad = H.clone()
    Hbad[d // 2, d // 2] = -1.0
    for unb in (True, False):
        ops.cholesky_config(unblocked_diag=unb)
        try:
            with pytest.raises(torch.linalg.LinAlgError) as ei:
                ops.cholesky_lt(Hbad)
            assert f"order {d // 2 + 1} " in str(ei.value)
        finally:
            ops.cholesky_config()
