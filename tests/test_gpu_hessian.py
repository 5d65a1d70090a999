"""-m gpu: K7 Hessian accumulation (quipamd_hessian_accum / quipamd_hessian_finish) through the C ABI, against the
oracle (oracle/quip_oracle.py: hessian_add_batch / hessian_post_batch, method.py:98-123) and the reference-generated
golden fixture tests/golden/method.npz (X, H64, Hraw)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, f16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from quip_amd import ops as o
    return o


@pytest.fixture(scope="module")
def O():
    from oracle import quip_oracle
    return quip_oracle


def _lower(d):
    return np.tril(np.ones((d, d), dtype=bool))


def _accum(ops, xs, d):
    Hacc = torch.zeros(d, d, dtype=torch.float64, device=DEV)
    for x in xs:
        ops.hessian_accum(Hacc, x)
    return Hacc


def test_golden_fixture(ops, O):
    g = load_golden("method")
    X = torch.from_numpy(f16(g["X"]).copy()).to(DEV)           # [6, 64, 96] fp16
    d = X.shape[-1]
    Hacc = _accum(ops, list(X), d)
    lo = _lower(d)
    np.testing.assert_allclose(Hacc.cpu().numpy()[lo], g["H64"][lo], rtol=1e-13, atol=1e-13)
    H = ops.hessian_finish(Hacc, X.shape[0]).cpu().numpy()
    np.testing.assert_array_equal(H, H.T)
    np.testing.assert_allclose(H, g["Hraw"], rtol=1.2e-7, atol=0)
    assert (H != g["Hraw"]).mean() < 1e-3                       # only fp32 rounding ties may differ


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("tokens,d", [(1, 16), (37, 96), (100, 203), (300, 520), (5, 1000), (64, 3600), (130, 4096)])
def test_matches_oracle(ops, O, dtype, tokens, d):
    g = torch.Generator().manual_seed(tokens * 7919 + d)
    x = (torch.randn(2, tokens, d, generator=g) * torch.linspace(0.1, 3.0, d)).to(dtype)
    ref = np.zeros((d, d), np.float64)
    n = O.hessian_add_batch(ref, x.float().numpy())
    xd = x.to(DEV)
    Hacc = _accum(ops, [xd[0], xd[1]], d)
    scale = np.abs(x.float().numpy().reshape(-1, d).astype(np.float64))
    bound = 1e-13 * (scale.T @ scale)                          # fp64 summation-order differences only
    lo = _lower(d)
    assert (np.abs(Hacc.cpu().numpy() - ref)[lo] <= bound[lo] + 1e-300).all()
    H = ops.hessian_finish(Hacc, n).cpu().numpy()
    want = O.hessian_post_batch(ref, n)
    np.testing.assert_array_equal(H, H.T)
    np.testing.assert_allclose(H, want, rtol=1.2e-7, atol=1e-30)
    assert (H != want).mean() < 1e-3


def test_strided_rows_and_unaligned_base(ops, O):
    tokens, d = 50, 136
    g = torch.Generator().manual_seed(5)
    buf = torch.randn(tokens, d + 24, generator=g).half().to(DEV)
    for view in (buf[:, :d], buf[:, 3:3 + d], buf[:, 8:8 + d]):   # ldx > d; misaligned base -> scalar loads
        ref = np.zeros((d, d), np.float64)
        O.hessian_add_batch(ref, view.float().cpu().numpy())
        Hacc = _accum(ops, [view], d)
        lo = _lower(d)
        np.testing.assert_allclose(Hacc.cpu().numpy()[lo], ref[lo], rtol=1e-12, atol=1e-12)


def test_full_size_against_fp64_gemm(ops):
    """BASELINE config B fc2 input: 2048 tokens x 8192 features, fp16 -- against the reference's own op (fp64 GEMM)."""
    tokens, d = 2048, 8192
    g = torch.Generator().manual_seed(11)
    x = torch.randn(tokens, d, generator=g).half().to(DEV)
    Hacc = _accum(ops, [x], d)
    x64 = x.double()
    ref = x64.t() @ x64
    H = ops.hessian_finish(Hacc, 1.0)
    want = ref.float()
    del x64
    assert torch.equal(H, H.t())
    mism = (H != want).float().mean().item()
    assert mism < 1e-3
    torch.testing.assert_close(H, want, rtol=1.2e-7, atol=0)
    lo = torch.tril(torch.ones(d, d, dtype=torch.bool, device=DEV))
    err = ((Hacc - ref).abs()[lo]).max().item()
    assert err < 1e-9 * tokens                                   # |x| ~ 1: products O(1), sums O(tokens)


def test_quantmethod_hook_path(ops):
    """QuantMethod.add_batch / post_batch route through K7 on the GPU and keep the reference's bookkeeping."""
    from quip_amd.method import QuantMethod
    d, m = 256, 32
    lin = torch.nn.Linear(d, m).half().to(DEV)
    qm = QuantMethod(lin)
    g = torch.Generator().manual_seed(3)
    X = torch.randn(4, 2, 40, d, generator=g).half().to(DEV)
    for j in range(4):
        qm.add_batch(X[j], None)                                # 3-D input: nsamples += 2
    assert qm.nsamples == 8 and qm._tri
    qm.post_batch()
    x64 = X.reshape(-1, d).double()
    want = ((x64.t() @ x64) / 8).float()
    assert qm.H.dtype == torch.float32 and not qm._tri
    torch.testing.assert_close(qm.H, want, rtol=1.2e-7, atol=0)
    # assigning H from outside (optq_ldlq_equiv.py:24) keeps the dense semantics
    qm2 = QuantMethod(lin)
    qm2.H = (x64.t() @ x64).clone()
    qm2.nsamples = 8
    qm2.post_batch()
    torch.testing.assert_close(qm2.H, want, rtol=1.2e-7, atol=0)


def test_degenerate(ops):
    Hacc = torch.zeros(32, 32, dtype=torch.float64, device=DEV)
    ops.hessian_accum(Hacc, torch.zeros(0, 32, dtype=torch.float16, device=DEV))
    assert Hacc.abs().sum().item() == 0
    H0 = torch.zeros(0, 0, dtype=torch.float64, device=DEV)
    ops.hessian_accum(H0, torch.zeros(4, 0, dtype=torch.float16, device=DEV))
    assert ops.hessian_finish(H0, 1).shape == (0, 0)
    with pytest.raises(RuntimeError):
        ops.hessian_accum(torch.zeros(4, 4, dtype=torch.float64), torch.zeros(2, 4))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tokens,d", [(1, 16), (37, 96), (300, 520), (1000, 203), (257, 4096)])
def test_fast_mode_error_model(ops, O, dtype, tokens, d):
    """opt-in fast mode: exact products, fp32 runs of 128 tokens, fp64 across runs.  Bound: 128 * 2^-24 of the
    absolute-value product sum (worst case of one run); observed error is orders of magnitude below it."""
    g = torch.Generator().manual_seed(tokens + d)
    x = (torch.randn(tokens, d, generator=g) * torch.linspace(0.1, 3.0, d)).to(dtype)
    ref = np.zeros((d, d), np.float64)
    O.hessian_add_batch(ref, x.float().numpy())
    Hacc = torch.zeros(d, d, dtype=torch.float64, device=DEV)
    ops.hessian_accum(Hacc, x.to(DEV), fast=True)
    ax = np.abs(x.float().numpy().astype(np.float64))
    bound = 128 * 2.0 ** -24 * (ax.T @ ax)
    lo = _lower(d)
    err = np.abs(Hacc.cpu().numpy() - ref)
    assert (err[lo] <= bound[lo] + 1e-300).all()
    dg = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
    assert (err[lo] / dg[lo]).max() < 2e-6                       # single call; shrinks ~1/sqrt(#runs) over a calibration pass
    H = ops.hessian_finish(Hacc, 1.0).cpu().numpy()
    np.testing.assert_array_equal(H, H.T)


def test_fast_mode_through_quantmethod_and_f32_inputs_stay_exact(ops):
    from quip_amd import method
    d = 256
    lin = torch.nn.Linear(d, 8).half().to(DEV)
    X = torch.randn(6, 64, d, generator=torch.Generator().manual_seed(2)).half().to(DEV)
    want = ((X.reshape(-1, d).double().t() @ X.reshape(-1, d).double()) / 6).float()
    method.HESSIAN_FAST = True
    try:
        qm = method.QuantMethod(lin)
        for j in range(6):
            qm.add_batch(X[j:j + 1], None)
        qm.post_batch()
        torch.testing.assert_close(qm.H, want, rtol=2e-6, atol=2e-6 * float(want.diagonal().max()))
        Hacc = torch.zeros(d, d, dtype=torch.float64, device=DEV)
        ops.hessian_accum(Hacc, X[0].float(), fast=True)         # f32 input: silently the exact path
        x64 = X[0].double()
        lo = torch.tril(torch.ones(d, d, dtype=torch.bool, device=DEV))
        assert ((Hacc - x64.t() @ x64).abs()[lo]).max().item() < 1e-10
    finally:
        method.HESSIAN_FAST = False
