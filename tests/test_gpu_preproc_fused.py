"""-m gpu: csrc/preproc.hip -- the rescale and trace + ridge chains of QuantMethod.preproc (method.py:140-165) in a few launches, against
the same chains as torch elementwise ops (the form the reference has them in).  Same operations in the same order; what differs is the
order of two fp32 sums (the column sums of squares of W, the trace of H): s and the trace factor agree to fp32 rounding, W to one ulp of
its dtype, H to 1e-6."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _chain(w, H):
    w = w.to(torch.float32)
    H = H / H.abs().max()
    diagH = torch.diag(H).clamp(min=1e-8)
    diagW2 = (w * w).sum(0).clamp(min=1e-8)
    s = (diagH / diagW2).sqrt().sqrt().to(torch.float32).clamp(min=1e-8)
    return w * s[None, :], (H / s[None, :]) / s[:, None], s


@pytest.mark.parametrize("m,d,dtype", [(2048, 2048, torch.float16), (300, 130, torch.float32), (4096, 1024, torch.bfloat16), (64, 4096, torch.float16)])
def test_rescale_matches_the_torch_chain(m, d, dtype):
    from quip_amd import ops
    torch.manual_seed(m + d)
    w = (0.02 * torch.randn(m, d, device=DEV) * (0.2 + torch.rand(d, device=DEV))).to(dtype)
    X = torch.randn(d + 32, d, device=DEV) * (0.1 + torch.rand(d, device=DEV))
    H = (X.T @ X / X.shape[0]).contiguous()
    _, _, s_ref = _chain(w, H)
    w2, H2 = w.clone(), H.clone()
    s = ops.preproc_rescale(w2, H2)
    # s: the column sums of squares are added in another order than torch's reduction -- last-bit differences only
    assert float(((s - s_ref).abs() / s_ref).max()) <= 3e-7
    # everything downstream of s is the reference's operation sequence, IEEE operation by IEEE operation: bit-identical given s
    assert torch.equal(w2, (w.to(torch.float32) * s[None, :]).to(dtype))
    assert torch.equal(H2, ((H / H.abs().max()) / s[None, :]) / s[:, None])


def test_trace_ridge_matches_the_torch_chain():
    from quip_amd import ops
    torch.manual_seed(3)
    for d in (130, 2048):
        X = torch.randn(d + 32, d, device=DEV)
        H = (X.T @ X / X.shape[0]).contiguous()
        ref = H * (d / (torch.trace(H) + 1e-8)) + 1e-2 * torch.eye(d, device=DEV)
        got = ops.preproc_trace_ridge(H.clone(), 1e-2)
        assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_preproc_fused_against_the_torch_form_of_the_method():
    """QuantMethod.preproc with and without the fused chains: same scaleWH / H / weights to rounding, caller's H untouched"""
    from quip_amd import method as M
    torch.manual_seed(5)
    d, m = 512, 256
    X = torch.randn(d + 64, d, device=DEV)
    H0 = (X.T @ X / X.shape[0]).contiguous()
    outs = {}
    for fused in (True, False):
        lin = torch.nn.Linear(d, m, bias=False).to(DEV).half()
        torch.manual_seed(9)
        lin.weight.data = (0.02 * torch.randn(m, d, device=DEV)).half()
        meth = M.QuantMethod(lin)
        meth.H = H0.clone()
        keep = meth.H
        M.FUSED_PREPROC = fused
        try:
            torch.manual_seed(11); import numpy as np; np.random.seed(11)
            meth.preproc(preproc_gptqH=True, percdamp=.01, preproc_rescale=True, preproc_proj=True, preproc_proj_extra=1)
        finally:
            M.FUSED_PREPROC = True
        assert torch.equal(keep, H0)
        outs[fused] = (meth.scaleWH.clone(), meth.H.clone(), lin.weight.data.float().clone())
    sa, Ha, wa = outs[True]
    sb, Hb, wb = outs[False]
    assert float(((sa - sb).abs() / sb).max()) <= 1e-6
    assert float((Ha - Hb).abs().max()) <= 1e-5 * float(Hb.abs().max())
    assert float((wa - wb).abs().max()) <= 2e-3 * float(wb.abs().max())
