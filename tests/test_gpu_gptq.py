"""-m gpu: OPTQ / GPTQ rounding on the K4 machinery (quipamd_gptq_round) against the oracle's restatement of the
reference loop (oracle/quip_oracle.py gptq_round, gptq.py:51-93) and against the reference-order torch loop that
quip_amd.gptq keeps as its general path."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fixture(m, d, bits, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(d, d, generator=g) / d ** 0.5
    sv = torch.arange(1, d + 1, dtype=torch.float32) ** -0.75
    X = (torch.randn(2 * d, d, generator=g) * sv) @ A                      # correlated Hessian
    H = X.T @ X / (2 * d)
    H = H + 0.01 * H.diag().mean() * torch.eye(d)
    W = 0.02 * torch.randn(m, d, generator=g)
    maxq = 2 ** bits - 1
    lo, hi = W.min(1).values.clamp(max=0), W.max(1).values.clamp(min=0)
    scale = (hi - lo) / maxq
    zero = torch.round(-lo / scale)
    return W, H, scale, zero, maxq


def _proxy(dW, H):
    return float(((dW @ H) * dW).sum())


@pytest.mark.parametrize("m,d,bits", [(16, 64, 4), (40, 128, 2), (33, 272, 4), (64, 512, 2)])
def test_matches_oracle_loop(m, d, bits):
    from quip_amd import ops
    from oracle import quip_oracle as O
    W, H, scale, zero, maxq = _fixture(m, d, bits, seed=m + d)
    Qo, codes_o = O.gptq_round(W.numpy(), H.numpy(), scale.numpy(), zero.numpy(), maxq)
    Hd = H.to(DEV)
    Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hd)), upper=True)
    wg = (W / scale[:, None] + zero[:, None]).to(DEV).contiguous()      # unclamped grid coordinates (quant.py:6-8)
    codes = ops.gptq_round(wg, Hinv.contiguous(), bits)
    mism = (codes.cpu().numpy() != codes_o).mean()
    assert mism <= 2e-3, mism                                    # fp-order flips only (cf. LDLQ vs round_ldl: 1e-3 class)
    Qk = ops.codes_to_weight(codes, 'a', scale.to(DEV), zero.to(DEV), maxq, out_dtype=torch.float32).cpu()
    p_k, p_o = _proxy(Qk - W, H), _proxy(torch.from_numpy(Qo) - W, H)
    p_near = _proxy(torch.from_numpy(O.quantize_qfna(W.numpy(), scale.numpy()[:, None], zero.numpy()[:, None], maxq)) - W, H)
    assert abs(p_k - p_o) <= 2e-2 * p_o and p_k < 0.9 * p_near


def test_gptq_class_kernel_path_equals_reference_order_loop():
    """GPTQ.fasterquant through the kernel vs the same class forced onto the column loop (what the reference runs)."""
    from quip_amd import gptq as G, quant as Q
    d, m, bits = 256, 48, 4
    W, H, _, _, _ = _fixture(m, d, bits, seed=5)
    outs = {}
    for use in (True, False):
        G.USE_KERNEL = use
        try:
            lin = torch.nn.Linear(d, m, bias=False).to(DEV).half()
            lin.weight.data = W.to(DEV).half()
            meth = G.GPTQ(lin)
            meth.quantizer = Q.Quantizer()
            meth.quantizer.configure(bits, perchannel=True, sym=False, qfn='a', mse=False)
            meth.H = H.clone().to(DEV)
            meth.preproc(preproc_gptqH=True, percdamp=.01)
            meth.fasterquant()
            outs[use] = (lin.weight.data.float().clone(), meth.error)
        finally:
            G.USE_KERNEL = True
    mism = (outs[True][0] != outs[False][0]).float().mean().item()
    assert mism <= 2e-3, mism
    assert abs(outs[True][1] - outs[False][1]) <= 2e-2 * outs[False][1]


def test_full_size_is_one_launch_fast():
    """OPT-1.3B fc2 shape (2048 x 8192): the kernel path finishes in milliseconds and keeps the grid invariant."""
    import time
    from quip_amd import ops
    m, d, bits = 2048, 8192, 2
    g = torch.Generator().manual_seed(0)
    X = torch.randn(d + 256, d, generator=g).to(DEV)
    H = X.T @ X / (d + 256)
    H += 0.01 * H.diag().mean() * torch.eye(d, device=DEV)
    Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True).contiguous()
    wg = (torch.rand(m, d, generator=g) * 3.6 - 0.3).to(DEV)
    ops.gptq_round(wg, Hinv, bits)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    codes, err = ops.gptq_round(wg, Hinv, bits, return_err=True)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.2
    assert int(codes.max()) <= 3
    # first column (no feedback yet): plain rounding of the grid coordinate
    assert torch.equal(codes[:, 0].float(), torch.clamp(torch.floor(wg[:, 0] + 0.5), 0, 3))
    # residuals stay within half a grid step unless clamped
    inside = (codes > 0) & (codes < 3)
    assert float(err[inside].abs().max()) <= 0.5 + 1e-5


def test_gptq_class_against_reference_golden():
    """GPTQ.fasterquant (kernel path) against the reference's own run (tests/golden/rounders.npz: gptq.py:19-115 on CPU)."""
    from quip_amd import gptq as G, quant as Q
    g = load_golden("rounders")
    W0 = torch.from_numpy(g["gptq_W0"].copy())
    m, d = W0.shape
    lin = torch.nn.Linear(d, m, bias=False).to(DEV)
    lin.weight.data = W0.to(DEV)
    meth = G.GPTQ(lin)
    meth.quantizer = Q.Quantizer()
    meth.quantizer.configure(4, perchannel=True, sym=False, qfn='a', mse=False)
    meth.H = torch.from_numpy(g["H"].copy()).to(DEV)
    meth.preproc(preproc_gptqH=True, percdamp=.01)
    meth.fasterquant()
    got = lin.weight.data.float().cpu().numpy()
    assert np.mean(got != g["gptq_w4_Q"]) <= 5e-3
    assert abs(meth.error - float(g["gptq_w4_error"])) <= 5e-3 * float(g["gptq_w4_error"])
