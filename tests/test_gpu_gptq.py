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


@pytest.mark.parametrize("d,qfn", [(200, 'a'), (72, 'a'), (200, 'b'), (136, 'c')])
def test_gptq_class_ragged_width_runs_on_the_kernels(d, qfn):
    """in_features % 16 != 0 (round 6): padded on the right with decoupled identity columns, rounded by the kernels, cut back -- against the
    same class forced onto the reference-order column loop (gptq.py:56-93); the padded columns leave no trace in the state kept for packing"""
    from quip_amd import gptq as G, quant as Q
    m, bits = 48, 4
    W, H, _, _, _ = _fixture(m, d, bits, seed=d)
    outs = {}
    for use in (True, False):
        G.USE_KERNEL = use
        try:
            lin = torch.nn.Linear(d, m, bias=False).to(DEV)
            lin.weight.data = W.to(DEV)
            meth = G.GPTQ(lin)
            meth.quantizer = Q.Quantizer()
            meth.quantizer.configure(bits, perchannel=True, sym=False, qfn=qfn, mse=False)
            meth.H = H.clone().to(DEV)
            meth.preproc(preproc_gptqH=True, percdamp=.01)
            meth.fasterquant()
            outs[use] = (lin.weight.data.float().clone(), meth.error, meth)
        finally:
            G.USE_KERNEL = True
    k, w = outs[True], outs[False]
    assert k[0].shape == (m, d) and bool(torch.isfinite(k[0]).all())
    step = float(w[0].abs().max()) / (2 ** bits - 1)
    assert ((k[0] - w[0]).abs() > 0.25 * step).float().mean().item() <= 1e-2
    assert abs(k[1] - w[1]) <= 2e-2 * w[1]
    if qfn == 'a':
        assert k[2].codes.shape == (m, d)
    if qfn == 'b':
        assert k[2].column_scale.shape[0] == d


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


# ---- groupsize / qfn b, c (gptq.py:69-76, quant.py:10-21) against the reference's own runs -------------------------------------
def _golden_cases():
    import ast
    doc = str(load_golden("gptq_groups")["__doc__"])
    return ast.literal_eval(doc[doc.index("[("):doc.index(")]") + 2])


def _run_case(g, name, d, m, bits, gs, qfn, sym, layer="linear", use_kernel=True):
    from quip_amd import gptq as G, quant as Q
    import transformers
    W0 = torch.from_numpy(g[f"{name}_W0"].copy())
    if layer == "conv1d":
        lin = transformers.Conv1D(m, d).to(DEV)
        lin.weight.data = W0.t().contiguous().to(DEV)
    else:
        lin = torch.nn.Linear(d, m, bias=False).to(DEV)
        lin.weight.data = W0.to(DEV)
    meth = G.GPTQ(lin)
    meth.quantizer = Q.Quantizer()
    meth.quantizer.configure(bits, perchannel=True, sym=sym, qfn=qfn, mse=False)
    meth.H = torch.from_numpy(g[f"H{d}"].copy()).to(DEV)
    if layer == "conv1d":                      # method.preproc does not know Conv1D (method.py:187 indexes the [d, m] weight by column):
        meth.H += .01 * meth.H.diag().mean() * torch.eye(d, device=DEV)          # damp by hand, like preproc_gptqH
        meth.preproc()
    else:
        meth.preproc(preproc_gptqH=True, percdamp=.01)
    G.USE_KERNEL = use_kernel
    try:
        meth.fasterquant(groupsize=gs)
    finally:
        G.USE_KERNEL = True
    return lin.weight.data.float().cpu().numpy(), meth


@pytest.mark.parametrize("case", range(8))
def test_gptq_groups_and_qfn_against_reference_golden(case):
    g = load_golden("gptq_groups")
    name, d, m, bits, gs, qfn, sym, kind = _golden_cases()[case]
    got, meth = _run_case(g, name, d, m, bits, gs, qfn, sym)
    ref = g[f"{name}_Q"]
    kernel = True                                                            # (round 3: qfn b runs on csrc/gptq_qfnb.hip)
    assert hasattr(meth, "group_scale") == (qfn != 'b' and gs != -1)       # the K4 launch served it (not the column walk) ...
    assert hasattr(meth, "column_scale") == (qfn == 'b')                   # ... or the per-column-scale kernel
    # a flipped code moves one weight by a whole grid step; everything after it in the row follows a slightly different path
    step = float(np.abs(ref).max()) / (2 ** bits - 1)
    flipped = np.abs(got - ref) > 0.25 * step
    assert flipped.mean() <= (1e-2 if kernel else 2e-2), flipped.mean()
    assert abs(meth.error - float(g[f"{name}_error"])) <= 2e-2 * float(g[f"{name}_error"])
    if gs != -1 and qfn != 'b':                                               # the quantiser left behind is the LAST group's
        np.testing.assert_allclose(meth.quantizer.scale.cpu().numpy().reshape(-1), g[f"{name}_scale"].reshape(-1), rtol=2e-2)


@pytest.mark.parametrize("gs,qfn,sym,bits", [(64, 'a', False, 3), (32, 'c', True, 4), (128, 'a', False, 2), (16, 'a', True, 4), (-1, 'c', False, 4)])
def test_gptq_groups_kernel_equals_column_walk(gs, qfn, sym, bits):
    """same class, same inputs: the one-launch kernel against the reference-order column walk, both on the GPU in fp32"""
    g = load_golden("gptq_groups")
    outs = {}
    for use in (True, False):
        outs[use], meth = _run_case(g, "w3_g64_a", 384, 40, bits, gs, qfn, sym, use_kernel=use)
    step = float(np.abs(outs[False]).max()) / (2 ** bits - 1)
    flipped = np.abs(outs[True] - outs[False]) > 0.25 * step
    assert flipped.mean() <= 5e-3, flipped.mean()


def test_gptq_groups_first_group_quantiser_is_exact():
    """group 0 sees the untouched weights: its (scale, zero) must equal Quantizer.find_params on W[:, :gs] bit for bit"""
    from quip_amd import ops, quant as Q
    W, H, _, _, _ = _fixture(48, 256, 4, seed=11)
    Hd = H.to(DEV)
    Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hd)), upper=True).contiguous()
    for gs, sym in ((32, False), (64, True), (128, False)):
        Qw, scale, zero, codes = ops.gptq_round_groups(W.to(DEV), Hinv, 4, gs, sym, 'a', return_codes=True)
        qz = Q.Quantizer()
        qz.configure(4, perchannel=True, sym=sym, qfn='a', mse=False)
        qz.find_params(W[:, :gs].to(DEV), weight=True)
        assert torch.equal(scale[:, 0], qz.scale.reshape(-1)) and torch.equal(zero[:, 0], qz.zero.reshape(-1).float())
        # and every emitted weight is scale * (code - zero) of its own group
        sc, zr = scale.repeat_interleave(gs, 1), zero.repeat_interleave(gs, 1)
        assert torch.equal(Qw, sc * (codes.float() - zr))
        assert int(codes.max()) <= 15


def test_gptq_groups_argument_errors():
    from quip_amd import ops
    W = torch.zeros(16, 96, device=DEV)
    Hinv = torch.eye(96, device=DEV)
    with pytest.raises(ValueError, match="groupsize"):
        ops.gptq_round_groups(W, Hinv, 4, 48)
    with pytest.raises(ValueError, match="groupsize"):
        ops.gptq_round_groups(W, Hinv, 4, 64)               # 96 % 64


# ---- the feedback matrix without H^-1 (csrc/trinv.hip) -------------------------------------------------------------------------
@pytest.mark.parametrize("d", [16, 128, 144, 256, 400, 1024, 1424])
def test_unit_upper_inverse(d):
    from quip_amd import ops
    g = torch.Generator().manual_seed(d)
    N = torch.triu(torch.randn(d, d, generator=g) / d ** 0.5, 1)
    junk = torch.tril(torch.full((d, d), 7.0))                                # the lower part and the diagonal must not be read
    X = ops.unit_upper_inverse((N + junk).to(DEV)).cpu().double()
    ref = torch.linalg.inv(torch.eye(d, dtype=torch.float64) + N.double())
    assert float((torch.triu(X) - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert torch.equal(torch.diagonal(X), torch.ones(d, dtype=torch.float64))


@pytest.mark.parametrize("d", [64, 192, 384, 2048])
def test_gptq_feedback_equals_the_inverse_cholesky_route(d):
    """FT from (flip, K8, triangular inverse) against gptq.py:51-54's route in float64"""
    from quip_amd import ops
    _, H, _, _, _ = _fixture(16, d, 4, seed=d)
    H = H + 0.01 * H.diag().mean() * torch.eye(d)
    FT = ops.gptq_feedback(H.to(DEV)).cpu().double()
    Hinv = torch.linalg.cholesky(torch.linalg.inv(H.double()), upper=True)
    ref = ops.gptq_feedback_matrix(Hinv)
    assert float((FT - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    assert float(torch.tril(FT).abs().max()) == 0.0


def test_gptq_feedback_rejects_indefinite_matrix():
    from quip_amd import ops
    H = torch.eye(64, device=DEV)
    H[5, 5] = -1.0
    with pytest.raises(torch.linalg.LinAlgError):
        ops.gptq_feedback(H)


def test_gptq_feedback_full_width_timing():
    """d = 8192 (OPT-1.3B fc2 / OPT-30B attention width class): a few tens of ms, against ~100 ms of the rocSOLVER triple"""
    import time
    from quip_amd import ops
    d = 8192
    g = torch.Generator().manual_seed(0)
    X = torch.randn(d + 512, d, generator=g).to(DEV)
    H = X.T @ X / (d + 512)
    H += 0.01 * H.diag().mean() * torch.eye(d, device=DEV)
    ops.gptq_feedback(H)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    FT = ops.gptq_feedback(H)
    torch.cuda.synchronize()
    t_ours = time.perf_counter() - t0
    t0 = time.perf_counter()
    Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
    torch.cuda.synchronize()
    t_ref = time.perf_counter() - t0
    ref = ops.gptq_feedback_matrix(Hinv)
    print(f"gptq_feedback d={d}: {t_ours * 1e3:.1f} ms; rocSOLVER cholesky/cholesky_inverse/cholesky: {t_ref * 1e3:.1f} ms")
    assert float((FT - ref).abs().max()) <= 2e-3
    assert t_ours < 0.25
