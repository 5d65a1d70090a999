"""-m gpu: K2 fused dequant-GEMM through the C ABI vs the oracle (fp64 of the reference formula on the same
bf16-rounded x).  Tolerance: 1e-3 relative (BASELINE.json north_star) for fp32 output; bf16 output adds its
own output rounding (2^-9 relative per element), so those cases are gated at 3e-3."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_F32 = 1e-3
TOL_BF16 = 3e-3


@pytest.fixture(scope="module")
def ops():
    from quip_amd import ops as o
    return o


@pytest.fixture(scope="module")
def O():
    from oracle import quip_oracle
    return quip_oracle


def _case(O, m, d, bs, bits, qfn, seed):
    rng = np.random.default_rng(seed)
    maxq = 2 ** bits - 1
    W = (0.02 * rng.standard_normal((m, d))).astype(np.float32)
    x = O.bf16_round(rng.standard_normal((bs, d)).astype(np.float32))
    if qfn == "b":
        scale = O.qfnb_scale(W)
        codes = np.clip(np.round(((W / scale + 1) / 2) * maxq), 0, maxq).astype(np.uint8)
        zero = None
    else:
        scale, zero = O.find_params_qfna(W, bits)
        codes = np.clip(np.round(W / scale) + zero, 0, maxq).astype(np.uint8)
    return W, x, codes, scale, zero, maxq


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.mark.parametrize("bits", [2, 4])
@pytest.mark.parametrize("qfn", ["a", "b"])
@pytest.mark.parametrize("m,d,bs", [(16, 256, 1), (64, 512, 4), (128, 1024, 16), (48, 768, 17), (256, 512, 64),
                                    (2048, 2048, 16)])
def test_dequant_gemm_matches_oracle(ops, O, bits, qfn, m, d, bs):
    W, x, codes, scale, zero, maxq = _case(O, m, d, bs, bits, qfn, seed=m + d + bs + bits)
    rng = np.random.default_rng(1)
    bias = rng.standard_normal(m).astype(np.float32)
    y_ref = O.dequant_linear(x, codes, qfn, scale, zero, maxq, bias)
    qs = ops.pack(torch.from_numpy(codes).to(DEV), bits, ops.LAYOUT_STREAM)
    xd = torch.from_numpy(x).to(DEV).to(torch.bfloat16)
    sc = torch.tensor(np.asarray(scale, np.float32).reshape(-1))
    zr = None if zero is None else torch.from_numpy(zero)
    y32 = ops.dequant_gemm(xd, qs, bits, qfn, sc, zr, torch.from_numpy(bias), out_dtype=torch.float32)
    assert _rel(y32.cpu().numpy().astype(np.float64), y_ref) <= TOL_F32
    y16 = ops.dequant_gemm(xd, qs, bits, qfn, sc, zr, torch.from_numpy(bias), out_dtype=torch.bfloat16)
    assert _rel(y16.float().cpu().numpy().astype(np.float64), y_ref) <= TOL_BF16


def test_one_hot_weights_detect_transposes(ops, O):
    """asymmetric structure: code 3 only at (r, k = perm[r]) -> y[b, r] picks out x[b, perm[r]]."""
    m, d, bs, bits = 64, 512, 16, 2
    rng = np.random.default_rng(5)
    codes = np.zeros((m, d), dtype=np.uint8)
    cols = rng.permutation(d)[:m]
    codes[np.arange(m), cols] = 3
    x = O.bf16_round(rng.standard_normal((bs, d)).astype(np.float32))
    # qfn a with scale 1, zero 0: What = q
    y_ref = 3.0 * x[:, cols]
    qs = ops.pack(torch.from_numpy(codes).to(DEV), bits, ops.LAYOUT_STREAM)
    y = ops.dequant_gemm(torch.from_numpy(x).to(DEV).to(torch.bfloat16), qs, bits, "a", torch.ones(m), torch.zeros(m),
                         None, out_dtype=torch.float32)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref, rtol=0, atol=2e-4 * np.abs(x).sum(1, keepdims=True).max())


def test_accumulate_contract_of_the_reference(ops, O):
    """quant.py:226-230: y pre-filled with bias, kernel accumulates into it (fp32)."""
    W, x, codes, scale, zero, maxq = _case(O, 64, 512, 1, 4, "a", seed=11)
    rng = np.random.default_rng(2)
    bias = rng.standard_normal(64).astype(np.float32)
    qs = ops.pack(torch.from_numpy(codes).to(DEV), 4, ops.LAYOUT_STREAM)
    y = torch.from_numpy(bias.copy()).reshape(1, -1).to(DEV)
    ops.dequant_gemm(torch.from_numpy(x).to(DEV).to(torch.bfloat16), qs, 4, "a", torch.from_numpy(scale),
                     torch.from_numpy(zero), None, out=y, accumulate=True)
    y_ref = O.dequant_linear(x, codes, "a", scale, zero, maxq, bias)
    assert _rel(y.cpu().numpy().astype(np.float64), y_ref) <= TOL_F32
    # and it equals the C restatement of vecquant4matmul on the CANONICAL packing of the same codes
    yc = bias.reshape(1, -1).copy()
    O.packed_matmul_c(x, O.pack_canonical(codes, 4), yc, scale, zero * scale, 4)
    assert _rel(y.cpu().numpy(), yc) <= TOL_F32


def test_headline_shape_linearity_and_dense_agreement(ops, O):
    """BASELINE config A: 4096x4096, w2 qfn b, bs=16.  Size-independent properties + dense fp32 matmul."""
    m = d = 4096
    bs, bits, maxq = 16, 2, 3
    g = torch.Generator().manual_seed(0)
    W = 0.02 * torch.randn(m, d, generator=g)
    Wd = W.to(DEV)
    s = ops.qfnb_scale(Wd)
    _, codes = ops.quantize(Wd, "b", s, None, maxq, want_codes=True)
    qs = ops.pack(codes, bits, ops.LAYOUT_STREAM)
    What = ops.codes_to_weight(codes, "b", s, None, maxq, out_dtype=torch.float32)
    x1 = torch.randn(bs, d, generator=g).to(torch.bfloat16).to(DEV)
    x2 = torch.randn(bs, d, generator=g).to(torch.bfloat16).to(DEV)
    f = lambda x: ops.dequant_gemm(x, qs, bits, "b", s, None, None, out_dtype=torch.float32)
    y1, y2 = f(x1), f(x2)
    dense = x1.float().double() @ What.double().T
    assert float((y1.double() - dense).norm() / dense.norm()) <= TOL_F32
    x3 = (x1.float() * 0.5).to(torch.bfloat16)                     # exact in bf16
    assert float((f(x3) - 0.5 * y1).norm() / y1.norm()) <= 1e-5    # homogeneity
    xs = (x1.float() + x2.float()).to(torch.bfloat16)
    ys_ref = (xs.float().double() @ What.double().T)
    assert float((f(xs).double() - ys_ref).norm() / ys_ref.norm()) <= TOL_F32
    # determinism
    assert torch.equal(f(x1), y1)


@pytest.mark.parametrize("qfn", ["a", "b"])
@pytest.mark.parametrize("m,d,bs,bits", [(4096, 4096, 16, 2), (2048, 2048, 7, 2), (1024, 4096, 16, 4), (64, 512, 3, 2)])
def test_accumulate_split_k_atomics(ops, O, qfn, m, d, bs, bits):
    """Under the accumulate contract small-m shapes take the split-K path (fp32 atomics into the caller's y,
    bias applied once): same answer as the oracle, and as every forced workgroup shape."""
    from quip_amd import _lib
    W, x, codes, scale, zero, maxq = _case(O, m, d, bs, bits, qfn, seed=3 * m + bs)
    rng = np.random.default_rng(4)
    bias = rng.standard_normal(m).astype(np.float32)
    y0 = rng.standard_normal((bs, m)).astype(np.float32)               # what the caller already holds in y
    y_ref = y0 + O.dequant_linear(x, codes, qfn, scale, zero, maxq, bias)
    qs = ops.pack(torch.from_numpy(codes).to(DEV), bits, ops.LAYOUT_STREAM)
    xd = torch.from_numpy(x).to(DEV).to(torch.bfloat16)
    sc = torch.tensor(np.asarray(scale, np.float32).reshape(-1))
    zr = None if zero is None else torch.from_numpy(zero)
    lib = _lib.load()
    try:
        for (rt, nw, split) in [(0, 0, 0), (4, 16, 4), (4, 16, 2), (2, 16, 2), (1, 16, 1), (4, 8, 8), (2, 8, 3)]:
            lib.quipamd_tune_dequant_gemm(rt, 0, nw, split)
            y = torch.from_numpy(y0.copy()).to(DEV)
            ops.dequant_gemm(xd, qs, bits, qfn, sc, zr, torch.from_numpy(bias), out=y, accumulate=True)
            assert _rel(y.cpu().numpy().astype(np.float64), y_ref) <= TOL_F32, (rt, nw, split)
    finally:
        lib.quipamd_tune_dequant_gemm(0, 0, 0, 0)


def test_forced_workgroup_shapes_agree(ops, O):
    """every (row tiles, waves) shape of both K2 kernels gives the oracle's answer (bf16-free fp32 output)."""
    from quip_amd import _lib
    m, d, bs, bits = 512, 2048, 16, 2
    W, x, codes, scale, zero, maxq = _case(O, m, d, bs, bits, "b", seed=9)
    y_ref = O.dequant_linear(x, codes, "b", scale, None, maxq, None)
    qs = ops.pack(torch.from_numpy(codes).to(DEV), bits, ops.LAYOUT_STREAM)
    xd = torch.from_numpy(x).to(DEV).to(torch.bfloat16)
    sc = torch.tensor(np.asarray(scale, np.float32).reshape(-1))
    lib = _lib.load()
    bad = []
    try:
        for cfg in [(1, 0, 16), (2, 0, 16), (4, 0, 16), (1, 0, 8), (2, 0, 8), (4, 0, 8), (1, 0, 4), (2, 0, 4), (4, 0, 4),
                    (1, 0, 2), (1, 0, 1), (2, 0, 2), (8, 0, 16), (8, 0, 8), (1, 0, 8, 200), (2, 0, 8, 200), (4, 0, 16, 200), (8, 0, 16, 200),
                    (1, 1, 16), (2, 1, 16), (4, 1, 16), (1, 1, 8), (2, 1, 8), (4, 1, 8), (1, 1, 4), (4, 1, 4), (1, 2, 8),
                    (2, 2, 8), (1, 2, 4), (2, 2, 4), (1, 4, 4), (2, 4, 4)]:
            lib.quipamd_tune_dequant_gemm(cfg[0], cfg[1], cfg[2], cfg[3] if len(cfg) > 3 else 0)
            y = ops.dequant_gemm(xd, qs, bits, "b", sc, None, None, out_dtype=torch.float32)
            if _rel(y.cpu().numpy().astype(np.float64), y_ref) > TOL_F32:
                bad.append(cfg)
    finally:
        lib.quipamd_tune_dequant_gemm(0, 0, 0, 0)
    assert not bad, f"workgroup shapes with wrong results: {bad}"


@pytest.mark.parametrize("m,d,bs", [(11008, 4096, 5), (4096, 11008, 16)])
def test_llama2_7b_mlp_shapes(ops, O, m, d, bs):
    """BASELINE configs[3]: Llama-2-7B gate/up (11008x4096) and down (4096x11008) projections, w2 qfn b (11008 = 43*256
    chunks: not a power of two, odd chunk count per wave)."""
    W, x, codes, scale, zero, maxq = _case(O, m, d, bs, 2, "b", seed=m + bs)
    y_ref = O.dequant_linear(x, codes, "b", scale, None, maxq, None)
    qs = ops.pack(torch.from_numpy(codes).to(DEV), 2, ops.LAYOUT_STREAM)
    xd = torch.from_numpy(x).to(DEV).to(torch.bfloat16)
    sc = torch.tensor(np.asarray(scale, np.float32).reshape(-1))
    y = ops.dequant_gemm(xd, qs, 2, "b", sc, None, None, out_dtype=torch.float32)
    assert _rel(y.cpu().numpy().astype(np.float64), y_ref) <= TOL_F32
    yacc = torch.zeros(bs, m, device=DEV)
    ops.dequant_gemm(xd, qs, 2, "b", sc, None, None, out=yacc, accumulate=True)
    assert _rel(yacc.cpu().numpy().astype(np.float64), y_ref) <= TOL_F32


@pytest.mark.parametrize("bits,qfn", [(2, "b"), (4, "a")])
@pytest.mark.parametrize("m,d,bs", [(256, 1024, 64), (128, 512, 33), (512, 2048, 100), (1024, 1024, 256)])
def test_batched_kernel_bs_over_16(ops, O, bits, qfn, m, d, bs):
    """bs > 16 goes through the batched kernel (shared x slabs, 64 batch rows per workgroup); every RT variant and the
    default agree with the oracle, ragged batch sizes included."""
    from quip_amd import _lib
    W, x, codes, scale, zero, maxq = _case(O, m, d, bs, bits, qfn, seed=m + d + bs)
    rng = np.random.default_rng(8)
    bias = rng.standard_normal(m).astype(np.float32)
    y_ref = O.dequant_linear(x, codes, qfn, scale, zero, maxq, bias)
    qs = ops.pack(torch.from_numpy(codes).to(DEV), bits, ops.LAYOUT_STREAM)
    xd = torch.from_numpy(x).to(DEV).to(torch.bfloat16)
    sc = torch.tensor(np.asarray(scale, np.float32).reshape(-1))
    zr = None if zero is None else torch.from_numpy(zero)
    lib = _lib.load()
    bad = []
    try:
        for (rt, sp) in [(0, 0), (1, 900), (2, 900), (4, 900)]:
            if rt and (m // 16) % (2 * rt):
                continue
            lib.quipamd_tune_dequant_gemm(rt, 0, 0, sp)
            y = ops.dequant_gemm(xd, qs, bits, qfn, sc, zr, torch.from_numpy(bias), out_dtype=torch.float32)
            if _rel(y.cpu().numpy().astype(np.float64), y_ref) > TOL_F32:
                bad.append((rt, sp))
    finally:
        lib.quipamd_tune_dequant_gemm(0, 0, 0, 0)
    assert not bad, f"batched-kernel variants with wrong results: {bad}"
    y16 = ops.dequant_gemm(xd, qs, bits, qfn, sc, zr, torch.from_numpy(bias), out_dtype=torch.bfloat16)
    assert _rel(y16.float().cpu().numpy().astype(np.float64), y_ref) <= TOL_BF16


def test_three_bit_codes_in_the_four_bit_container_and_quant3linear():
    """--wbits 3: codes 0..7 ride in the 4-bit STREAM container; K2 applies the 3-bit grid (maxq 7) in its epilogue.
    Quant3Linear / make_quant3 keep the reference's names and pack(linear, scales, zeros) protocol (quant.py:173-246)."""
    from quip_amd import ops, quant as Q
    from oracle import quip_oracle as O
    m, d, bs = 64, 256, 5
    g = torch.Generator().manual_seed(0)
    W = 0.02 * torch.randn(m, d, generator=g)
    x = torch.randn(bs, d, generator=g)
    # qfn b with maxq 7
    s = ops.qfnb_scale(W.to(DEV))
    What, codes = ops.quantize(W.to(DEV), 'b', s, None, 7, want_codes=True)
    assert int(codes.max()) <= 7 and int(codes.max()) > 3
    pk = ops.pack(codes, 3, ops.LAYOUT_STREAM)
    assert pk.numel() == m * d * 4 // 32 and torch.equal(ops.unpack(pk, 3, ops.LAYOUT_STREAM, m, d), codes)
    y = ops.dequant_gemm(x.to(DEV).bfloat16(), pk, 3, 'b', s, None, None, out_dtype=torch.float32, m=m)
    want = x.bfloat16().double() @ What.double().cpu().T
    assert float((y.double().cpu() - want).norm() / want.norm()) < 1e-3
    # the reference's layer by name: fake-quantised Linear + per-row grid -> packed forward
    sc, zr = O.find_params_qfna(W.numpy(), 3)
    Wq = torch.from_numpy(O.quantize_qfna(W.numpy(), sc, zr, 7))
    lin = torch.nn.Linear(d, m)
    lin.weight.data = Wq.clone()
    holder = torch.nn.Sequential(lin).to(DEV)
    Q.make_quant3(holder, ["0"])
    assert isinstance(holder[0], Q.Quant3Linear) and holder[0].bits == 3
    holder[0].pack(lin.to(DEV), torch.from_numpy(sc), torch.from_numpy(zr))
    got = holder[0](x.to(DEV))
    want = x.bfloat16().double() @ Wq.double().T + lin.bias.detach().double().cpu()
    assert float((got.double().cpu() - want).norm() / want.norm()) < 2e-3
    codes_ref = np.clip(np.round((Wq.numpy() + zr * sc) / sc), 0, 7)
    np.testing.assert_array_equal(ops.unpack(holder[0].qweight, 3, ops.LAYOUT_STREAM, m, d).cpu().numpy(), codes_ref)
