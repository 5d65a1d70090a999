"""-m gpu: the second-generation K2 kernels (csrc/dqgemm_v2.h: "h", "s", "mb"), every one FORCED through
quipamd_dequant_gemm_cfg, bf16 and fp16 activations, against the oracle (fp64 of the reference formula,
quant.py:13-14 / 6-8 / 222-233, on the same 16-bit-rounded x).  Tolerance 1e-3 relative on fp32 output (BASELINE.json
north_star); 16-bit outputs add their own rounding (bf16 2^-9, fp16 2^-12 per element)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_F32 = 1e-3
TOL_16 = {torch.bfloat16: 3e-3, torch.float16: 1e-3}
FAM_OLD, FAM_H, FAM_S, FAM_MB, FAM_PF = 1, 2, 3, 4, 5


@pytest.fixture(scope="module")
def ops():
    from quip_amd import ops as o
    return o


@pytest.fixture(scope="module")
def O():
    from oracle import quip_oracle
    return quip_oracle


def _round16(x, dt):
    t = torch.from_numpy(x).to(dt)
    return t, t.float().numpy()


def _case(O, m, d, bs, bits, qfn, dt, seed):
    rng = np.random.default_rng(seed)
    maxq = 2 ** bits - 1
    W = (0.02 * rng.standard_normal((m, d))).astype(np.float32)
    xt, x = _round16(rng.standard_normal((bs, d)).astype(np.float32), dt)
    if qfn == "b":
        scale = O.qfnb_scale(W)
        codes = np.clip(np.round(((W / scale + 1) / 2) * maxq), 0, maxq).astype(np.uint8)
        zero = None
    else:
        scale, zero = O.find_params_qfna(W, bits)
        codes = np.clip(np.round(W / scale) + zero, 0, maxq).astype(np.uint8)
    return xt, x, codes, scale, zero, maxq


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def _run(ops, O, m, d, bs, bits, qfn, dt, cfg, seed=0, bias=True, check16=True):
    xt, x, codes, scale, zero, maxq = _case(O, m, d, bs, bits, qfn, dt, seed)
    rng = np.random.default_rng(seed + 1)
    b = rng.standard_normal(m).astype(np.float32) if bias else None
    y_ref = O.dequant_linear(x, codes, qfn, scale, zero, maxq, b)
    qs = ops.pack(torch.from_numpy(codes).to(DEV), bits, ops.LAYOUT_STREAM)
    sc = torch.tensor(np.asarray(scale, np.float32).reshape(-1))
    zr = None if zero is None else torch.from_numpy(zero)
    bt = None if b is None else torch.from_numpy(b)
    y32 = ops.dequant_gemm(xt.to(DEV), qs, bits, qfn, sc, zr, bt, out_dtype=torch.float32, cfg=cfg)
    r32 = _rel(y32.cpu().numpy().astype(np.float64), y_ref)
    assert r32 <= TOL_F32, (cfg, r32)
    if check16:
        y16 = ops.dequant_gemm(xt.to(DEV), qs, bits, qfn, sc, zr, bt, cfg=cfg)
        assert y16.dtype == dt
        assert _rel(y16.float().cpu().numpy().astype(np.float64), y_ref) <= TOL_16[dt], cfg
    return r32


H_CFGS = {2: [(FAM_H, 8, 1), (FAM_H, 8, 2), (FAM_H, 4, 4)], 4: [(FAM_H, 8, 2), (FAM_H, 8, 4)]}


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("bits", [2, 4])
@pytest.mark.parametrize("qfn", ["a", "b"])
@pytest.mark.parametrize("m,d,bs", [(16, 256, 1), (64, 512, 4), (128, 1024, 16), (48, 768, 9), (256, 2048, 8), (4096, 4096, 16),
                                    (2048, 2048, 3), (176, 4096, 11)])
def test_h_kernel(ops, O, dt, bits, qfn, m, d, bs):
    """one-pass kernel: every (waves, chunks per wave) instantiation whose LDS holds the layer's K, exact fit and ragged
    K, full and half (bs <= 8) slabs."""
    nkc = d // (512 // bits)
    ran = 0
    for cfg in H_CFGS[bits]:
        if nkc > cfg[1] * cfg[2]:
            continue
        _run(ops, O, m, d, bs, bits, qfn, dt, cfg, seed=m + d + bs)
        ran += 1
    assert ran


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("qfn", ["a", "b"])
@pytest.mark.parametrize("m,d,bs", [(2048, 8192, 1), (2048, 8192, 8), (96, 8192, 3), (512, 6144, 5), (64, 4352, 2), (8192, 8192, 4),
                                    (2048, 8192, 16), (2048, 8192, 9), (1024, 6144, 12)])      # 9..16 rows: two half-batch problems in one launch
def test_half_slab_one_pass_kernel_up_to_d_8192(ops, O, dt, qfn, m, d, bs):
    """round 6: at bs <= 8 a slab of the one-pass kernel holds 8 rows (4 KiB per 256-column chunk), so 32 chunks -- d = 8192, OPT's fc2 in a
    blocked-operator decode step -- fit one workgroup's LDS: the default heuristic takes 16 < chunks <= 32 there (exact fit and ragged K); 9..16
    rows of up to 128 row tiles run it as two half-batch problems sharing the weights in one grouped launch."""
    _run(ops, O, m, d, bs, 2, qfn, dt, None, seed=m + d + bs)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("bits,qfn", [(2, "b"), (2, "a"), (4, "a"), (4, "b")])
@pytest.mark.parametrize("m,d,bs", [(112, 256, 1), (224, 512, 16), (16, 768, 5), (1808, 1024, 16), (4096, 2304, 7), (336, 7168, 16)])
def test_s_kernel(ops, O, dt, bits, qfn, m, d, bs):
    """weight-stream kernel: row tiles not a multiple of the workgroup's, odd stage counts under the k-split (d = 768,
    2304), one stage only (d = 256), every instantiation."""
    for cfg in [(FAM_S, 7, 2), (FAM_S, 4, 2), (FAM_S, 8, 1), (FAM_S, 2, 4), (FAM_S, 1, 8)]:     # (2, 4), (1, 8): round 5, short and wide
        _run(ops, O, m, d, bs, bits, qfn, dt, cfg, seed=m + d + bs, check16=(cfg[1] == 7))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("bits,qfn", [(2, "b"), (2, "a"), (4, "a")])
@pytest.mark.parametrize("m,d,bs", [(256, 256, 17), (128, 512, 33), (512, 1024, 128), (1040, 768, 100), (3072, 512, 300), (64, 2048, 64)])
def test_mb_kernel(ops, O, dt, bits, qfn, m, d, bs):
    """batched kernel: ragged batch and row counts (workgroup tiles 256 x 128 and 128 x 64), more row blocks than 8
    (the XCD-aware block order), both tile shapes."""
    for cfg in [(FAM_MB, 44), (FAM_MB, 22), (FAM_MB, 45), (FAM_MB, 23)] + ([(FAM_MB, 48), (FAM_MB, 49), (FAM_MB, 46), (FAM_MB, 47)] if bits == 2 else []):     # 45 / 23 (round 5): four loader waves; 48 / 49: 4 x 8 tiles per wave; 47: the 32x32x16 form
        _run(ops, O, m, d, bs, bits, qfn, dt, cfg, seed=m + d + bs, check16=(cfg[1] == 44))


@pytest.mark.parametrize("cfg", [(FAM_H, 8, 2), (FAM_S, 7, 2), (FAM_MB, 22)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_one_hot_weights_detect_transposes(ops, O, cfg, dt):
    """asymmetric structure: code 3 only at (r, k = perm[r]) -> y[b, r] picks out x[b, perm[r]] (a transposed operand or
    a wrong k order inside a tile cannot pass)."""
    m, d, bits = 64, 512, 2
    bs = 48 if cfg[0] == FAM_MB else 16
    rng = np.random.default_rng(5)
    codes = np.zeros((m, d), dtype=np.uint8)
    cols = rng.permutation(d)[:m]
    codes[np.arange(m), cols] = 3
    xt, x = _round16(rng.standard_normal((bs, d)).astype(np.float32), dt)
    y_ref = 3.0 * x[:, cols]                                      # qfn a with scale 1, zero 0: What = q
    qs = ops.pack(torch.from_numpy(codes).to(DEV), bits, ops.LAYOUT_STREAM)
    y = ops.dequant_gemm(xt.to(DEV), qs, bits, "a", torch.ones(m), torch.zeros(m), None, out_dtype=torch.float32, cfg=cfg)
    # 1e-4: the multi-exponent dequantisation (offsets up to 128) costs ~1e-5 of cancellation; a transposed operand costs O(1)
    assert _rel(y.cpu().numpy().astype(np.float64), y_ref) <= 1e-4


def test_accumulate_contract_and_determinism(ops, O):
    """y += result (quant.py:226-230) through the new kernels; two identical calls give identical bits."""
    for cfg, (m, d, bs) in [((FAM_H, 8, 2), (256, 2048, 16)), ((FAM_S, 7, 2), (224, 1024, 6)), ((FAM_MB, 44), (512, 512, 130))]:
        xt, x, codes, scale, zero, maxq = _case(O, m, d, bs, 2, "b", torch.float16, seed=m)
        y0 = np.random.default_rng(2).standard_normal((bs, m)).astype(np.float32)
        y_ref = y0 + O.dequant_linear(x, codes, "b", scale, None, maxq, None)
        qs = ops.pack(torch.from_numpy(codes).to(DEV), 2, ops.LAYOUT_STREAM)
        sc = torch.tensor([float(scale)])
        y = torch.from_numpy(y0.copy()).to(DEV)
        ops.dequant_gemm(xt.to(DEV), qs, 2, "b", sc, None, None, out=y, accumulate=True, cfg=cfg)
        assert _rel(y.cpu().numpy().astype(np.float64), y_ref) <= TOL_F32
        a = ops.dequant_gemm(xt.to(DEV), qs, 2, "b", sc, None, None, out_dtype=torch.float32, cfg=cfg)
        b = ops.dequant_gemm(xt.to(DEV), qs, 2, "b", sc, None, None, out_dtype=torch.float32, cfg=cfg)
        assert torch.equal(a, b)


def test_unsupported_forced_kernels_fail_loudly(ops):
    from quip_amd._lib import QuipAmdError
    codes = torch.randint(0, 4, (64, 8192), dtype=torch.uint8).to(DEV)
    qs = ops.pack(codes, 2, ops.LAYOUT_STREAM)
    x = torch.zeros(4, 8192, dtype=torch.bfloat16, device=DEV)
    sc = torch.tensor([0.05])
    with pytest.raises(QuipAmdError):                                  # K does not fit one pass of LDS
        ops.dequant_gemm(x, qs, 2, 'b', sc, None, None, cfg=(FAM_H, 8, 2))
    with pytest.raises(QuipAmdError):                                  # bs > 16 is not a stream-kernel shape
        ops.dequant_gemm(torch.zeros(40, 8192, dtype=torch.bfloat16, device=DEV), qs, 2, 'b', sc, None, None, cfg=(FAM_S, 7, 2))
    with pytest.raises(QuipAmdError):
        ops.dequant_gemm(x, qs, 2, 'b', sc, None, None, cfg=(FAM_S, 5, 5))


def _sampled_rows_case(ops, m, d, bs, bits, dt, seed):
    """full-size layer, codes drawn on the GPU, reference on a sample of rows (every row of the first / a middle / the
    last tile + a stride): the size the oracle cannot do whole in seconds."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    codes = torch.randint(0, 2 ** bits, (m, d), generator=g, dtype=torch.uint8)
    x = torch.randn(bs, d, generator=g).to(dt)
    rows = np.unique(np.concatenate([np.arange(16), np.arange(m // 32 * 16, m // 32 * 16 + 16), np.arange(m - 16, m), np.arange(0, m, 97)]))
    s = 0.05
    maxq = 2 ** bits - 1
    What = ((codes[rows].double() / maxq) * 2 - 1) * s
    y_ref = (x.double() @ What.T).numpy()
    qs = ops.pack(codes.to(DEV), bits, ops.LAYOUT_STREAM)
    return x.to(DEV), qs, torch.tensor([s]), rows, y_ref


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,d", [(7168, 7168), (28672, 7168), (7168, 28672)])
@pytest.mark.parametrize("bs", [1, 16, 256])
def test_opt30b_shapes_default_heuristic(ops, m, d, bs, dt):
    """BASELINE configs[4] shapes (OPT-30B: 7168^2, fc1 28672 x 7168, fc2 7168 x 28672), w2 qfn b, whatever kernel the
    heuristic picks, on sampled rows."""
    x, qs, sc, rows, y_ref = _sampled_rows_case(ops, m, d, bs, 2, dt, seed=m + d + bs)
    y = ops.dequant_gemm(x, qs, 2, "b", sc, None, None, out_dtype=torch.float32, m=m)
    assert _rel(y[:, rows].cpu().numpy().astype(np.float64), y_ref) <= TOL_F32


@pytest.mark.parametrize("m,d,bs,cfg", [(8192, 8192, 16, (FAM_S, 7, 2)), (8192, 8192, 64, (FAM_MB, 44)), (8192, 8192, 64, (FAM_OLD,)),
                                        (8192, 8192, 16, (FAM_OLD,)), (28672, 7168, 16, (FAM_S, 4, 2)), (4096, 4096, 2048, (FAM_MB, 44))])
def test_large_shapes_forced(ops, m, d, bs, cfg):
    """K2 at d = 8192 with bs > 8 and at prefill size vs the oracle formula, new and round-1 kernels."""
    x, qs, sc, rows, y_ref = _sampled_rows_case(ops, m, d, bs, 2, torch.bfloat16, seed=m + bs)
    y = ops.dequant_gemm(x, qs, 2, "b", sc, None, None, out_dtype=torch.float32, m=m, cfg=cfg)
    assert _rel(y[:, rows].cpu().numpy().astype(np.float64), y_ref) <= TOL_F32


def test_the_prefill_lab_family_left_the_library(ops):
    """family 5 (scripts/dqgemm_pf_lab.hip: correct, slower than mb at every shape) is refused, not silently rerouted"""
    from quip_amd import _lib
    x = torch.zeros(256, 256, dtype=torch.bfloat16, device=DEV)
    qs = torch.zeros(256 * 256 * 2 // 32, dtype=torch.int32, device=DEV)
    with pytest.raises(_lib.QuipAmdError):
        ops.dequant_gemm(x, qs, 2, 'b', torch.ones(1, device=DEV), None, None, cfg=(FAM_PF, 21))


def test_fp16_layer_keeps_its_mantissa(ops):
    """An fp16 model's packed layer runs its activations in fp16 (quant.py:226-229 widens x, never narrows it): the
    layer-level error vs the dense fake-quantised layer is <= 2e-3 (the bf16 detour measured 1e-2);
    model.half() leaves the packed layer's float32 buffers alone."""
    from quip_amd import quant as Q
    m, d, bs = 512, 1024, 24
    g = torch.Generator().manual_seed(0)
    W = (0.02 * torch.randn(m, d, generator=g)).to(DEV)
    s = ops.qfnb_scale(W)
    What, codes = ops.quantize(W, 'b', s, None, 3, want_codes=True)
    ql = Q.QuantLinear(d, m, bits=2, qfn='b').pack(codes, s, bias=torch.randn(m, generator=g))
    holder = torch.nn.Sequential(ql).half()
    assert ql.scales.dtype == torch.float32 and ql.bias.dtype == torch.float32
    x = torch.randn(bs, d, generator=g).half().to(DEV)
    got = holder(x)
    assert got.dtype == torch.float16
    want = x.double() @ What.double().T + ql.bias.double()
    assert float((got.double() - want).norm() / want.norm()) <= 2e-3
    xb = x.bfloat16()
    gotb = ql(xb)
    assert gotb.dtype == torch.bfloat16
    wantb = xb.double() @ What.double().T + ql.bias.double()
    assert float((gotb.double() - wantb).norm() / wantb.norm()) <= 5e-3


@pytest.mark.parametrize("form", [74, 72, 81, 4])
@pytest.mark.parametrize("m,groups,bs", [(11008, 2, 16), (4096, 3, 5), (2064, 2, 9)])
def test_grouped_weight_stream_kernel(ops, O, form, m, groups, bs):
    """round 6: dq_sg_kernel -- the weight-stream kernel on 2 / 3 problems of one shape, each with its OWN x~ (gate / up, q / k / v of a
    5..16-row decode step at d = 4096) -- in its three workgroup forms, and the grouped h kernel beside it, against the fp64 formula per
    problem (rows of a sample for the big shape: the oracle's dense product is the slow part).  Ragged row-tile counts included
    (688 = 7 x 98 + 2 tiles; 129 tiles)."""
    d, bits, dt = 4096, 2, torch.float16
    xs, qws, scs, refs = [], [], [], []
    rng = np.random.default_rng(form + m)
    rows = np.sort(rng.choice(m, size=min(m, 512), replace=False))
    for g in range(groups):
        xt, x, codes, scale, zero, maxq = _case(O, m, d, bs, bits, "b", dt, seed=10 * g + bs)
        xs.append(xt.to(DEV))
        qws.append(ops.pack(torch.from_numpy(codes).to(DEV), bits, ops.LAYOUT_STREAM))
        scs.append(torch.tensor(np.asarray(scale, np.float32).reshape(-1)).to(DEV))
        refs.append(O.dequant_linear(x, codes[rows], "b", scale, None, maxq, None))
    outs = [torch.full((bs, m), float("nan"), dtype=torch.float32, device=DEV) for _ in range(groups)]
    ops.dequant_gemm_grouped_config(form)
    try:
        ops.dequant_gemm_grouped(xs, qws, bits, "b", scs, None, outs, m)
        torch.cuda.synchronize()
    finally:
        ops.dequant_gemm_grouped_config(0)
    for g in range(groups):
        got = outs[g].cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all(), (form, g)
        assert _rel(got[:, rows], refs[g]) <= TOL_F32, (form, g)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("qfn", ["b", "a"])
@pytest.mark.parametrize("m,bs", [(4096, 16), (4096, 1), (176, 11), (2048, 8), (16, 5)])
def test_register_x_kernel(ops, O, dt, qfn, m, bs):
    """round 6: dq_hr_kernel (family h, p1 = 44) -- the B fragments as 16-byte global loads into registers behind counted waits, no LDS
    staging of x -- at its only K (d = 4096), every batch size (rows past bs re-read row bs - 1), fp32 and 16-bit outputs, bias, qfn a / b,
    and one-hot weights (a wrong k order inside a chunk or between a wave's chunks cannot pass)."""
    cfg = (FAM_H, 44, 4)
    _run(ops, O, m, 4096, bs, 2, qfn, dt, cfg, seed=m + bs)
    if m == 176:
        rng = np.random.default_rng(9)
        d = 4096
        codes = np.zeros((m, d), dtype=np.uint8)
        perm = rng.permutation(d)[:m]
        codes[np.arange(m), perm] = 3
        xt, x = _round16(rng.standard_normal((bs, d)).astype(np.float32), dt)
        qs = ops.pack(torch.from_numpy(codes).to(DEV), 2, ops.LAYOUT_STREAM)
        sc = torch.tensor([1.0])
        y = ops.dequant_gemm(xt.to(DEV), qs, 2, "b", sc, None, None, out_dtype=torch.float32, cfg=cfg).cpu().numpy().astype(np.float64)
        want = O.dequant_linear(x, codes, "b", np.float32(1.0), None, 3, None)
        assert _rel(y, want) <= TOL_F32
