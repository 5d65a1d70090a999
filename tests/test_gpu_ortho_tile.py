"""-m gpu: the tiled (many-workgroup) form of the small-batch Kronecker operator (csrc/ortho_tile.hip) against the one-workgroup
kernel it stands in for at decode batch sizes, and against the dense orthogonal matrix in float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _op(n, seed):
    from quip_amd import ops, method
    np.random.seed(seed)
    torch.manual_seed(seed)
    return ops.OrthoOp(method.gen_rand_ortho_butterfly_noblock(n), DEV)


def _dense(op):
    """Q with Q x = apply_rows(x): from the identity through the one-workgroup fp32 kernel"""
    from quip_amd import ops
    eye = torch.eye(op.n, device=DEV)
    old = ops.USE_TILES
    ops.USE_TILES = False
    try:
        keep = ops.OrthoOp.use_split
        ops.OrthoOp.use_split = False
        Q = op.apply_rows(eye).t().contiguous()            # row r of apply_rows(I) = Q e_r = column r of Q
        ops.OrthoOp.use_split = keep
    finally:
        ops.USE_TILES = old
    return Q.double()


@pytest.mark.parametrize("n,pq", [(2048, (64, 32)), (4096, (64, 64)), (8192, (128, 64))])
@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("rows", [1, 3, 8])
def test_tiles_equal_the_single_workgroup_kernel(n, pq, transpose, rows):
    from quip_amd import ops
    op = _op(n, seed=n + rows)
    assert (op.p, op.q) == pq and op.tile_supported and op.tile_ok
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, n, generator=g).to(DEV)
    cs = (0.5 + torch.rand(n, generator=g)).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    res = torch.randn(rows, n, generator=g).to(DEV).half()
    # the two operand sets a decode step has (csrc/ortho_tile.hip): activation side and output side
    cases = {"V": dict(xdt=torch.float16, odt=torch.bfloat16, kw=dict(colscale=cs)),
             "V16": dict(xdt=torch.float16, odt=torch.float16, kw=dict(colscale=cs)),
             "U": dict(xdt=torch.float32, odt=torch.float16, kw=dict(bias=bias, residual=res, relu=True)),
             "U32": dict(xdt=torch.float32, odt=torch.float32, kw=dict(bias=bias))}
    outs = {}
    for tiles in (True, False):
        for name, c in cases.items():
            xi = x.to(c["xdt"])
            out = torch.empty(rows, n, dtype=c["odt"], device=DEV)
            d = op.small_op(xi, out, transpose=transpose, **c["kw"])
            if tiles:
                ops.ortho_tile_ops([d], [op.store_inv(transpose)], rows)
            else:
                ops.ortho_small_ops([d], rows)
            outs[(tiles, name)] = out.clone()
    for name in cases:
        assert torch.equal(outs[(True, name)], outs[(False, name)]), name           # same products, same order: bit-identical
    Q = _dense(op)
    ref = x.double() @ (Q if transpose else Q.t()) + bias.double()
    got = outs[(True, "U32")].double()
    assert float((got - ref).abs().max()) <= 2e-4 * float(ref.abs().max())


def test_tiles_refuse_an_operand_set_they_were_not_compiled_for():
    from quip_amd import ops
    op = _op(2048, seed=3)
    x = torch.randn(1, 2048, device=DEV).half()
    out = torch.empty(1, 2048, dtype=torch.bfloat16, device=DEV)
    d = op.small_op(x, out, bias=torch.zeros(2048, device=DEV))                   # f16 input with a bias: neither side
    with pytest.raises(RuntimeError, match="neither"):
        ops.ortho_tile_ops([d], [op.store_inv(False)], 1)


@pytest.mark.parametrize("n", [2048, 8192])
def test_tiles_with_layernorm_and_three_ops(n):
    """q / k / v share the input: three operators, one launch, LayerNorm folded in"""
    from quip_amd import ops
    opsl = [_op(n, seed=s) for s in (1, 2, 3)]
    g = torch.Generator().manual_seed(n)
    x = torch.randn(2, n, generator=g).to(DEV).half()
    ln = torch.nn.LayerNorm(n).to(DEV).half()
    ln.weight.data = (1 + 0.1 * torch.randn(n, generator=g)).to(DEV).half()
    ln.bias.data = (0.1 * torch.randn(n, generator=g)).to(DEV).half()
    cs = [(0.5 + torch.rand(n, generator=g)).to(DEV) for _ in opsl]
    res = {}
    for tiles in (True, False):
        ops.USE_TILES = tiles
        try:
            outs = [torch.empty(2, n, dtype=torch.bfloat16, device=DEV) for _ in opsl]
            descs = [o.small_op(x, out, colscale=c, ln=(ln.weight, ln.bias, ln.eps)) for o, out, c in zip(opsl, outs, cs)]
            if tiles:
                ops.ortho_tile_ops(descs, [o.store_inv(False) for o in opsl], 2)
            else:
                ops.ortho_small_ops(descs, 2)
            res[tiles] = [o.float() for o in outs]
        finally:
            ops.USE_TILES = True
    for a, b in zip(res[True], res[False]):
        assert float((a - b).abs().max()) <= 2 ** -7 * float(b.abs().max())      # one bf16 ulp: the statistics sum in another order
        assert float((a != b).float().mean()) <= 2e-2


def test_tiles_refuse_other_shapes():
    from quip_amd import _lib
    lib = _lib.load()
    assert lib.quipamd_ortho_apply_tiles_supported(64, 32) == 1 and lib.quipamd_ortho_apply_tiles_supported(128, 64) == 1
    assert lib.quipamd_ortho_apply_tiles_supported(96, 32) == 0
    op = _op(1024, seed=0)                                  # 32 x 32
    assert not op.tile_ok and not op.tile_supported
    x = torch.randn(1, 1024, device=DEV)
    y = op.apply_rows(x)                                    # falls back to the one-workgroup kernel
    assert torch.isfinite(y).all()


# ---- p x 16 operators with a large p (csrc/ortho_bigp.hip; Llama's 11008 = 688 x 16) ------------------------------------------------
@pytest.mark.parametrize("n", [11008, 2048 * 3])                 # 688 x 16;  6144 = 2^11 * 3 -> 192 x 32?  (see the skip)
@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("rows", [1, 4, 9, 16, 40])        # up to ops.BIGP_ROWS: a batch of sequences in the decode engine
def test_bigp_equals_the_general_two_launch_kernel(n, transpose, rows):
    from quip_amd import ops
    op = _op(n, seed=n % 97 + rows)
    if not op.bigp_ok:
        pytest.skip(f"{n} factors as {op.p} x {op.q}")
    assert not op.small_ok
    g = torch.Generator().manual_seed(n + rows)
    x = torch.randn(rows, n, generator=g).to(DEV)
    cs = (0.5 + torch.rand(n, generator=g)).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    res = torch.randn(rows, n, generator=g).to(DEV).half()
    # reference: the general kernel (rows > BIGP_ROWS is never sent to the p x 16 kernel: pad the batch)
    pad = lambda t: torch.cat([t, torch.zeros(ops.BIGP_ROWS + 1 - rows, n, dtype=t.dtype, device=DEV)], 0)
    ref_v = op.apply_rows(pad(x.half()), transpose=transpose, colscale=cs, out_dtype=torch.float32)[:rows]
    ref_u = op.apply_rows(pad(x), transpose=transpose, out_dtype=torch.float32, bias=bias)[:rows]
    got_v = op.apply_rows(x.half(), transpose=transpose, colscale=cs, out_dtype=torch.float32)
    got_u = op.apply_rows(x, transpose=transpose, out_dtype=torch.float32, bias=bias)
    assert float((got_v - ref_v).abs().max()) <= 2e-5 * float(ref_v.abs().max())
    assert float((got_u - ref_u).abs().max()) <= 2e-5 * float(ref_u.abs().max())
    # the fused epilogue: bias + residual + relu, f16 out
    out = torch.empty(rows, n, dtype=torch.float16, device=DEV)
    d = op.small_op(x, out, transpose=transpose, bias=bias, residual=res, relu=True)
    ops.ortho_apply_ops([(op, d, transpose)], rows)
    want = torch.relu(ref_u + res.float())
    assert float((out.float() - want).abs().max()) <= 2e-3 * float(want.abs().max())


def test_bigp_silu_gate_input_form():
    """activation side of Llama's down_proj: the kernel forms silu(gate) * up on load (both f16, rounded like the two torch launches)"""
    from quip_amd import ops
    n, rows = 11008, 2
    op = _op(n, seed=5)
    assert op.bigp_ok and not op.small_ok
    g = torch.Generator().manual_seed(3)
    gate = (2 * torch.randn(rows, n, generator=g)).to(DEV).half()
    up = torch.randn(rows, n, generator=g).to(DEV).half()
    cs = (0.5 + torch.rand(n, generator=g)).to(DEV)
    out = torch.empty(rows, n, dtype=torch.float32, device=DEV)
    ops.ortho_apply_ops([(op, op.small_op(gate, out, colscale=cs, residual=up, relu=True), False)], rows)
    h = torch.nn.functional.silu(gate) * up                                      # f16, as the Llama block computes it
    want = op.apply_rows(h, colscale=cs, out_dtype=torch.float32)
    assert float((out - want).abs().max()) <= 1e-3 * float(want.abs().max())    # an f16 ulp of h here and there (__expf vs expf)
