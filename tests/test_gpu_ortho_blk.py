"""-m gpu: csrc/ortho_blk.hip -- the BLOCKED butterfly (method.py:34-35; what the reference's --incoh_processing really selects) on a handful of
rows, with the decoder block's neighbouring elementwise work fused in -- against the same chain evaluated in fp64 from the operator's dense
matrix (built from the generator tuple with torch.einsum in float64: the reference's index form, no kernel of this repo).

Gate: 1e-3 relative l2 per application (measured ~3e-4: fp16 factors; the activations travel as fp16 hi + lo), exact zeros where the
operator's support says so is not claimed.  QuantLinear.forward / packed_forward_fused on blocked operators are covered end to end by
tests/test_gpu_decode_e2e.py (pre_proj_extra = 0)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(params=["one launch per operator", "two launches"], autouse=True)
def _launch_form(request):
    """every test runs in both forms: the second stage's workgroups computing their slice of the first stage in the prologue (default up to
    n = 2048 and 4 rows), and the two stage launches with the fp32 image in between"""
    from quip_amd import ops
    ops.ortho_blocked_config(request.param == "one launch per operator")      # True: wherever the rows fit LDS, not only up to the default n
    yield
    ops.ortho_blocked_config()


def _op(n, seed):
    from quip_amd import ops, method
    np.random.seed(seed)
    torch.manual_seed(seed)
    return ops.OrthoOp(method.gen_rand_ortho_butterfly(n), DEV)


from test_gpu_decode_fused import _dense                   # the operator's dense fp64 matrix from its generator tuple (torch.einsum, no kernel of ours)


@pytest.mark.parametrize("n", [768, 2048, 4096, 8192, 11008])
@pytest.mark.parametrize("rows", [1, 3, 8, 16, 21, 64])          # 16 per workgroup; 21 = a full group + a ragged one; 64 = the limit
def test_blocked_operator_small_rows(n, rows):
    op = _op(n, n + rows)
    assert op.blocked and op.blk_ok
    Q = _dense(op)
    torch.manual_seed(rows)
    for transpose in (False, True):
        for dt in (torch.float16, torch.float32):
            x = torch.randn(rows, n, device=DEV).to(dt)
            cs = 0.5 + torch.rand(n, device=DEV)
            bias = torch.randn(n, device=DEV)
            got = op.apply_rows_blocked(x, transpose=transpose, colscale=cs, bias=bias, out_dtype=torch.float32)
            M = Q.t() if transpose else Q
            want = (x.double() * cs.double()) @ M.t() + bias.double()
            rel = float((got.double() - want).norm() / want.norm())
            assert rel <= 1e-3, (n, rows, transpose, dt, rel)
    # the route QuantLinear.forward takes: apply_rows(..., fast16=True) is this kernel, without it the general fp32 launches
    x = torch.randn(rows, n, device=DEV).half()
    a = op.apply_rows(x, fast16=True, out_dtype=torch.float32)
    b = op.apply_rows(x, out_dtype=torch.float32)
    assert float((a - b).norm() / b.norm()) <= 1e-3
    assert torch.equal(a, op.apply_rows_blocked(x, out_dtype=torch.float32))


@pytest.mark.parametrize("rows", [2, 19])
@pytest.mark.parametrize("n,norm", [(2048, "ln"), (4096, "rms"), (2048, None)])
def test_activation_side_with_the_norm_and_scale_in_front(n, norm, rows):
    """x~ = V (Norm(x) (/) s) in bf16 -- what packed_forward_fused hands to the grouped GEMM"""
    op = _op(n, 5)
    Q = _dense(op)
    torch.manual_seed(n)
    x = (torch.randn(rows, n, device=DEV) * 2 + 0.3).half()
    g = (1 + 0.1 * torch.randn(n, device=DEV)).half()
    b = (0.05 * torch.randn(n, device=DEV)).half()
    cs = 0.5 + torch.rand(n, device=DEV)
    ln = None if norm is None else (g, b, 1e-5) if norm == "ln" else (g, None, 1e-5)
    got = op.apply_rows_blocked(x, colscale=cs, ln=ln, out_dtype=torch.bfloat16)
    xf = x.float()
    if norm == "ln":
        h = torch.nn.functional.layer_norm(xf, (n,), g.float(), b.float(), 1e-5).half()
    elif norm == "rms":
        h = g * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).half()          # HF LlamaRMSNorm
    else:
        h = x
    want = (h.double() * cs.double()) @ Q.t()
    assert got.dtype == torch.bfloat16
    assert float((got.double() - want).norm() / want.norm()) <= 3e-3                        # bf16 output: 2^-9


@pytest.mark.parametrize("rows", [1, 2, 4])
@pytest.mark.parametrize("n", [512, 1024, 2048, 2560, 4096, 8192])
@pytest.mark.parametrize("norm", ["ln", "rms", None])
def test_rows_in_registers_forms(n, norm, rows):
    """round 6: up to 256 x 4 sixteen-byte chunks of input (n a multiple of 512) stay in the threads' registers from the load to the fp32 image in
    LDS -- statistics on the DPP network, gains and scale applied there -- and a stage of <= 4 k-steps is finished by wave 0 alone.  Every
    shape class of the dispatch (1 / 4 chunks per thread, 1 / 8 factor chunks, n8 = 320: five wave slots per row), with an offset mean (the
    shifted statistics), the residual in each dtype and the gate on the input."""
    op = _op(n, 11)
    Q = _dense(op)
    torch.manual_seed(n + rows)
    x = (torch.randn(rows, n, device=DEV) * 1.5 + 3.0).half()
    g = (1 + 0.1 * torch.randn(n, device=DEV)).half()
    b = (0.05 * torch.randn(n, device=DEV)).half()
    cs = 0.5 + torch.rand(n, device=DEV)
    ln = None if norm is None else (g, b, 1e-5) if norm == "ln" else (g, None, 1e-5)
    xf = x.float()
    if norm == "ln":
        h = torch.nn.functional.layer_norm(xf, (n,), g.float(), b.float(), 1e-5).half()
    elif norm == "rms":
        h = g * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).half()
    else:
        h = x
    want = (h.double() * cs.double()) @ Q.t()
    got = op.apply_rows_blocked(x, colscale=cs, ln=ln, out_dtype=torch.float16)
    assert float((got.double() - want).norm() / want.norm()) <= 1.5e-3
    # rows of a batch do not see each other: row 0 alone gives the same bits
    if rows * n // 8 <= 1024:                                             # (beyond, the batch takes the general form and the lone row does not)
        alone = op.apply_rows_blocked(x[:1].contiguous(), colscale=cs, ln=ln, out_dtype=torch.float16)
        assert torch.equal(alone[0], got[0])
    for rdt in (torch.float16, torch.bfloat16, torch.float32):
        res = torch.randn(rows, n, device=DEV).to(rdt)
        bias = 0.1 * torch.randn(n, device=DEV)
        got = op.apply_rows_blocked(x, transpose=True, ln=ln, bias=bias, residual=res, relu=(rdt == torch.float16), out_dtype=torch.float32)
        want = h.double() @ Q + bias.double() + res.double()
        if rdt == torch.float16:
            want = torch.relu(want)
        assert float((got.double() - want).norm() / want.norm()) <= 1.5e-3, rdt
    if norm is None:
        up = torch.randn(rows, n, device=DEV).half()
        got = op.apply_rows_blocked(x, colscale=cs, gate_up=up, out_dtype=torch.float32)
        want = ((torch.nn.functional.silu(x) * up).double() * cs.double()) @ Q.t()
        assert float((got.double() - want).norm() / want.norm()) <= 1e-3


@pytest.mark.parametrize("rows", [2, 18])
def test_output_side_with_bias_residual_relu_and_the_gated_input(rows):
    n = 11008
    op = _op(n, 9)
    Q = _dense(op)
    torch.manual_seed(1)
    y = torch.randn(rows, n, device=DEV)
    bias = 0.1 * torch.randn(n, device=DEV)
    res = torch.randn(rows, n, device=DEV).half()
    got = op.apply_rows_blocked(y, transpose=True, bias=bias, residual=res, relu=True, out_dtype=torch.float16)
    want = torch.relu(y.double() @ Q + bias.double() + res.double())
    assert float((got.double() - want).norm() / want.norm()) <= 1.5e-3
    # silu(gate) * up formed on load (Llama's down_proj input)
    gate = torch.randn(rows, n, device=DEV).half()
    up = torch.randn(rows, n, device=DEV).half()
    cs = 0.5 + torch.rand(n, device=DEV)
    got = op.apply_rows_blocked(gate, colscale=cs, gate_up=up, out_dtype=torch.float32)
    hin = (torch.nn.functional.silu(gate) * up)
    want = (hin.double() * cs.double()) @ Q.t()
    assert float((got.double() - want).norm() / want.norm()) <= 1e-3


def test_blocked_rows_refuses_what_it_cannot_run():
    from quip_amd import _lib, ops
    op = _op(2048, 3)
    with pytest.raises(AssertionError):
        op.apply_rows_blocked(torch.zeros(65, 2048, device=DEV))                 # more than 64 rows: the general launches' job
    lib = _lib.load()
    assert lib.quipamd_ortho_blocked_supported(688, 16) == 1 and lib.quipamd_ortho_blocked_supported(43, 16) == 0
    a = ops.BlkOp()
    a.p, a.q, a.rows = 64, 32, 1
    with pytest.raises(_lib.QuipAmdError):
        _lib.call("quipamd_ortho_blocked_rows", __import__("ctypes").byref(a), None, None)
