"""-m gpu: csrc/gptq_qfnb.hip -- OPTQ with the qfn-b quantiser (every column on its own scale, recomputed from all rows of the updated
column: quant.py:158-160 inside gptq.py:56-93) against the reference's column walk (quip_amd.gptq._column_walk, the same loop in
torch, fp32, on the GPU).  The two differ in summation order only (the column's sum of squares, the far-field products): a weight
whose grid coordinate sits within rounding of a half flips, and everything behind it in its row follows another path.  On these fixtures
the kernel agrees with the fp32 walk code for code (and the fp32 walk with an fp64 walk to 1e-3 at d >= 2048): gates flipped codes <= 2e-3,
proxy loss within 1 %.  The kernel itself is deterministic."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fixture(m, d, seed):
    g = torch.Generator().manual_seed(seed)
    W = (0.02 * torch.randn(m, d, generator=g)).to(DEV)
    X = torch.randn(2 * d + 64, d, generator=g).to(DEV) * (0.5 + torch.rand(d, generator=g).to(DEV))
    H = X.T @ X / X.shape[0]
    H += 0.01 * H.diag().mean() * torch.eye(d, device=DEV)
    return W, H


@pytest.mark.parametrize("m,d,bits", [(40, 384, 2), (24, 128, 4), (256, 512, 2), (2064, 256, 4), (4100, 128, 3), (48, 80, 2), (8200, 256, 2), (2048, 2048, 2)])
def test_kernel_against_the_column_walk(m, d, bits):
    from quip_amd import ops, gptq as G, quant as Q
    W, H = _fixture(m, d, 7 * m + d)
    Qk, cs = ops.gptq_round_qfnb(W.clone(), ops.gptq_feedback(H), bits)
    Qk2, cs2 = ops.gptq_round_qfnb(W.clone(), ops.gptq_feedback(H), bits)
    assert torch.equal(Qk, Qk2) and torch.equal(cs, cs2)              # granules summed in a fixed order
    qz = Q.Quantizer()
    qz.configure(bits, perchannel=True, sym=False, qfn='b', mse=False)
    qz.find_params(W, weight=True)
    Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
    Qw = G._column_walk(W.clone(), Hinv, qz, 128, -1)
    # the first column processed sees the untouched weights: its scale is the reference formula on W[:, 0]
    s0 = 2.4 * W[:, 0].square().mean().sqrt() + 1e-16
    assert abs(float(cs[0]) - float(s0)) <= 1e-6 * float(s0)
    step = 2.0 * cs[None, :] / (2 ** bits - 1)                        # grid step of every column
    flipped = ((Qk - Qw).abs() > 0.25 * step).float().mean().item()
    assert flipped <= 2e-3, flipped
    proxy = lambda Qq: float((((Qq - W) @ H) * (Qq - W)).sum())
    assert abs(proxy(Qk) - proxy(Qw)) <= 1e-2 * proxy(Qw)
    # every emitted weight lies on its column's grid
    t = (Qk / cs[None, :] + 1) / 2 * (2 ** bits - 1)
    assert float((t - t.round()).abs().max()) <= 1e-3 and float(t.min()) >= -1e-3 and float(t.max()) <= 2 ** bits - 1 + 1e-3


def test_fasterquant_takes_the_kernel_for_qfn_b():
    from quip_amd import gptq as G, quant as Q
    m, d = 192, 256
    W, H = _fixture(m, d, 3)
    outs = {}
    for use in (True, False):
        lin = torch.nn.Linear(d, m, bias=False).to(DEV)
        lin.weight.data = W.clone()
        meth = G.GPTQ(lin)
        meth.quantizer = Q.Quantizer()
        meth.quantizer.configure(2, perchannel=True, sym=False, qfn='b', mse=False)
        meth.H = H.clone()
        meth.preproc()
        G.USE_KERNEL = use
        try:
            meth.fasterquant(groupsize=64 if use else -1)            # qfn b ignores what find_params leaves: any groupsize, one kernel
        finally:
            G.USE_KERNEL = True
        assert hasattr(meth, "column_scale") == use
        outs[use] = (lin.weight.data.float().clone(), meth.error)
    assert abs(outs[True][1] - outs[False][1]) <= 2e-2 * outs[False][1]


def test_a_grid_that_is_not_co_resident_is_an_error_not_a_hang():
    """the chain kernel's workgroups wait for each other.  With the test hook launching one workgroup too few, the
    others poll granules nobody writes: the bounded poll gives up, the abort word is raised, ops raises, and GPTQ.fasterquant falls back
    to the column walk (the 'co-resident' route of gptq._kernel_round) -- within seconds, with the device still usable."""
    import time
    from quip_amd import _lib, ops
    torch.manual_seed(3)
    m, d = 256, 256
    W = (torch.randn(m, d, device=DEV) * 0.02).contiguous()
    X = torch.randn(2 * d, d, device=DEV)
    H = X.T @ X / (2 * d) + 0.01 * torch.eye(d, device=DEV)
    FT = ops.gptq_feedback(H)
    good, _ = ops.gptq_round_qfnb(W.clone(), FT, 2)
    ops.gptq_qfnb_debug(short_grid=1, spin_limit=20000)
    try:
        t0 = time.time()
        with pytest.raises(_lib.QuipAmdError, match="co-resident"):
            ops.gptq_round_qfnb(W.clone(), FT, 2)
        assert time.time() - t0 < 30.0
    finally:
        ops.gptq_qfnb_debug(0, 0, 0)
    again, _ = ops.gptq_round_qfnb(W.clone(), FT, 2)                  # the device and the library are fine afterwards
    assert torch.equal(again, good)


@pytest.mark.parametrize("m,d", [(200, 256), (2048, 384), (2100, 256), (4096, 256), (8192, 208), (11008, 128), (16384, 96)])
def test_every_chain_form_gives_the_same_sweep(m, d):
    """round 6: the pipelined chain on one XCD (default up to 16384 rows; 1 / 2 / 4 / 8 chain waves per workgroup, 64-column lazy blocks beyond
    4096 rows), the pipelined chain across the
    XCDs (forced: 1) and the barrier-per-phase chain of rounds 3-5 (forced: 64) differ in the ORDER the column's squares are summed only:
    the same gates as against the column walk, and each form deterministic."""
    from quip_amd import ops
    W, H = _fixture(m, d, m + d)
    FT = ops.gptq_feedback(H)
    outs = {}
    try:
        for form in (0, 1, 2, 64):
            ops.gptq_qfnb_debug(0, 0, form)
            q, cs = ops.gptq_round_qfnb(W.clone(), FT, 2)
            q2, cs2 = ops.gptq_round_qfnb(W.clone(), FT, 2)
            assert torch.equal(q, q2) and torch.equal(cs, cs2), form
            outs[form] = (q, cs)
    finally:
        ops.gptq_qfnb_debug(0, 0, 0)
    assert torch.equal(outs[0][0], outs[2][0])                          # the default at these sizes IS the one-XCD form
    ref, csr = outs[64]
    step = 2.0 * csr[None, :] / 3
    for form in (0, 1):
        flipped = ((outs[form][0] - ref).abs() > 0.25 * step).float().mean().item()
        assert flipped <= 2e-3, (form, flipped)
        assert float((outs[form][1] - csr).abs().max() / csr.abs().max()) <= 1e-4


def test_one_xcd_roll_call_failure_falls_back_without_touching_anything():
    """the one-XCD form's first exchange is its roll call: with a participant missing (test hook) it raises abort word 2 before anything is
    written, ops repeats the sweep across the XCDs -- where the same hook makes it fail for good: the error of the co-residency test; with
    the hook lifted in between (spin limit only) the repeated sweep completes and equals the forced cross-XCD run."""
    from quip_amd import _lib, ops
    m, d = 512, 256
    W, H = _fixture(m, d, 5)
    FT = ops.gptq_feedback(H)
    ops.gptq_qfnb_debug(0, 0, 1)
    try:
        want, _ = ops.gptq_round_qfnb(W.clone(), FT, 2)
    finally:
        ops.gptq_qfnb_debug(0, 0, 0)
    lib = _lib.load()
    wt = W.flip(1).t().contiguous()
    keep = wt.clone()
    qt = torch.full_like(wt, 7.0)
    cs = torch.full((d,), 7.0, device=DEV)
    ws = torch.empty(int(lib.quipamd_gptq_qfnb_workspace_bytes(m, d)), dtype=torch.uint8, device=DEV)
    ops.gptq_qfnb_debug(short_grid=1, spin_limit=20000, force_rows=2)
    try:
        _lib.call("quipamd_gptq_round_qfnb", ops._p(wt), ops._p(FT), 2, ops._p(qt), ops._p(cs), ops._p(ws), m, d, ops._stream())
        off = int(lib.quipamd_gptq_qfnb_info_offset(m, d))
        assert int(ws[off:off + 4].view(torch.int32).item()) == 2
        assert torch.equal(wt, keep) and bool((qt == 7.0).all()) and bool((cs == 7.0).all())
        ops.gptq_qfnb_debug(0, 0, 1)                                    # what ops does next (without the hook)
        _lib.call("quipamd_gptq_round_qfnb", ops._p(wt), ops._p(FT), 2, ops._p(qt), ops._p(cs), ops._p(ws), m, d, ops._stream())
        assert int(ws[off:off + 4].view(torch.int32).item()) == 0
        assert torch.equal(qt.t().flip(1).contiguous(), want)
    finally:
        ops.gptq_qfnb_debug(0, 0, 0)


def test_more_rows_than_one_xcd_holds_take_the_cross_xcd_forms():
    """beyond 16384 rows (512 per workgroup x 32 CUs) the sweep leaves the one-XCD form: 20480 rows run the round-3 chain with 128 rows per
    workgroup (the pipelined cross-XCD form needs <= 256 workgroups of 64 rows), forcing the one-XCD form is an error, and the result is the
    forced 128-row run's bit for bit."""
    from quip_amd import _lib, ops
    m, d = 20480, 128
    W, H = _fixture(m, d, 11)
    FT = ops.gptq_feedback(H)
    q0, c0 = ops.gptq_round_qfnb(W.clone(), FT, 2)
    try:
        ops.gptq_qfnb_debug(0, 0, 128)
        q1, c1 = ops.gptq_round_qfnb(W.clone(), FT, 2)
        ops.gptq_qfnb_debug(0, 0, 2)
        with pytest.raises(_lib.QuipAmdError, match="do not fit one XCD"):
            ops.gptq_round_qfnb(W.clone(), FT, 2)
    finally:
        ops.gptq_qfnb_debug(0, 0, 0)
    assert torch.equal(q0, q1) and torch.equal(c0, c1)
