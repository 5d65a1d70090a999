"""-m gpu: csrc/decode_head.hip -- the two ends of a decode step as one launch each.

quipamd_decode_head against the same chain in fp64 from the packed layer's own tensors and torch's norm / matmul:
    t = U^T y + bias + residual (rounded to fp16: the residual stream),  h = LayerNorm / RMSNorm (t) (fp16),  logits = W h (fp16)
Gates: logits within 2e-3 (relative l2; the pass runs on fp16 factors), and every logit within 2 fp16 ulps of the largest one plus
that relative error.  The argmax partials + quipamd_decode_embed pick EXACTLY torch.argmax of the logits the launch wrote (ties: smallest
index); the embedding sum is bit-identical to torch's fp16 add."""
import numpy as np
import pytest
import torch

from test_gpu_decode_fused import _layer, _dense, _norm64, _LN, _RMS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,norm,bs,vocab,y32", [(2048, "ln", 1, 50272, False), (2048, "ln", 3, 1000, False), (4096, "rms", 1, 32000, True),
                                                 (4096, "rms", 4, 777, True), (2048, "rms", 2, 512, True), (4096, "ln", 2, 4099, False)])
def test_head_matches_the_chain_in_fp64(n, norm, bs, vocab, y32):
    from quip_amd import ops
    from quip_amd.quant import fused_head, fused_head_ok
    ql, _ = _layer(512, n, 31 + n % 7 + bs, bias=(norm == "ln"))
    torch.manual_seed(vocab + bs)
    g = (1 + 0.1 * torch.randn(n, device=DEV)).half()
    b = (0.05 * torch.randn(n, device=DEV)).half()
    ln_mod = _LN(g, b, 1e-5) if norm == "ln" else _RMS(g, 1e-5)
    ln64 = (g, b, 1e-5) if norm == "ln" else (g, None, 1e-5)
    assert fused_head_ok(ql, bs, ln_mod)
    W = (0.02 * torch.randn(vocab, n, device=DEV)).half()
    y = torch.randn(bs, n, device=DEV) * 0.5
    y = y if y32 else y.half()
    res = torch.randn(bs, n, device=DEV).half()
    logits = torch.full((bs, vocab), float("nan"), dtype=torch.float16, device=DEV)
    pv = torch.full((bs, ops.HEAD_PARTS), float("-inf"), device=DEV)
    pi = torch.full((bs, ops.HEAD_PARTS), -1, dtype=torch.int32, device=DEV)
    pos = torch.tensor([5], dtype=torch.int64, device=DEV)
    fused_head(ql, ql.to_zt(y), res, ln_mod, W, logits, pv, pi, pos_inc=pos)
    assert int(pos) == 6 and not torch.isnan(logits).any()
    t64 = y.half().double() @ _dense(ql.U, transpose=True).t() + (0 if ql.bias is None else ql.bias.half().double()) + res.double()
    h64 = _norm64(t64.half().double(), ln64)
    want = h64.half().double() @ W.double().t()
    rel = float((logits.double() - want).norm() / want.norm())
    assert rel <= 2e-3, rel
    assert float((logits.double() - want).abs().max()) <= (2 * 2.0 ** -11 + 2e-3) * float(want.abs().max())
    # the token: the embed launch of the next step reduces the partials
    tok = (0.02 * torch.randn(vocab, n, device=DEV)).half()
    ptab = (0.02 * torch.randn(64, n, device=DEV)).half()
    ids = torch.full((bs,), 3, dtype=torch.int64, device=DEV)
    x = torch.empty(bs, n, dtype=torch.float16, device=DEV)
    ops.decode_embed(tok, ids, x, pos_table=ptab, pos=pos, pos_offset=2, part_val=pv, part_idx=pi)
    assert torch.equal(ids, logits.float().argmax(-1))           # torch.argmax: first maximal index
    assert torch.equal(x, tok[ids] + ptab[int(pos) + 2])


def test_head_without_an_operator_and_embed_without_partials():
    from quip_amd import ops
    n, vocab, bs = 2048, 3001, 2
    torch.manual_seed(1)
    g = (1 + 0.1 * torch.randn(n, device=DEV)).half()
    b = (0.05 * torch.randn(n, device=DEV)).half()
    W = (0.02 * torch.randn(vocab, n, device=DEV)).half()
    x = torch.randn(bs, n, device=DEV).half()
    logits = torch.empty(bs, vocab, dtype=torch.float16, device=DEV)
    ops.decode_head(W, logits, g, b, 1e-5, x=x)
    want = torch.nn.functional.layer_norm(x.float(), (n,), g.float(), b.float(), 1e-5).half().double() @ W.double().t()
    assert float((logits.double() - want).norm() / want.norm()) <= 1e-3
    # partials that were never written (-1): the caller's ids stand; no position table: the token embedding alone
    pv = torch.full((bs, ops.HEAD_PARTS), float("-inf"), device=DEV)
    pi = torch.full((bs, ops.HEAD_PARTS), -1, dtype=torch.int32, device=DEV)
    ids = torch.tensor([7, 2999], dtype=torch.int64, device=DEV)
    out = torch.empty(bs, n, dtype=torch.float16, device=DEV)
    ops.decode_embed(W, ids, out, part_val=pv, part_idx=pi)
    assert ids.tolist() == [7, 2999] and torch.equal(out, W[ids])
    # ties: the smallest index wins, across workgroups too
    pv.fill_(1.5)
    pi.copy_(torch.arange(ops.HEAD_PARTS, device=DEV, dtype=torch.int32).flip(0)[None, :] * 10 + 4)
    ops.decode_embed(W, ids, out, part_val=pv, part_idx=pi)
    assert ids.tolist() == [4, 4]


def test_head_rejects():
    from quip_amd import ops, _lib
    W = torch.zeros(100, 1024, dtype=torch.float16, device=DEV)
    with pytest.raises(_lib.QuipAmdError):                             # n = 1024
        ops.decode_head(W, torch.zeros(1, 100, dtype=torch.float16, device=DEV), torch.ones(1024, dtype=torch.float16, device=DEV), None, 1e-5,
                        x=torch.zeros(1, 1024, dtype=torch.float16, device=DEV))
