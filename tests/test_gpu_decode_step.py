"""-m gpu: the assembled decode step (scripts/decode_opt.py: OPT block on packed w2 layers with incoherence operators,
LayerNorm / bias / residual / ReLU folded into the operator launches, single-launch attention, chained hand-overs) against
its dense fp16 twin -- the whole-step logits check that used to live only in the script.
The twin holds U^T What V / s folded into dense fp16 weights, so the difference is activation rounding (the packed path runs
its GEMM activations in the 16-bit dtype of the model) plus fp16 rounding of the folded weights: gated at 1e-2 of the logits'
norm (measured 3.8e-3)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _mod():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "decode_opt.py")
    spec = importlib.util.spec_from_file_location("decode_opt", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_packed_decode_step_matches_its_dense_twin():
    e_plain, e_fused, e_fused_attn, chained_equal, e_v3 = _mod().decode_check(layers=2, bits=2)
    assert e_plain <= 1e-2 and e_fused <= 1e-2 and e_fused_attn <= 1e-2, (e_plain, e_fused, e_fused_attn)
    assert e_v3 is not None and e_v3 <= 1e-2, e_v3          # csrc/decode_fused.hip: 5 launches per block, batch 2
    assert chained_equal, "chained hand-over launches must reproduce the unchained step bit for bit"


def test_packed_w4_decode_step_runs_on_the_fused_launches():
    """round 4: --wbits 4 qfn-b models on the same five launches per block (csrc/decode_fused.hip templated on the container width)"""
    e_plain, e_fused, e_fused_attn, chained_equal, e_v3 = _mod().decode_check(layers=2, bits=4)
    assert e_plain <= 1e-2 and e_fused <= 1e-2 and e_fused_attn <= 1e-2, (e_plain, e_fused, e_fused_attn)
    assert e_v3 is not None and e_v3 <= 1e-2, e_v3


def test_llama_w4_decode_step_runs_on_the_fused_launches():
    """round 4: the 11008-wide tail (csrc/decode_bigp.hip) takes the 4-bit container too: a --wbits 4 Llama decodes on the six launches per block"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "decode_llama.py")
    spec = importlib.util.spec_from_file_location("decode_llama", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    e_plain, e_fused, e_v3 = mod.decode_check(layers=2, bits=4)
    assert e_plain <= 1e-2 and e_fused <= 1e-2, (e_plain, e_fused)
    assert e_v3 is not None and e_v3 <= 1e-2, e_v3


def test_llama_decode_step_matches_its_dense_twin():
    """the Llama block (scripts/decode_llama.py: RMSNorm folded into the activation-side operator launch, one-launch rotary,
    gate / up grouped, 11008-wide operators on the general K3 path) against dense fp16 twins, 4 tokens, batch 2"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "decode_llama.py")
    spec = importlib.util.spec_from_file_location("decode_llama", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    e_plain, e_fused, e_v3 = mod.decode_check(layers=2, bits=2)
    assert e_plain <= 1e-2 and e_fused <= 1e-2, (e_plain, e_fused)
    # small = hidden 2048 (64 x 32 operators, head_dim 128): the fused launches incl. rotary in the attention prologue
    assert e_v3 is not None and e_v3 <= 1e-2, e_v3


def test_rope_inplace_matches_hf_formula():
    import torch
    from quip_amd import ops
    torch.manual_seed(0)
    bs, heads, kvh, hd, maxpos = 3, 8, 4, 128, 64
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    emb = torch.cat([torch.outer(torch.arange(maxpos, dtype=torch.float32), inv)] * 2, -1)
    cos, sin = emb.cos().cuda().contiguous(), emb.sin().cuda().contiguous()
    q = torch.randn(bs, heads * hd, device="cuda").half()
    k = torch.randn(bs, kvh * hd, device="cuda").half()
    pos = torch.tensor([37], device="cuda")

    def ref(x, h):
        xf = x.float().view(bs, h, hd)
        rot = torch.cat((-xf[..., hd // 2:], xf[..., :hd // 2]), -1)             # rotate_half
        return (xf * cos[37] + rot * sin[37]).reshape(bs, h * hd)
    rq, rk = ref(q, heads), ref(k, kvh)
    ops.rope_inplace(q, k, cos, sin, pos, heads, kvh)
    assert float((q.float() - rq).abs().max()) <= 2e-3 and float((k.float() - rk).abs().max()) <= 2e-3
    # a position past the tables never reads them: q / k come back untouched
    q0, k0 = q.clone(), k.clone()
    for bad in (maxpos, maxpos + 1000, -1):
        ops.rope_inplace(q, k, cos, sin, torch.tensor([bad], device="cuda"), heads, kvh)
        assert torch.equal(q, q0) and torch.equal(k, k0)


@pytest.mark.parametrize("n", [2048, 4096])
def test_rmsnorm_fold_in_the_operator_kernels(n):
    """ln = (gamma, None, eps): RMSNorm inside the V-side launch, tiled and one-workgroup kernels, against torch"""
    import numpy as np
    import torch
    from quip_amd import ops, method
    np.random.seed(n)
    torch.manual_seed(n)
    op = ops.OrthoOp(method.gen_rand_ortho_butterfly_noblock(n), "cuda:0")
    x = (torch.randn(2, n, device="cuda") * 3 + 0.5).half()
    g = (1 + 0.1 * torch.randn(n, device="cuda")).half()
    cs = (0.5 + torch.rand(n, device="cuda"))
    xf = x.float()
    normed = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * g.float() * cs
    old = ops.USE_TILES
    try:
        got = {}
        for tiles in (True, False):
            ops.USE_TILES = tiles
            out = torch.empty(2, n, dtype=torch.float32, device="cuda")
            xt = torch.empty(2, n, dtype=torch.bfloat16, device="cuda")
            ops.ortho_apply_ops([(op, op.small_op(x, xt, colscale=cs, ln=(g, None, 1e-5)), False)], 2)
            got[tiles] = xt.float()
        ops.USE_TILES = False
        ref = op.apply_rows(normed.contiguous())
    finally:
        ops.USE_TILES = old
    for tiles in (True, False):
        assert float((got[tiles] - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max()), tiles
