"""-m gpu: single-token decode attention (quipamd_decode_attention) against the eager chain it replaces, evaluated in
fp64 (scripts/decode_opt.py Block.forward; HF OPTAttention in the reference's benchmark(), opt.py:431-482)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(q, k, v, kc, vc, pos):
    bs, heads, maxlen, hd = kc.shape
    kc, vc = kc.clone(), vc.clone()
    kc[:, :, pos] = k.view(bs, heads, hd)
    vc[:, :, pos] = v.view(bs, heads, hd)
    s = torch.einsum("bhd,bhtd->bht", q.view(bs, heads, hd).double(), kc[:, :, :pos + 1].double()) / math.sqrt(hd)
    o = torch.einsum("bht,bhtd->bhd", torch.softmax(s, -1), vc[:, :, :pos + 1].double())
    return o.reshape(bs, heads * hd), kc, vc


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bs,heads,hd,maxlen,pos", [(1, 32, 64, 2048, 0), (1, 32, 64, 2048, 1), (1, 32, 64, 2048, 130),
                                                    (2, 8, 64, 512, 511), (3, 4, 128, 300, 257), (1, 2, 64, 4096, 4000)])
def test_matches_eager_chain(dtype, bs, heads, hd, maxlen, pos):
    from quip_amd import ops
    g = torch.Generator().manual_seed(pos + 7 * heads)
    mk = lambda *s: torch.randn(*s, generator=g).to(dtype).to(DEV)
    q, k, v = mk(bs, heads * hd), mk(bs, heads * hd), mk(bs, heads * hd)
    kc, vc = mk(bs, heads, maxlen, hd), mk(bs, heads, maxlen, hd)
    want, kc_w, vc_w = _ref(q, k, v, kc, vc, pos)
    p = torch.tensor([pos], dtype=torch.int64, device=DEV)
    got = ops.decode_attention(q, k, v, kc, vc, p)
    assert torch.equal(kc, kc_w) and torch.equal(vc, vc_w)                 # cache append is exact, nothing else touched
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2                       # output rounding of the dtype
    assert (got.double() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())


def test_graph_replay_advances_with_the_device_position():
    from quip_amd import ops
    bs, heads, hd, maxlen = 1, 4, 64, 64
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g).half().to(DEV)
    xs = [(mk(bs, heads * hd), mk(bs, heads * hd), mk(bs, heads * hd)) for _ in range(5)]
    kc, vc = torch.zeros(bs, heads, maxlen, hd, dtype=torch.float16, device=DEV), torch.zeros(bs, heads, maxlen, hd, dtype=torch.float16, device=DEV)
    kr, vr = kc.clone(), vc.clone()
    q, k, v = (t.clone() for t in xs[0])
    pos = torch.zeros(1, dtype=torch.int64, device=DEV)
    out = torch.zeros(bs, heads * hd, dtype=torch.float16, device=DEV)

    def step():
        out.copy_(ops.decode_attention(q, k, v, kc, vc, pos))
        pos.add_(1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    pos.zero_()
    kc.zero_()
    vc.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    pos.zero_()
    kc.zero_()
    vc.zero_()
    for i, (qi, ki, vi) in enumerate(xs):
        q.copy_(qi), k.copy_(ki), v.copy_(vi)
        graph.replay()
        want, kr, vr = _ref(qi, ki, vi, kr, vr, i)
        assert (out.double() - want).abs().max().item() < 2e-3
    assert int(pos.item()) == 5 and torch.equal(kc, kr)


def test_out_of_range_position_is_a_no_op():
    from quip_amd import ops
    kc = torch.ones(1, 2, 8, 64, dtype=torch.float16, device=DEV)
    vc = kc.clone()
    x = torch.ones(1, 128, dtype=torch.float16, device=DEV)
    ops.decode_attention(x, 2 * x, 2 * x, kc, vc, torch.tensor([8], dtype=torch.int64, device=DEV))
    assert bool((kc == 1).all()) and bool((vc == 1).all())
    with pytest.raises(RuntimeError):
        ops.decode_attention(x.cpu(), x.cpu(), x.cpu(), kc, vc, torch.tensor([0]))
