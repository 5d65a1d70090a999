import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import test_gpu_decode_fused as T
from quip_amd import ops
from quip_amd.quant import packed_u_stage, fused_attention
DEV = "cuda:0"
n, heads, hd, bs, pos, maxlen = 2048, 32, 64, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 0, 64
qkv = [T._layer(n, n, 700 + i)[0] for i in range(3)]
torch.manual_seed(1)
kc = (0.5 * torch.randn(bs, heads, maxlen, hd, device=DEV)).half(); vc = (0.5 * torch.randn(bs, heads, maxlen, hd, device=DEV)).half()
ys = [(0.5 * torch.randn(bs, n, device=DEV)).half() for _ in range(3)]
p_t = torch.tensor([pos], device=DEV)
kc0, vc0 = kc.clone(), vc.clone()
q, k, v = packed_u_stage(qkv, [y.float() for y in ys], torch.float16)
want = ops.decode_attention(q, k, v, kc0, vc0, p_t)
got = fused_attention(qkv, ys, kc, vc, p_t)
torch.cuda.synchronize()
print("k ref  ", k[0, :8].tolist())
print("kc new ", kc[0, 0, pos, :8].tolist())
print("kc ref ", kc0[0, 0, pos, :8].tolist())
print("changed rows new:", (kc != kc0).any(-1).nonzero()[:5].tolist(), int((kc != kc0).any(-1).sum()))
for b in range(bs):
    print("b", b, "k rel", float((kc[b,:,pos].float()-kc0[b,:,pos].float()).norm()/kc0[b,:,pos].float().norm()), "v rel", float((vc[b,:,pos].float()-vc0[b,:,pos].float()).norm()/vc0[b,:,pos].float().norm()), "out rel", float((got[b].float()-want[b].float()).norm()/want[b].float().norm()))
print("out rel", float((got.float() - want.float()).norm() / want.float().norm()))
print("got", got[0, :8].tolist()); print("want", want[0, :8].tolist())
