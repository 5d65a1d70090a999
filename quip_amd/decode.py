"""The decode loop of the reference's `benchmark()` (opt.py:431-482, llama.py:418-471) on packed layers.

The reference times `model(input_ids[:, i], past_key_values=...)` token by token on the Hugging Face model whose Linears
were fake-quantised (its packed kernels were never runnable, quant.py:166-169).  Here the same loop runs on the packed
`QuantLinear` layers themselves:

    with decode.collect_packed() as packed:          # QuantMethod.free() hands every quantised Linear over as a QuantLinear
        opt_sequential(model, dataloader, DEV)       # the reference's own driver, unmodified (opt.py:29-190)
    packed.install(model)                            # make_quant: nn.Linear -> QuantLinear inside the HF model
    engine = decode.DecodeEngine.from_hf(model)      # static KV cache, device-resident position, ONE hipGraph per token
    decode.benchmark(model, input_ids, check=True)   # opt.py:431-482's loop and printout

`from_hf` reads the architecture off an `OPTForCausalLM` / `LlamaForCausalLM` (embeddings, norms, head, the block's Linears --
packed or still dense) and picks the fastest launch sequence the layers allow:

    v3_head  5 (OPT) / 6 (Llama) launches per block + 2 per token: csrc/decode_fused.hip, decode_attn.hip, decode_bigp.hip,
             decode_head.hip -- 2-bit qfn-b layers with Kronecker operators (preproc_proj_extra = 1), fp16, <= 4 rows
    fused    three launches per packed layer group (operator / grouped dequant-GEMM / operator), single-launch attention --
             any packed layer; an operator the small-batch kernels cannot take (the blocked butterfly preproc_proj_extra = 0
             yields, a 688 x 16 factor) runs on the general K3 launches for that side only
    plain    layer.forward() per Linear (also dense nn.Linear models: the fp16 baseline of the same harness)

There is no CPU path: the engine raises on a model that is not on a GPU."""
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import method as _method
from . import ops
from .quant import (QuantLinear, make_quant, packed_forward_fused, fused_stage, fused_ok, fused_attention, fused_attention_ok,
                    fused_u_only, fused_head, fused_head_ok, fused_bigp_tail, bigp_tail_ok, packed_u_stage, stage_operands, attention_operands)

MODES = ("plain", "fused", "v3", "v3_head")
OPERAND_PREFETCH = False   # True: v3 launches carry 8 spare workgroups that pull the step-independent operands of a LATER launch of the block into
                           # every XCD's L2 (csrc/prefetch.h; in a decode step they are cold once per token).  Measured NEUTRAL (profiles/
                           # r06f_decode_ab.jsonl, one box, alternating: OPT-1.3B 1183-1194 tok/s off, 1180-1191 on; Llama-2-7B 610-617 / 618-620;
                           # 4 sequences -2 %): what the warm operands save the consuming launch, the extra workgroups cost the hosting one.  Off.


# ------------------------------------------------------------------------------------------------ packed layers out of a driver run
class collect_packed:
    """context manager: while it is active every `QuantMethod.free()` (the last call the reference's drivers make on a method,
    opt.py:166, llama.py:150) first turns the method's integer state into a packed `QuantLinear` (QuantLinear.from_method: codes,
    grid, rescale vector, operators -- what the reference throws away, method.py:223-225).  Methods without integer codes
    (Nearest, GPTQ's general path) are skipped and stay dense.  `install(model)` swaps the collected layers into the model."""

    def __init__(self):
        self.layers = []          # [(the nn.Linear that was quantised, its QuantLinear)]

    def __enter__(self):
        self._orig = orig = _method.QuantMethod.free
        got = self.layers

        def free(m):
            if getattr(m, 'codes', None) is not None and isinstance(m.layer, nn.Linear):
                got.append((m.layer, QuantLinear.from_method(m, m.layer)))
                m.codes = None
            return orig(m)
        _method.QuantMethod.free = free
        return self

    def __exit__(self, *exc):
        _method.QuantMethod.free = self._orig
        return False

    def named(self, model):
        by_id = {id(lin): ql for lin, ql in self.layers}
        return {name: by_id[id(mod)] for name, mod in model.named_modules() if id(mod) in by_id}

    def install(self, model):
        """replace the quantised nn.Linear modules of `model` by their packed layers; returns {dotted name: QuantLinear}"""
        named = self.named(model)
        make_quant(model, named)
        return named


# ------------------------------------------------------------------------------------------------ OPT
def _is_packed(*layers):
    return all(isinstance(l, QuantLinear) for l in layers)


class OPTBlock(nn.Module):
    """one pre-LN OPT decoder layer over given modules (nn.Linear or QuantLinear; HF's OPTDecoderLayer parts)"""

    def __init__(self, ln1, q_proj, k_proj, v_proj, out_proj, ln2, fc1, fc2, heads):
        super().__init__()
        self.ln1, self.ln2 = ln1, ln2
        self.q_proj, self.k_proj, self.v_proj, self.out_proj, self.fc1, self.fc2 = q_proj, k_proj, v_proj, out_proj, fc1, fc2
        self.h = ln1.weight.numel()
        self.heads, self.hd = heads, self.h // heads
        self.fused = False        # packed layers: q/k/v grouped, LayerNorm / residual / ReLU folded into the operator launches
        self.fused_attn = True    # cache append + q K^T + softmax + p V as one launch (csrc/decode_attn.hip)

    @classmethod
    def random(cls, h, ffn, heads, dtype):
        mk = lambda i, o: nn.Linear(i, o, bias=True, dtype=dtype)
        return cls(nn.LayerNorm(h, dtype=dtype), mk(h, h), mk(h, h), mk(h, h), mk(h, h), nn.LayerNorm(h, dtype=dtype), mk(h, ffn), mk(ffn, h), heads)

    def forward(self, x, kc, vc, pos, mask):
        """x [bs, h]; kc / vc [bs, heads, maxlen, hd]; pos int64 [1] on the device; mask [maxlen] additive (eager attention only)."""
        bs = x.shape[0]
        if self.fused:
            q, k, v = packed_forward_fused([self.q_proj, self.k_proj, self.v_proj], x, ln=self.ln1)
        else:
            hn = self.ln1(x)
            q, k, v = self.q_proj(hn), self.k_proj(hn), self.v_proj(hn)
        if self.fused_attn:
            o = ops.decode_attention(q.contiguous(), k.contiguous(), v.contiguous(), kc, vc, pos)
        else:                                               # the eager chain of HF's attention: nine launches
            q, k, v = (t.view(bs, self.heads, 1, self.hd) for t in (q, k, v))
            kc.index_copy_(2, pos, k)
            vc.index_copy_(2, pos, v)
            att = torch.matmul(q, kc.transpose(2, 3)) * (1.0 / self.hd ** 0.5) + mask    # [bs, heads, 1, maxlen]
            att = torch.softmax(att.float(), -1).to(x.dtype)
            o = torch.matmul(att, vc).reshape(bs, self.h)
        if self.fused:
            x = packed_forward_fused([self.out_proj], o, residual=x)[0]
            hmid = packed_forward_fused([self.fc1], x, ln=self.ln2, relu=True)[0]
            return packed_forward_fused([self.fc2], hmid, residual=x)[0]
        x = x + self.out_proj(o)
        x = x + self.fc2(F.relu(self.fc1(self.ln2(x))))
        return x


class OPTDecoder(nn.Module):
    """token + learned position embedding (offset 2), the blocks, final LayerNorm, head (tied to the embedding unless given)"""
    arch = "opt"

    def __init__(self, tok, posemb, blocks, lnf, heads, lm_head_weight=None):
        super().__init__()
        self.tok, self.posemb, self.lnf = tok, posemb, lnf
        self.blocks = blocks if isinstance(blocks, nn.ModuleList) else nn.ModuleList(blocks)
        self.h, self.layers_n, self.heads = tok.weight.shape[1], len(self.blocks), heads
        self._head_w = lm_head_weight

    v3 = False               # csrc/decode_fused.hip: everything between two GEMMs in the consuming GEMM's prologue -- 5 launches per block
    v3_attn = True           # with v3: the output-side operators of q / k / v in the attention launch (csrc/decode_attn.hip)
    fused_head = False       # with v3: embedding (+ the previous step's argmax) and [U_fc2^T + residual -> final LN -> lm_head -> argmax
                             # partials, pos += 1] as one launch each (csrc/decode_head.hip)

    @property
    def head_weight(self):
        return self.tok.weight if self._head_w is None else self._head_w

    @property
    def kv_heads(self):
        return self.heads

    def packed(self):
        return all(_is_packed(b.q_proj, b.k_proj, b.v_proj, b.out_proj, b.fc1, b.fc2) for b in self.blocks)

    def v3_ok(self, bs):
        b = self.blocks[0]
        return (self.packed() and fused_ok([b.q_proj, b.k_proj, b.v_proj], bs, prev=b.fc2) and fused_ok([b.out_proj], bs, norm=False)
                and fused_ok([b.fc1], bs, prev=b.out_proj) and fused_ok([b.fc2], bs, prev=b.fc1, norm=False, residual=False))

    def head_ok(self, bs):
        return fused_head_ok(self.blocks[-1].fc2, bs, self.lnf)

    def embed(self, ids, pos):
        return self.tok(ids) + F.embedding(pos + 2, self.posemb.weight)

    def head(self, x):
        return F.linear(self.lnf(x), self.head_weight)

    def step_v3(self, x, pos, caches):
        prev, y2, x = self.blocks_v3(x, pos, caches)
        return fused_u_only(prev, y2, residual=x)

    def step_fused_head(self, ids, pos, caches, logits, part_val, part_idx):
        x = torch.empty((ids.numel(), self.h), dtype=torch.float16, device=ids.device)
        ops.decode_embed(self.tok.weight, ids, x, pos_table=self.posemb.weight, pos=pos, pos_offset=2, part_val=part_val, part_idx=part_idx)
        prev, y2, x = self.blocks_v3(x, pos, caches)
        return fused_head(prev, y2, x, self.lnf, self.head_weight, logits, part_val, part_idx, pos_inc=pos)

    def _operand_lists(self):
        """per block: the step-independent operand tensors of each of its five launches (quant.stage_operands), built once"""
        pf = self.__dict__.get('_pf_lists')
        if pf is None:
            pf, prev = [], None
            for blk in self.blocks:
                qkv = [blk.q_proj, blk.k_proj, blk.v_proj]
                pf.append({"qkv": stage_operands(qkv, prev=prev, ln=blk.ln1), "attn": attention_operands(qkv), "o": stage_operands([blk.out_proj]),
                           "fc1": stage_operands([blk.fc1], prev=blk.out_proj, ln=blk.ln2), "fc2": stage_operands([blk.fc2], prev=blk.fc1)})
                prev = blk.fc2
            # (the fc1 -> fc2 launch's per-lane tables are built by its first call: keep the lists only once they are in)
            if all(b.fc2.__dict__.get('_pair_tables') or (b.fc2.V.p, b.fc2.V.q) != (128, 64) for b in self.blocks):
                self.__dict__['_pf_lists'] = pf
        return pf

    def blocks_v3(self, x, pos, caches):
        """per block: [U_fc2^T(prev) + residual -> LN1 -> V_qkv -> GEMM qkv] [U_qkv^T + attention] [V_o -> GEMM o]
        [U_o^T + residual -> LN2 -> V_fc1 -> GEMM fc1] [U_fc1^T + relu -> V_fc2 -> GEMM fc2]; returns (fc2 of the last block, its output
        in the projected basis, the residual stream): the last U_fc2^T + residual belongs to whatever ends the step.
        With OPERAND_PREFETCH the qkv launch (192 workgroups on 256 CUs) also pulls the attention launch's operands into the L2s, the
        attention launch (32 workgroups) those of out_proj and fc1, the out_proj launch (64) those of fc2 and of the next block's qkv."""
        dt = x.dtype
        prev, y2 = None, None
        h16 = torch.float16                                     # y consumed by another fused launch: fp16 (its scatter rounds to fp16 anyway)
        pf = self._operand_lists() if OPERAND_PREFETCH else None
        for bi, (blk, (kc, vc)) in enumerate(zip(self.blocks, caches)):
            qkv = [blk.q_proj, blk.k_proj, blk.v_proj]
            attn_u = self.v3_attn and fused_attention_ok(qkv, kc)
            ydt = h16 if attn_u else torch.float32
            if pf is not None and attn_u:
                ops.decode_prefetch_next(pf[bi]["attn"])
            if prev is None:
                ys, _ = fused_stage(qkv, x=x, ln=blk.ln1, y_dtype=ydt)
            else:
                ys, x = fused_stage(qkv, prev=prev, y_prev=y2, residual=x, ln=blk.ln1, store=True, y_dtype=ydt)
            if attn_u:                                          # U_q^T, U_k^T, U_v^T + bias in the attention launch's prologue
                if pf is not None:
                    ops.decode_prefetch_next((pf[bi]["o"] + pf[bi]["fc1"])[:40])
                o = fused_attention(qkv, ys, kc, vc, pos)
            else:                                               # (the fused launches hand y over in ZT order: K3 wants the natural one)
                q, k, v = packed_u_stage(qkv, [l.from_zt(y) for l, y in zip(qkv, ys)], dt)
                o = ops.decode_attention(q, k, v, kc, vc, pos)
            if pf is not None:
                ops.decode_prefetch_next((pf[bi]["fc2"] + (pf[bi + 1]["qkv"] if bi + 1 < len(pf) else []))[:40])
            yo = fused_stage([blk.out_proj], x=o, y_dtype=h16)[0][0]
            (y1,), x = fused_stage([blk.fc1], prev=blk.out_proj, y_prev=yo, residual=x, ln=blk.ln2, store=True, y_dtype=h16)
            y2 = fused_stage([blk.fc2], prev=blk.fc1, y_prev=y1, relu=True, y_dtype=h16)[0][0]
            prev = blk.fc2
        return prev, y2, x

    def step(self, ids, pos, caches, arange):
        """one token for every batch row: ids int64 [bs], pos int64 [1]; returns logits [bs, vocab]."""
        x = self.embed(ids, pos)
        if self.v3:
            return self.head(self.step_v3(x, pos, caches))
        mask = None
        if not all(b.fused_attn for b in self.blocks):
            mask = torch.where(arange <= pos, 0.0, float("-inf")).to(x.dtype)
        for blk, (kc, vc) in zip(self.blocks, caches):
            x = blk(x, kc, vc, pos, mask)
        return self.head(x)


# ------------------------------------------------------------------------------------------------ Llama
class RMSNorm(nn.Module):
    def __init__(self, h, eps, dtype):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(h, dtype=dtype))
        self.eps = eps

    def forward(self, x):                                   # HF LlamaRMSNorm: fp32 statistics, cast, then the gain
        xf = x.float()
        return self.weight * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)).to(x.dtype)


class LlamaBlock(nn.Module):
    def __init__(self, n1, q_proj, k_proj, v_proj, o_proj, n2, gate_proj, up_proj, down_proj, heads):
        super().__init__()
        self.n1, self.n2 = n1, n2
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = q_proj, k_proj, v_proj, o_proj
        self.gate_proj, self.up_proj, self.down_proj = gate_proj, up_proj, down_proj
        self.h = n1.weight.numel()
        self.heads, self.hd = heads, self.h // heads
        self.fused = False

    @classmethod
    def random(cls, h, ffn, heads, eps, dtype):
        mk = lambda i, o: nn.Linear(i, o, bias=False, dtype=dtype)
        return cls(RMSNorm(h, eps, dtype), mk(h, h), mk(h, h), mk(h, h), mk(h, h), RMSNorm(h, eps, dtype), mk(h, ffn), mk(h, ffn), mk(ffn, h), heads)

    def forward(self, x, kc, vc, pos, cos, sin):
        """x [bs, h]; kc / vc [bs, heads, maxlen, hd]; pos int64 [1] on the device; cos / sin fp32 [maxpos, hd]."""
        if self.fused:
            q, k, v = packed_forward_fused([self.q_proj, self.k_proj, self.v_proj], x, ln=self.n1)
        else:
            hn = self.n1(x)
            q, k, v = self.q_proj(hn), self.k_proj(hn), self.v_proj(hn)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        ops.rope_inplace(q, k, cos, sin, pos, self.heads)
        o = ops.decode_attention(q, k, v, kc, vc, pos)
        if self.fused:
            x = packed_forward_fused([self.o_proj], o, residual=x)[0]
            g, u = packed_forward_fused([self.gate_proj, self.up_proj], x, ln=self.n2)
            return packed_forward_fused([self.down_proj], g, residual=x, gate_up=u)[0]       # silu(g) * u formed inside the V launch
        x = x + self.o_proj(o)
        hn = self.n2(x)
        return x + self.down_proj(F.silu(self.gate_proj(hn)) * self.up_proj(hn))


def rotary_tables(inv_freq, maxpos):
    """HF LlamaRotaryEmbedding's cos / sin in its duplicated-halves layout, fp32 [maxpos, hd]"""
    fr = torch.outer(torch.arange(maxpos, dtype=torch.float32), inv_freq.float().cpu())
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos().contiguous(), emb.sin().contiguous()


class LlamaDecoder(nn.Module):
    arch = "llama"
    NAMES = ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]

    def __init__(self, tok, blocks, norm, lm_head, heads, inv_freq, maxpos):
        super().__init__()
        self.tok, self.norm, self.lm_head = tok, norm, lm_head
        self.blocks = blocks if isinstance(blocks, nn.ModuleList) else nn.ModuleList(blocks)
        self.h, self.layers_n, self.heads = tok.weight.shape[1], len(self.blocks), heads
        cos, sin = rotary_tables(inv_freq, maxpos)
        self.register_buffer("cos", cos, persistent=False)
        self.register_buffer("sin", sin, persistent=False)

    def _apply(self, fn, recurse=True):                     # .to(dev) / .half() must not narrow the rotary tables
        cos, sin = self.cos, self.sin
        super()._apply(fn, recurse)
        self.cos, self.sin = cos.to(self.tok.weight.device), sin.to(self.tok.weight.device)
        return self

    v3 = False               # csrc/decode_fused.hip + decode_attn.hip + decode_bigp.hip: 6 launches per block
    fused_head = False       # with v3: embedding (+ the previous step's argmax) and [U_down^T + residual -> final RMSNorm -> lm_head -> argmax
                             # partials, pos += 1] as one launch each (csrc/decode_head.hip)

    @property
    def head_weight(self):
        return self.lm_head.weight

    def packed(self):
        return all(_is_packed(*(getattr(b, n) for n in self.NAMES)) for b in self.blocks)

    def v3_ok(self, bs):
        b = self.blocks[0]
        qkv = [b.q_proj, b.k_proj, b.v_proj]
        return (self.packed() and fused_ok(qkv, bs, prev=b.down_proj) and fused_ok([b.o_proj], bs, norm=False)
                and fused_ok([b.gate_proj, b.up_proj], bs, prev=b.o_proj) and b.down_proj.U is not None and b.down_proj.U.fused_ok
                and bigp_tail_ok([b.gate_proj, b.up_proj], b.down_proj, bs)
                and fused_attention_ok(qkv, torch.empty((bs, self.heads, 1, b.hd), dtype=torch.float16, device='meta')))

    def head_ok(self, bs):
        return fused_head_ok(self.blocks[-1].down_proj, bs, self.norm)

    def embed(self, ids, pos):
        return self.tok(ids)

    def head(self, x):
        return F.linear(self.norm(x), self.head_weight)

    def step_v3(self, x, pos, caches):
        prev, yd, x = self.blocks_v3(x, pos, caches)
        return fused_u_only(prev, yd.to(torch.float16), residual=x)

    def step_fused_head(self, ids, pos, caches, logits, part_val, part_idx):
        x = torch.empty((ids.numel(), self.h), dtype=torch.float16, device=ids.device)
        ops.decode_embed(self.tok.weight, ids, x, part_val=part_val, part_idx=part_idx)
        prev, yd, x = self.blocks_v3(x, pos, caches)
        return fused_head(prev, yd, x, self.norm, self.head_weight, logits, part_val, part_idx, pos_inc=pos)

    def _operand_lists(self):
        """per block: the step-independent operand tensors of the attention, o_proj and gate / up launches (quant.stage_operands), built once"""
        pf = self.__dict__.get('_pf_lists')
        if pf is None:
            pf = [{"attn": attention_operands([b.q_proj, b.k_proj, b.v_proj]), "o": stage_operands([b.o_proj]),
                   "gu": stage_operands([b.gate_proj, b.up_proj], prev=b.o_proj, ln=b.n2)} for b in self.blocks]
            self.__dict__['_pf_lists'] = pf
        return pf

    def blocks_v3(self, x, pos, caches):
        """per block, six launches: [U_down^T(prev) + residual -> RMSNorm -> V_qkv -> GEMM q,k,v] [U_qkv^T + rotary + attention]
        [V_o -> GEMM o] [U_o^T + residual -> RMSNorm -> V_gate/up -> GEMM gate, up] [U_gate^T, U_up^T (/) s: 688 x 16, decode_bigp.hip]
        [silu * up -> V_down -> GEMM down, K-slices through fp32 atomics]; a 688 x 688 factor is 0.9 MB, not a workgroup's pass: the
        11008-wide operators are cut over the p index (csrc/decode_bigp.hip)"""
        h16 = torch.float16
        prev, yd = None, None
        pf = self._operand_lists() if OPERAND_PREFETCH else None
        for bi, (blk, (kc, vc)) in enumerate(zip(self.blocks, caches)):
            qkv = [blk.q_proj, blk.k_proj, blk.v_proj]
            if pf is not None:                                      # (the q / k / v launch is 192 workgroups: 64 CUs are free for the 8 extra ones)
                ops.decode_prefetch_next(pf[bi]["attn"])
            if prev is None:
                ys, _ = fused_stage(qkv, x=x, ln=blk.n1, y_dtype=h16)
            else:
                ys, x = fused_stage(qkv, prev=prev, y_prev=yd, residual=x, ln=blk.n1, store=True, y_dtype=h16)
            if pf is not None:                                      # the attention launch (32 workgroups): operands of o_proj and of gate / up
                ops.decode_prefetch_next((pf[bi]["o"] + pf[bi]["gu"])[:40])
            o = fused_attention(qkv, ys, kc, vc, pos, self.cos, self.sin)
            yo = fused_stage([blk.o_proj], x=o, y_dtype=h16)[0][0]
            gu = [blk.gate_proj, blk.up_proj]
            ygu, x = fused_stage(gu, prev=blk.o_proj, y_prev=yo, residual=x, ln=blk.n2, store=True, y_dtype=h16)
            yd = fused_bigp_tail(gu, blk.down_proj, ygu)                       # fp32 accumulator, ZT order of down_proj's U
            prev = blk.down_proj
        return prev, yd, x

    def step(self, ids, pos, caches, arange):
        x = self.embed(ids, pos)
        if self.v3:
            return self.head(self.step_v3(x, pos, caches))
        for blk, (kc, vc) in zip(self.blocks, caches):
            x = blk(x, kc, vc, pos, self.cos, self.sin)
        return self.head(x)


# ------------------------------------------------------------------------------------------------ Hugging Face binding
def decoder_from_hf(model, max_len=2048):
    """OPTDecoder / LlamaDecoder over the modules of a Hugging Face `OPTForCausalLM` / `LlamaForCausalLM` (shared, not copied): the
    decoder Linears may be nn.Linear or the QuantLinear layers `make_quant` / `collect_packed.install` put there."""
    cfg = model.config
    mt = getattr(cfg, "model_type", None)
    if mt == "opt":
        d = model.model.decoder
        if not cfg.do_layer_norm_before or getattr(d, "project_in", None) is not None or getattr(d, "project_out", None) is not None:
            raise NotImplementedError("post-LN / projected-embedding OPT variants (opt-350m) are not served by the decode engine")
        if cfg.activation_function != "relu" or d.final_layer_norm is None:
            raise NotImplementedError("OPT decode engine: ReLU feed-forward and a final LayerNorm are assumed")
        heads = cfg.num_attention_heads
        blocks = [OPTBlock(l.self_attn_layer_norm, l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj, l.self_attn.out_proj,
                           l.final_layer_norm, l.fc1, l.fc2, heads) for l in d.layers]
        tied = model.lm_head.weight.data_ptr() == d.embed_tokens.weight.data_ptr()
        return OPTDecoder(d.embed_tokens, d.embed_positions, blocks, d.final_layer_norm, heads, None if tied else model.lm_head.weight)
    if mt == "llama":
        m = model.model
        heads = cfg.num_attention_heads
        if getattr(cfg, "num_key_value_heads", heads) != heads:
            raise NotImplementedError("grouped-query attention is not served by the decode engine (Llama-2-7B/13B are multi-head)")
        if cfg.hidden_act != "silu" or getattr(cfg, "attention_bias", False) or getattr(cfg, "mlp_bias", False):
            raise NotImplementedError("Llama decode engine: SiLU gate, no biases")
        hd = getattr(cfg, "head_dim", None) or cfg.hidden_size // heads
        if hd * heads != cfg.hidden_size:
            raise NotImplementedError("head_dim * heads != hidden_size")
        rot = m.rotary_emb
        if float(getattr(rot, "attention_scaling", 1.0)) != 1.0:
            raise NotImplementedError("scaled rotary embeddings")
        blocks = [LlamaBlock(l.input_layernorm, l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj, l.self_attn.o_proj,
                             l.post_attention_layernorm, l.mlp.gate_proj, l.mlp.up_proj, l.mlp.down_proj, heads) for l in m.layers]
        return LlamaDecoder(m.embed_tokens, blocks, m.norm, model.lm_head, heads, rot.inv_freq, max_len)
    raise NotImplementedError(f"decode engine: model_type {mt!r} (the reference drives OPT and Llama)")


def set_mode(dec, mode):
    """apply one of MODES to a decoder's flags"""
    assert mode in MODES, mode
    for b in dec.blocks:
        b.fused = mode == "fused"
        if hasattr(b, "fused_attn"):
            b.fused_attn = True
    dec.v3 = mode in ("v3", "v3_head")
    dec.fused_head = mode == "v3_head"
    dec.mode = mode


def best_mode(dec, bs, dtype):
    """the fastest launch sequence the decoder's layers allow (module docstring)"""
    if dec.packed():
        if dtype == torch.float16 and dec.v3_ok(bs):
            return "v3_head" if dec.head_ok(bs) else "v3"
        return "fused"
    return "plain"


class DecodeEngine:
    """benchmark()'s per-token loop as ONE hipGraph replay per token: static KV cache [bs, heads, max_len, hd] per block, position and
    current token on the device.  `forward(ids)` is `model(ids, past_key_values=...)` for the next position; `generate` is greedy."""

    def __init__(self, decoder, bs=1, max_len=2048, mode="auto", graph=True):
        self.dec = decoder
        w = decoder.tok.weight
        if not w.is_cuda:
            raise RuntimeError("DecodeEngine needs the model on a GPU (there is no CPU path)")
        if getattr(decoder, "arch", None) == "opt":       # learned positions: the table (offset 2) bounds the sequence
            max_len = min(max_len, decoder.posemb.weight.shape[0] - 2)
        elif getattr(decoder, "arch", None) == "llama":
            max_len = min(max_len, decoder.cos.shape[0])
        self.dev, self.dtype, self.bs, self.max_len = w.device, w.dtype, bs, max_len
        self.mode = best_mode(decoder, bs, self.dtype) if mode == "auto" else mode
        set_mode(decoder, self.mode)
        hd = decoder.h // decoder.heads
        z = lambda: torch.zeros(bs, decoder.heads, max_len, hd, dtype=self.dtype, device=self.dev)
        self.caches = [(z(), z()) for _ in range(decoder.layers_n)]
        self.arange = torch.arange(max_len, device=self.dev)
        self.ids = torch.zeros(bs, dtype=torch.int64, device=self.dev)
        self.pos = torch.zeros(1, dtype=torch.int64, device=self.dev)
        vocab = decoder.head_weight.shape[0]
        self.logits = torch.zeros(bs, vocab, dtype=self.dtype, device=self.dev)
        self._fh = self.mode == "v3_head"
        if self._fh:     # the token comes out of the head launch's partials at the start of the next step; index -1 = "none": the step reads `ids`
            self.part_val = torch.full((bs, ops.HEAD_PARTS), float("-inf"), dtype=torch.float32, device=self.dev)
            self.part_idx = torch.full((bs, ops.HEAD_PARTS), -1, dtype=torch.int32, device=self.dev)
        self._graph = None
        self._want_graph = graph
        self._steps = 0

    @classmethod
    def from_hf(cls, model, packed=None, bs=1, max_len=2048, mode="auto", graph=True, device=None):
        """engine over a Hugging Face causal LM; `packed` ({dotted name: QuantLinear}, e.g. quant.load_packed(...)) is swapped in first.
        Parts of the model still on the CPU (the reference's drivers park blocks there, llama.py:162) are moved to `device`."""
        if packed:
            make_quant(model, packed)
        dec = decoder_from_hf(model, max_len=max_len)
        if device is None:
            qls = [m for m in model.modules() if isinstance(m, QuantLinear)]
            device = qls[0].qweight.device if qls else next(model.parameters()).device
        dec.to(device)
        return cls(dec, bs=bs, max_len=max_len, mode=mode, graph=graph)

    # -- one step ---------------------------------------------------------------------------------
    def _one(self):
        if self._fh:
            self.dec.step_fused_head(self.ids, self.pos, self.caches, self.logits, self.part_val, self.part_idx)
            return
        lg = self.dec.step(self.ids, self.pos, self.caches, self.arange)
        self.logits.copy_(lg)
        ops.argmax_rows(lg, out=self.ids)                # one launch (6 us kernel); torch's generic reduction: 18 us for 50272 logits
        self.pos.add_(1)

    def reset(self):
        self.pos.zero_()
        self._steps = 0
        if self._fh:
            self.part_idx.fill_(-1)

    @torch.no_grad()
    def _prepare(self):
        if self._graph is not None or not self._want_graph:
            return
        keep = self.ids.clone()
        self._one()                                      # warm-up (allocator, table builds); the caches are rewritten from position 0
        self.reset()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._one()
            self.reset()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._one()
        self.reset()
        self.ids.copy_(keep)
        self._graph = g

    @torch.no_grad()
    def step(self):
        """advance one position from the engine's own state (free-running greedy: the next token is the previous step's argmax)"""
        if self._steps >= self.max_len:
            raise RuntimeError("DecodeEngine: KV cache full (max_len %d)" % self.max_len)
        self._prepare()
        if self._graph is not None:
            self._graph.replay()
        else:
            self._one()
        self._steps += 1

    @torch.no_grad()
    def forward(self, ids):
        """feed token ids [bs] (int or tensor) at the next position; returns the logits buffer [bs, vocab] (valid until the next step)"""
        self._prepare()
        if torch.is_tensor(ids):
            self.ids.copy_(ids.reshape(-1))
        else:
            self.ids.fill_(int(ids))
        if self._fh:
            self.part_idx.fill_(-1)                      # teacher-forced: the embed launch reads `ids`, not the previous argmax
        self.step()
        return self.logits

    @torch.no_grad()
    def generate(self, first, n):
        """greedy continuation: returns int64 [n, bs] -- the n tokens following `first` (model.generate(do_sample=False) with a cache)"""
        self.forward(first)
        out = []
        for i in range(n):
            if self._fh:                                 # the embed launch of step i + 1 writes step i's argmax into `ids`
                if i == n - 1:
                    out.append(self.logits.float().argmax(-1))
                else:
                    self.step()
                    out.append(self.ids.clone())
            else:                                        # the step itself leaves its argmax in `ids`
                out.append(self.ids.clone())
                if i < n - 1:
                    self.step()
        return torch.stack(out)

    @torch.no_grad()
    def benchmark(self, input_ids, check=False, verbose=False):
        """opt.py:431-482 / llama.py:418-471: feed `input_ids` [1, n] one token per step, synchronise and time every step; returns
        {'median_s', 'times', 'ppl' (check=True)}"""
        input_ids = input_ids.to(self.dev).reshape(self.bs, -1)
        n = input_ids.shape[1]
        self._prepare()
        self.reset()
        torch.cuda.synchronize()
        times, tot = [], 0.0
        for i in range(n):
            tick = time.time()
            lg = self.forward(input_ids[:, i])
            torch.cuda.synchronize()
            times.append(time.time() - tick)
            if verbose:
                print(i, times[-1])
            if check and i != n - 1:
                tot += F.cross_entropy(lg.float(), input_ids[:, i + 1]).float()
        out = {"median_s": float(np.median(times)), "times": times}
        if check:
            out["ppl"] = float(torch.exp(tot / (n - 1)))
        return out


def benchmark(model, input_ids, check=False, engine=None):
    """drop-in for the reference's `benchmark(model, input_ids, check=False)` (opt.py:431, llama.py:418): same loop, same printout
    ('Median:', 'PPL:'), on the decode engine built over `model` (kept on the model between calls)"""
    eng = engine or getattr(model, "_quip_decode_engine", None)
    n = input_ids.numel()
    if eng is None or eng.max_len < n:
        eng = DecodeEngine.from_hf(model, max_len=max(n, 16))
        model.__dict__["_quip_decode_engine"] = eng
    print('Benchmarking ...')
    res = eng.benchmark(input_ids, check=check, verbose=True)
    print('Median:', res["median_s"])
    if check:
        print('PPL:', res["ppl"])
    return res
