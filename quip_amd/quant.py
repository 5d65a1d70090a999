"""Grid functions, `Quantizer` and the packed low-bit Linear -- the surface of the reference's quant.py
(quant.py:6-246) with the arithmetic in HIP kernels (quip_amd/csrc/gridmap.hip, dqgemm.hip, pack.hip).

Same names / signatures / attribute semantics as the reference so `from quant import *` call sites
(opt.py:10, gptq.py:9) keep working; what is new is `QuantLinear`, the packed 2/4-bit layer the reference
never had a runnable kernel for (its Quant3Linear/Quant4Linear call an absent `quant_cuda`, quant.py:166-169).
"""
import torch
import torch.nn as nn

from . import ops


def _as_maxq(maxq):
    return int(maxq.item()) if torch.is_tensor(maxq) else int(maxq)


def _f64(x):
    """float64 tensors (optq_ldlq_equiv.py builds its FakeLayer in float64 "for numerics", gptq.py:25-27) have no kernel: the
    grid formulas are evaluated with torch ops on the tensor's own device, exactly as the reference writes them."""
    return torch.is_tensor(x) and x.dtype == torch.float64


def quantize_qfna(x, scale, zero, maxq):
    """quant.py:6-8  s*(clamp(round(x/s)+z, 0, maxq) - z), per-row scale/zero."""
    if _f64(x):
        return scale * (torch.clamp(torch.round(x / scale) + zero, 0, maxq) - zero)
    return ops.quantize(x, 'a', scale, zero, _as_maxq(maxq))


def quantize_qfnb(x, scale, maxq):
    """quant.py:10-15  symmetric scalar-scale grid."""
    if _f64(x):
        q = torch.clamp(torch.round(((x / scale + 1) / 2) * maxq), 0, maxq)
        return ((q / maxq) * 2 - 1) * scale
    s = scale if torch.is_tensor(scale) else torch.tensor([float(scale)])
    return ops.quantize(x, 'b', s, None, _as_maxq(maxq))


def quantize_qfnc(x, scale, zero, maxq):
    """quant.py:17-21  clamp-then-round variant (OPTQ == LDLQ equivalence)."""
    if _f64(x):
        return scale * (torch.round(torch.clamp(x / scale + zero, 0, maxq)) - zero)
    return ops.quantize(x, 'c', scale, zero, _as_maxq(maxq))


class Quantizer(nn.Module):
    """quant.py:23-163.  Buffers maxq / scale / zero; configure / find_params / quantize / enabled / ready."""

    def __init__(self, shape=1):
        super().__init__()
        self.register_buffer('maxq', torch.tensor(0))
        self.register_buffer('scale', torch.zeros(shape))
        self.register_buffer('zero', torch.zeros(shape))

    def configure(self, bits, perchannel=False, sym=True, qfn='a', mse=False, norm=2.4, grid=100, maxshrink=.8):
        self.maxq = torch.tensor(2 ** bits - 1)
        self.perchannel, self.sym, self.qfn, self.mse = perchannel, sym, qfn, mse
        self.norm, self.grid, self.maxshrink = norm, grid, maxshrink

    def find_params(self, x, weight=False):
        if self.qfn in ('a', 'c'):
            self.find_params_qfna(x, weight=weight)
        elif self.qfn == 'b':
            self.find_params_qfnb(x)

    def find_params_qfna(self, x, weight=False):
        """Per-row (perchannel) or per-tensor min/max grid, quant.py:57-136.  O(m) glue on the tensor's
        device; the fp32 promotion of the reference (quant.py:76-78) is kept."""
        if self.mse:
            raise NotImplementedError("mse grid search is unreachable in the reference (quant.py:104 calls an "
                                      "undefined `quantize`); every caller passes mse=False")
        dev = x.device
        self.maxq = self.maxq.to(dev)
        shape = x.shape
        if self.perchannel:
            if weight:
                flat = x.flatten(1)
            elif len(shape) == 4:
                flat = x.permute(1, 0, 2, 3).flatten(1)
            elif len(shape) == 3:
                flat = x.reshape(-1, shape[-1]).t()
            else:
                flat = x.t()
        else:
            flat = x.flatten().unsqueeze(0)
        zeros = torch.zeros(flat.shape[0], device=dev)
        lo = torch.minimum(flat.min(1)[0], zeros)          # fp32 from here on
        hi = torch.maximum(flat.max(1)[0], zeros)
        if self.sym:
            hi = torch.maximum(lo.abs(), hi)
            lo = torch.where(lo < 0, -hi, lo)
        degenerate = (lo == 0) & (hi == 0)
        lo = torch.where(degenerate, torch.full_like(lo, -1), lo)
        hi = torch.where(degenerate, torch.full_like(hi, 1), hi)
        self.scale = (hi - lo) / self.maxq
        self.zero = torch.full_like(self.scale, (self.maxq + 1) / 2) if self.sym else torch.round(-lo / self.scale)
        if not self.perchannel:
            reps = shape[0] if weight else (shape[1] if len(shape) != 3 else shape[2])
            self.scale, self.zero = self.scale.repeat(reps), self.zero.repeat(reps)
        if weight:
            view = [-1] + [1] * (len(shape) - 1)
        elif len(shape) == 4:
            view = (1, -1, 1, 1)
        elif len(shape) == 3:
            view = (1, 1, -1)
        else:
            view = (1, -1)
        self.scale, self.zero = self.scale.reshape(view), self.zero.reshape(view)

    def find_params_qfnb(self, x):
        self.maxq = self.maxq.to(x.device)
        self.scale = None       # recomputed from the tensor handed to quantize(), quant.py:138-142
        self.zero = None

    def quantize(self, x):
        if self.qfn == 'a':
            assert self.ready()
            return quantize_qfna(x, self.scale, self.zero, self.maxq)
        if self.qfn == 'b':
            assert torch.all(self.maxq != 0)
            if _f64(x):
                self.scale = 2.4 * x.square().mean().sqrt() + 1e-16
                return quantize_qfnb(x, self.scale, self.maxq)
            s = ops.qfnb_scale(x)                               # 2.4*rms(x)+1e-16 in x's dtype, quant.py:150
            self.scale = s.to(x.dtype).reshape(())
            return ops.quantize(x, 'b', s, None, _as_maxq(self.maxq))
        if self.qfn == 'c':
            assert self.ready()
            return quantize_qfnc(x, self.scale, self.zero, self.maxq)
        return NotImplementedError()                            # sic: the reference returns it (quant.py:157)

    def enabled(self):
        return self.maxq > 0

    def ready(self):
        return self.scale is not None and bool(torch.all(self.scale != 0))


# Llama's down_proj is a split-K launch (csrc/decode_bigp.hip): by default its K-slices meet in y through fp32 atomics, and two runs of the
# same step can differ in the last bits (then, after an fp16 rounding downstream, in a greedy token near a tie).  True: the slices meet in a
# fixed order through a scratch buffer -- bit-identical runs, one more L2 round trip per MLP tail.  Set it before the engine captures.
DETERMINISTIC_SPLITK = False

# Decode steps of several sequences: up to ops.FUSED_MAX_ROWS rows can ride in the single fused launch per layer group (its prologue
# walks the rows one after the other, 1.3 - 2.7 us per row and launch), or the step takes [prologue-only launch, one workgroup per row] +
# [dequant-GEMM] instead (fused_stage, fused_bigp_tail: one more launch per group, 2.9 us at OPT-1.3B's sizes, 4.5 at Llama-2-7B's).
# TWO_LAUNCH_ROWS = the row count from which the second form is taken, 2 .. ops.FUSED_MAX_ROWS + 1; None = by the layer's size, from
# profiles/r05s_two_launch_*.jsonl: OPT-1.3B (hidden 2048) 3 rows 1.32 -> 1.10 ms per step, 4 rows 1.55 -> 1.11, 2 rows 1.00 -> 1.09;
# Llama-2-7B (hidden 4096) 2 / 3 / 4 rows 1.87 / 2.17 / 2.45 fused against 2.33 / 2.42 / 2.53.  Set it before the engine captures.
TWO_LAUNCH_ROWS = None


def two_launch_from(ql):
    """the row count from which a decode step through packed layer `ql` takes the two-launch form (see TWO_LAUNCH_ROWS)"""
    if TWO_LAUNCH_ROWS is not None:
        return max(2, min(int(TWO_LAUNCH_ROWS), ops.FUSED_MAX_ROWS + 1))
    return 3 if min(ql.infeatures, ql.outfeatures) <= 2048 else ops.FUSED_MAX_ROWS + 1


class QuantLinear(nn.Module):
    """Packed 2/4-bit Linear (the runnable successor of Quant3Linear / Quant4Linear, quant.py:173-233,
    zeroShot/models/quant.py:183-212).  Holds codes in the STREAM layout plus, when the layer was quantised
    with incoherence processing, the structured operators so that

        y = U^T ( What2 ( V (x / s) ) ) + bias

    where What2 = dequant(codes) lives in the projected basis.  forward() accepts any batch shape."""

    def __init__(self, infeatures, outfeatures, bits=2, qfn='b'):
        super().__init__()
        assert bits in (2, 3, 4)                  # 3-bit codes (--wbits 3) are stored in the 4-bit container
        self.infeatures, self.outfeatures, self.bits, self.qfn = infeatures, outfeatures, bits, qfn
        self.register_buffer('qweight', torch.zeros(infeatures * outfeatures * ops.container_bits(bits) // 32, dtype=torch.int32))
        self.register_buffer('scales', torch.zeros(1 if qfn == 'b' else outfeatures))
        self.register_buffer('zeros', torch.zeros(outfeatures) if qfn != 'b' else None)
        self.register_buffer('bias', None)
        self.register_buffer('inv_scaleWH', None)
        self.U = None     # ops.OrthoOp over out features
        self.V = None     # ops.OrthoOp over in features
        self._drop_derived()

    _F32_BUFFERS = ('scales', 'zeros', 'bias', 'inv_scaleWH')

    def _apply(self, fn, recurse=True):
        """model.half() / .to(dtype) must not narrow the grid parameters: the kernels read them as float32 (and fp16 could
        not hold them).  Device moves are followed, dtype changes of these buffers are undone without a round trip."""
        keep = {n: getattr(self, n, None) for n in self._F32_BUFFERS}
        dev0 = self.qweight.device
        only_decode = self.qweight.numel() == 0 and self.infeatures * self.outfeatures > 0
        qd = self.__dict__.get('_qweight_d')
        super()._apply(fn, recurse)
        if self.qweight.device != dev0:       # everything the decode launches derived from the packed state holds device pointers
            self._drop_derived()
            # the operators are ops.OrthoOp objects with a fixed .device (not buffers): rebuild them from their state on the new device, or the
            # fused launches would get factor pointers on the old GPU beside codes on the new one
            for side in ('U', 'V'):
                op = getattr(self, side, None)
                if op is not None and torch.device(op.device) != self.qweight.device:
                    setattr(self, side, ops.OrthoOp(op.state(), self.qweight.device))
            if only_decode and qd is not None:
                # after decode_only() the decode-order words are the ONLY copy of the weights: they travel with the module (the tables
                # derived from them are still rebuilt on the new device)
                self.__dict__['_qweight_d'] = qd.to(self.qweight.device)
        for n, old in keep.items():
            new = getattr(self, n, None)
            if old is not None and new is not None and new.dtype != torch.float32:
                self._buffers[n] = old.to(device=new.device, dtype=torch.float32)
        return self

    @torch.no_grad()
    def pack(self, codes, scale, zero=None, bias=None, scaleWH=None, U=None, V=None):
        """codes uint8 [out,in] on the GPU; scale float[1] (qfn b) or [out] (qfn a); U/V reference-style
        (B, p_in, p_out) tuples or ops.OrthoOp."""
        dev = codes.device
        # what the decode launches derive from the packed state (decode_qweight, bias16, the layer-pair tables) is rebuilt on next use;
        # tables other layers keep about THIS one are keyed by its generation
        self._drop_derived()
        self.qweight = ops.pack(codes, self.bits, ops.LAYOUT_STREAM)
        self.scales = scale.to(dev, torch.float32).reshape(-1).clone()
        self.zeros = None if zero is None else zero.to(dev, torch.float32).reshape(-1).clone()
        self.bias = None if bias is None else bias.detach().to(dev, torch.float32).clone()
        self.inv_scaleWH = None if scaleWH is None else (1.0 / scaleWH.to(dev, torch.float32))
        self.U = U if (U is None or isinstance(U, ops.OrthoOp)) else ops.OrthoOp(U, dev)
        self.V = V if (V is None or isinstance(V, ops.OrthoOp)) else ops.OrthoOp(V, dev)
        return self

    _GEN = [0]

    def _drop_derived(self):
        """forget what the decode launches derived from the packed state (decode_qweight, bias16, the layer-pair / MLP-tail tables) and take a
        new generation number: tables OTHER layers keep about this one are keyed by it (process-wide counter: a recycled id() cannot
        collide with a dead layer's entry), and a device move invalidates them like a re-pack does"""
        for k in ('_qweight_d', '_bias16', '_pair_tables', '_bigp_tail', '_splitk_ws'):
            self.__dict__.pop(k, None)
        QuantLinear._GEN[0] += 1
        self.__dict__['_pack_gen'] = QuantLinear._GEN[0]

    def packed_bytes(self):
        """bytes of codes this layer holds on its device: the STREAM words, plus the decode-order copy once a fused decode launch built it"""
        qd = self.__dict__.get('_qweight_d')
        return self.qweight.numel() * 4 + (0 if qd is None else qd.numel() * 4)

    @torch.no_grad()
    def decode_only(self):
        """keep ONLY the decode-order codes (the fused decode launches read nothing else): frees the natural-order STREAM words, after which
        forward() / the K3-side stages of this layer raise.  For serving from the fused engine at 2 bits per weight instead of 4."""
        self.decode_qweight()
        self.qweight = torch.empty(0, dtype=torch.int32, device=self.qweight.device)
        return self

    @classmethod
    @torch.no_grad()
    def from_method(cls, method, layer):
        """Packed layer from a Balance (or any QuantMethod that kept integer codes) AFTER fasterquant() and BEFORE
        free(): codes + grid parameters in the projected basis, plus the rescale vector and the structured operators
        of QuantMethod.preproc -- the state the reference throws away (vector_balance.py:528-530, method.py:223-225)."""
        codes = method.codes
        bits = int(torch.log2(method.quantizer.maxq.float() + 1).round().item())
        qfn = method.quantizer.qfn if method.quantizer.qfn in ('a', 'b') else 'a'
        ql = cls(codes.shape[1], codes.shape[0], bits=bits, qfn=qfn)
        U = getattr(method, '_U', None) if getattr(method, 'preproc_proj', False) else None
        V = getattr(method, '_V', None) if getattr(method, 'preproc_proj', False) else None
        sWH = method.scaleWH if getattr(method, 'preproc_rescale', False) else None
        bias = None if getattr(layer, 'bias', None) is None else layer.bias.data
        return ql.pack(codes, method.qscale, method.qzero, bias=bias, scaleWH=sWH, U=U, V=V)

    @torch.no_grad()
    def decode_qweight(self):
        """the codes as the fused decode launches read them (include/quip_amd.h "permutations folded into the packing"): rows in ZT order
        of U, columns in image order of V -- built once (unpack, index, pack on the device) and kept beside `qweight`"""
        qd = self.__dict__.get('_qweight_d')
        if qd is None or qd.device != self.qweight.device:
            if self.qweight.numel() == 0:
                raise RuntimeError("QuantLinear.decode_only() dropped the natural-order codes: re-pack the layer to rebuild its decode copy")
            m, d = self.outfeatures, self.infeatures
            codes = ops.unpack(self.qweight, self.bits, ops.LAYOUT_STREAM, m, d)
            if self.U is not None and self.U.fold_ok:         # (an operator no decode launch can run keeps the natural order: its side
                perm = torch.empty(m, dtype=torch.int64, device=codes.device)   #  is served by the K3 kernels)
                perm[self.U.zt_rows()] = torch.arange(m, device=codes.device)          # new row r holds old row perm[r]
                codes = codes[perm]
            if self.V is not None and self.V.fold_ok:
                perm = torch.empty(d, dtype=torch.int64, device=codes.device)
                perm[self.V.image_cols()] = torch.arange(d, device=codes.device)
                codes = codes[:, perm]
            qd = ops.pack(codes.contiguous(), self.bits, ops.LAYOUT_STREAM)
            self.__dict__['_qweight_d'] = qd
        return qd

    def to_zt(self, y):
        """a vector in this layer's natural output order -> ZT order (what its decode launch produces)"""
        if self.U is None or not self.U.fold_ok:
            return y
        out = torch.empty_like(y)
        out[..., self.U.zt_rows()] = y
        return out

    def from_zt(self, y):
        return y if (self.U is None or not self.U.fold_ok) else y[..., self.U.zt_rows()]

    def packed_state(self):
        """everything needed to rebuild the layer, as CPU tensors / plain Python (the packed checkpoint record):
        STREAM-layout codes, grid parameters, bias, 1/scaleWH and the generator tuples of U and V."""
        if self.qweight.numel() == 0 and self.infeatures * self.outfeatures:
            raise RuntimeError("QuantLinear.decode_only() dropped the natural-order codes this record is made of: save the packed checkpoint first")
        cpu = lambda t: None if t is None else t.detach().cpu()
        return {"infeatures": self.infeatures, "outfeatures": self.outfeatures, "bits": self.bits, "qfn": self.qfn,
                "qweight": cpu(self.qweight), "scales": cpu(self.scales), "zeros": cpu(self.zeros), "bias": cpu(self.bias),
                "inv_scaleWH": cpu(self.inv_scaleWH),
                "U": None if self.U is None else self.U.state(), "V": None if self.V is None else self.V.state()}

    @classmethod
    def from_packed_state(cls, st, device):
        ql = cls(st["infeatures"], st["outfeatures"], bits=st["bits"], qfn=st["qfn"])
        dev = lambda t: None if t is None else t.to(device)
        ql.qweight, ql.scales, ql.zeros = dev(st["qweight"]), dev(st["scales"]), dev(st["zeros"])
        ql.bias, ql.inv_scaleWH = dev(st["bias"]), dev(st["inv_scaleWH"])
        ql.U = None if st["U"] is None else ops.OrthoOp(st["U"], device)
        ql.V = None if st["V"] is None else ops.OrthoOp(st["V"], device)
        ql._drop_derived()
        return ql

    def act_dtype(self, x):
        """dtype of the activations fed to K2: the reference operator widens x to fp32 (quant.py:226-229), so an fp16 model
        must not lose mantissa bits to a bf16 cast -- fp16 x runs on the fp16 MFMA pipe where a kernel for it exists
        (in_features % 256 == 0 for any batch; up to 16 rows also any in_features <= 4096); bf16 / fp32 x run in bf16
        (fp32: range over mantissa)."""
        rows = x.numel() // max(x.shape[-1], 1)
        if x.dtype == torch.float16 and (self.infeatures % 256 == 0 or (rows <= 16 and self.infeatures <= 4096)):
            return torch.float16
        return torch.bfloat16

    def forward(self, x):
        if self.qweight.numel() == 0 and self.infeatures * self.outfeatures:
            raise RuntimeError("QuantLinear.decode_only() kept the decode-order codes only: this layer runs through the fused decode launches")
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        adt = self.act_dtype(x)
        # a decode step: blocked operators may take csrc/ortho_blk.hip (fp16 factors, ~3e-4 per stage) -- never for a float32 caller, who
        # gets the fp32-factor K3 launches whatever the row count
        few = x2.shape[0] <= ops.BLK_MAX_ROWS and x.dtype != torch.float32
        if self.V is not None:
            xt = self.V.apply_rows(x2.contiguous(), colscale=self.inv_scaleWH, out_dtype=adt, fast16=few)
        else:
            xt = x2.to(adt)
            if self.inv_scaleWH is not None:
                xt = (x2.float() * self.inv_scaleWH).to(adt)
        if self.U is None:
            y = ops.dequant_gemm(xt, self.qweight, self.bits, self.qfn, self.scales, self.zeros, self.bias,
                                 out_dtype=torch.float32, m=self.outfeatures)
        else:
            y = ops.dequant_gemm(xt, self.qweight, self.bits, self.qfn, self.scales, self.zeros, None,
                                 out_dtype=torch.float32, m=self.outfeatures)
            y = self.U.apply_rows(y, transpose=True, out_dtype=x.dtype, bias=self.bias, fast16=few)   # fp32 in, caller's dtype out
        return y.to(x.dtype).reshape(*shape[:-1], self.outfeatures)


class Quant3Linear(QuantLinear):
    """The reference's packed 3-bit layer (quant.py:173-233) by name and call protocol:
    `Quant3Linear(infeatures, outfeatures)`, `.pack(linear, scales, zeros)` from a fake-quantised nn.Linear and its
    per-row grid (qfn a), `forward(x)` = bias + sum (scales q - zeros scales) x.  Differences: the codes live on the GPU
    in the 4-bit STREAM container that K2 reads (the reference's 32-codes-in-3-words array had no runnable kernel,
    quant.py:166-169), and forward() accepts any batch, not only a single token (quant.py:233)."""

    def __init__(self, infeatures, outfeatures):
        super().__init__(infeatures, outfeatures, bits=3, qfn='a')

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """A checkpoint written by the REFERENCE's Quant3Linear (opt.py:303-315 opt_pack3, quant.py:176-197) loads as it is:
        qweight int32 [in/32*3, out] in the 32-codes-in-3-words packing, scales [out,1], zeros [out,1] = zero * scale, bias.
        The codes are repacked on the device (quipamd_repack_canonical_to_stream), no host pass."""
        qw = state_dict.get(prefix + 'qweight')
        if qw is not None and qw.dim() == 2:
            sc, zr, b = state_dict[prefix + 'scales'], state_dict[prefix + 'zeros'], state_dict.get(prefix + 'bias')
            if tuple(qw.shape) != (self.infeatures * 3 // 32, self.outfeatures) or sc.numel() != self.outfeatures \
                    or zr.numel() != self.outfeatures or (b is not None and b.numel() != self.outfeatures):
                raise RuntimeError(f"{prefix}: not a reference Quant3Linear record for a {self.infeatures} -> {self.outfeatures} layer")
            if self.qweight.device.type != 'cuda':
                # the reference's flow (opt.py load_quant3): make_quant3 on the CPU, load_state_dict, THEN .to(dev).  The device
                # repack has no CPU twin, so the canonical words wait on the module and are repacked by the first _apply that
                # lands on a GPU; the state dict handed on to nn.Module holds nothing for this layer's buffers.
                self._reference_pending = tuple(None if t is None else t.detach().clone() for t in (qw, sc, zr, b))
                for k in ('qweight', 'scales', 'zeros', 'bias'):
                    state_dict.pop(prefix + k, None)
                missing = args[2] if len(args) > 2 else kwargs.get('missing_keys')
                super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
                if missing is not None:                      # they are not missing, they are pending
                    for k in ('qweight', 'scales', 'zeros', 'bias'):
                        if prefix + k in missing:
                            missing.remove(prefix + k)
                return
            conv = reference_packed_buffers(qw, sc, zr, 3, self.infeatures, self.outfeatures, self.qweight.device)
            state_dict[prefix + 'qweight'], state_dict[prefix + 'scales'], state_dict[prefix + 'zeros'] = conv
            if b is not None:                                # only after every shape check above has passed
                self.bias = torch.zeros(self.outfeatures, device=self.qweight.device)
                state_dict[prefix + 'bias'] = b.reshape(-1).float()
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        pend = getattr(self, '_reference_pending', None)
        if pend is not None and self.qweight.device.type == 'cuda':
            qw, sc, zr, b = pend
            dev = self.qweight.device
            self.qweight, self.scales, self.zeros = reference_packed_buffers(qw, sc, zr, 3, self.infeatures, self.outfeatures, dev)
            self.bias = None if b is None else b.to(dev, torch.float32).reshape(-1).clone()
            self._reference_pending = None
        return self

    def forward(self, x):
        if getattr(self, '_reference_pending', None) is not None:
            raise RuntimeError("Quant3Linear holds a reference-format checkpoint that is repacked on the GPU: move the module "
                               "to the device first (there is no CPU fallback)")
        return super().forward(x)

    @torch.no_grad()
    def pack(self, linear, scales, zeros, **kw):
        if not isinstance(linear, nn.Module):            # QuantLinear.pack(codes, scale, zero, ...) protocol
            return super().pack(linear, scales, zeros, **kw)
        dev = linear.weight.device if linear.weight.is_cuda else torch.device(_DEV)
        sc = scales.to(dev, torch.float32).reshape(-1, 1)
        zr = zeros.to(dev, torch.float32).reshape(-1, 1)
        W = linear.weight.data.to(dev, torch.float32)
        codes = torch.round((W + zr * sc) / sc).clamp_(0, 7).to(torch.uint8)          # quant.py:186-191
        bias = None if linear.bias is None else linear.bias.data
        return super().pack(codes, sc, zr, bias=bias)


_DEV = 'cuda:0'


def reference_packed_buffers(qweight, scales, zeros, bits, infeatures, outfeatures, device):
    """(qweight, scales, zeros) of a layer packed by the reference -- Quant3Linear.pack (quant.py:185-220: [in/32*3, out])
    or Quant4Linear (zeroShot/models/quant.py:183-199: [in/8, out]), zeros = zero * scale -- as the buffers QuantLinear keeps:
    STREAM-layout codes (repacked on the device), per-row scale, integer zero."""
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError("a reference-format checkpoint is repacked on the GPU: move the module to the device first "
                           "(there is no CPU fallback)")
    assert tuple(qweight.shape) == (infeatures * bits // 32, outfeatures), "not the reference's [in*bits/32, out] packing"
    qs = ops.repack_canonical_to_stream(qweight.to(device=device, dtype=torch.int32), bits, outfeatures, infeatures)
    sc = scales.to(device, torch.float32).reshape(-1)
    zs = zeros.to(device, torch.float32).reshape(-1)
    zi = torch.where(sc != 0, zs / sc, torch.zeros_like(zs)).round()          # zeros = zero * scale, zero an integer
    return qs, sc.clone(), zi


def from_reference_packed(qweight, scales, zeros, bias, bits, device=_DEV):
    """QuantLinear (qfn a) from the buffers of a reference-packed layer (3 or 4 bit, see reference_packed_buffers)."""
    outfeatures = qweight.shape[1]
    infeatures = qweight.shape[0] * 32 // bits
    ql = (Quant3Linear(infeatures, outfeatures) if bits == 3 else QuantLinear(infeatures, outfeatures, bits=bits, qfn='a')).to(device)
    ql.qweight, ql.scales, ql.zeros = reference_packed_buffers(qweight, scales, zeros, bits, infeatures, outfeatures, device)
    ql.bias = None if bias is None else bias.detach().to(device, torch.float32).reshape(-1).clone()
    return ql


def make_quant3(module, names, name=''):
    """replace the named nn.Linear children by (empty) Quant3Linear layers, to be filled by .pack() or a state dict
    (quant.py:236-246)."""
    if isinstance(module, Quant3Linear):
        return
    for attr, tmp in list(module.named_children()):       # the reference walks dir(module); children cover containers too
        name1 = name + '.' + attr if name != '' else attr
        if name1 in names:
            setattr(module, attr, Quant3Linear(tmp.in_features, tmp.out_features))
    for name1, child in module.named_children():
        make_quant3(child, names, name + '.' + name1 if name != '' else name1)


def _ln_params(ln):
    """(gamma, beta, eps) of an nn.LayerNorm; an RMSNorm module (torch.nn.RMSNorm, HF LlamaRMSNorm: weight + eps /
    variance_epsilon, no bias) gives beta = None, which the kernels read as RMSNorm."""
    if ln is None:
        return None
    eps = getattr(ln, 'eps', None)
    if eps is None:
        eps = getattr(ln, 'variance_epsilon', None)
    if eps is None:
        eps = torch.finfo(ln.weight.dtype).eps              # torch.nn.RMSNorm(eps=None)
    beta = getattr(ln, 'bias', None)
    if beta is None and isinstance(ln, nn.LayerNorm):       # LayerNorm(bias=False) still subtracts the mean: a zero beta, not RMSNorm
        beta = getattr(ln, '_quip_zero_beta', None)
        if beta is None or beta.device != ln.weight.device or beta.dtype != ln.weight.dtype:
            beta = torch.zeros_like(ln.weight)
            ln._quip_zero_beta = beta
    return (ln.weight, beta, eps)


def _fusable(ql, rows):
    return (ql.U is not None and ql.V is not None and ql.U.small_ok and ql.V.small_ok and rows <= ops.OrthoOp.SMALL_ROWS
            and ql.qfn == 'b')


def _norm(ln, x):
    if ln is None:
        return x
    if isinstance(ln, nn.Module):
        return ln(x)
    g, b, eps = ln                                          # a (gamma, beta, eps) tuple: LayerNorm, or RMSNorm when beta is None
    xf = x.float()
    if b is None:
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * g.float()).to(x.dtype)
    return torch.nn.functional.layer_norm(xf, (x.shape[-1],), g.float(), b.float(), eps).to(x.dtype)


def packed_forward_fused(qls, x, ln=None, residual=None, relu=False, gate_up=None):
    """Forward of 1..4 packed layers that share the input x [rows, d] (q / k / v of a block, gate / up of a Llama MLP, or a
    single layer) in THREE launches total, with the neighbouring elementwise work of the decoder block folded in (a decode
    step is launch-latency bound):
        launch 1   xt_i = V_i ( Norm(x) (/) s_i )                          ln: nn.LayerNorm / an RMSNorm module / None
        launch 2   y_i  = What_i xt_i                                      grouped fused dequant-GEMM
        launch 3   out_i = [relu]( U_i^T y_i + bias_i + residual )
    Returns the list of outputs in x's dtype.  Each side is decided on its own: an operator that does not fit the small-batch
    kernels (Llama's 11008 = 688 x 16) takes the general K3 launches for THAT side only, with the norm / residual / relu
    it would have absorbed done by torch ops; layers without operators at all take the plain forward.
    gate_up: x is the GATE of a Llama MLP and gate_up the `up` projection [rows, d]; the layer's input is silu(x) * gate_up
    (llama's down_proj(act_fn(gate) * up)), formed on load inside launch 1 when the V-side operator is p x 16, else by torch."""
    rows = x.shape[0]
    assert x.dim() == 2
    if gate_up is not None:
        one = len(qls) == 1 and qls[0].V is not None and qls[0].U is not None and qls[0].qfn == 'b'
        in_kernel = (one and ln is None and qls[0].V.bigp_ok and not qls[0].V.small_ok and rows <= ops.BIGP_ROWS and x.dtype == torch.float16
                     and x.stride(0) == gate_up.stride(0) and gate_up.dtype == torch.float16)
        in_blk = (one and ln is None and qls[0].V.blk_ok and rows <= ops.BLK_MAX_ROWS and x.dtype in ops._DT and gate_up.dtype == x.dtype
                  and x.is_contiguous() and gate_up.is_contiguous())
        if not (in_kernel or in_blk):
            x, gate_up = torch.nn.functional.silu(x) * gate_up, None
    same = len({(q.infeatures, q.outfeatures, q.bits) for q in qls}) == 1
    if not same or any(q.U is None or q.V is None or q.qfn != 'b' for q in qls) or rows > ops.OrthoOp.SMALL_ROWS:
        h = _norm(ln, x)
        outs = [q(h) for q in qls]
        if residual is not None:
            outs = [o + residual for o in outs]
        return [torch.relu(o) for o in outs] if relu else outs
    dev, m, d = x.device, qls[0].outfeatures, qls[0].infeatures
    x = x.contiguous()
    # launch 1
    fast = lambda o, norm: o.small_ok or (o.bigp_ok and norm is None and rows <= ops.BIGP_ROWS)

    def launchable(entries):
        """a p x 16-only operator exists for the compiled operand sets of csrc/ortho_bigp.hip alone (ops._tile_form): any other
        combination of dtypes / operands (a bf16 model, an fp32 residual ...) takes the general K3 launches for that side"""
        return all(o.small_ok or ops._tile_form(d_) is not None for o, d_, _ in entries)
    v_entries = None
    if all(fast(q.V, ln) for q in qls):
        xts = [torch.empty((rows, d), dtype=torch.bfloat16, device=dev) for _ in qls]
        lnp = _ln_params(ln)
        # a layer packed with preproc_proj but no rescale has no column scale: the kernels read ones
        cs = [q.inv_scaleWH if q.inv_scaleWH is not None else q.V.one_scale() for q in qls]
        if gate_up is not None:                                 # silu(x) * up on load (csrc/ortho_bigp.hip)
            gate_up = gate_up.contiguous()
            v_entries = [(q.V, q.V.small_op(x, xt, colscale=c, residual=gate_up, relu=True), False) for q, xt, c in zip(qls, xts, cs)]
        else:
            v_entries = [(q.V, q.V.small_op(x, xt, colscale=c, ln=lnp), False) for q, xt, c in zip(qls, xts, cs)]
        if not launchable(v_entries):
            v_entries = None
    blk = lambda o, dt: o.blk_ok and rows <= ops.BLK_MAX_ROWS and dt in ops._DT
    if v_entries is not None:
        ops.ortho_apply_ops(v_entries, rows)
    elif all(blk(q.V, x.dtype) for q in qls) and (ln is None or _ln_params(ln)[0].dtype == torch.float16):
        # blocked butterfly (what --incoh_processing really yields): two launches per operator with the norm / silu * up / column scale of
        # the block fused into the first (csrc/ortho_blk.hip)
        lnp, gu = _ln_params(ln), (None if gate_up is None else gate_up.contiguous())
        # x~ in the model's own 16-bit type where K2 has a kernel for it (QuantLinear.act_dtype): an fp16 model keeps its mantissa bits, and
        # the grouped one-pass kernels (csrc/dqgemm_v2.hip: fp16 only) take q / k / v -- in bf16 they fell to the round-1 tile kernel
        xt_dtype = qls[0].act_dtype(x)
        xts = []
        for i in range(0, len(qls), 3):                         # up to three operators per launch pair (q / k / v, gate / up)
            xts += ops.ortho_blocked_multi([(q.V, x, dict(colscale=q.inv_scaleWH, ln=lnp, gate_up=gu)) for q in qls[i:i + 3]], xt_dtype)
        gate_up = None
    else:
        if gate_up is not None:
            x, gate_up = torch.nn.functional.silu(x) * gate_up, None
        h = _norm(ln, x)
        xts = [q.V.apply_rows(h, colscale=q.inv_scaleWH, out_dtype=torch.bfloat16) for q in qls]
    # launch 2
    ys = [torch.empty((rows, m), dtype=torch.float32, device=dev) for _ in qls]
    ops.dequant_gemm_grouped(xts, [q.qweight for q in qls], qls[0].bits, 'b', [q.scales for q in qls], None, ys, m)
    # launch 3
    res = None if residual is None else residual.contiguous()
    if all(fast(q.U, None) for q in qls):
        outs = [torch.empty((rows, m), dtype=x.dtype, device=dev) for _ in qls]
        u_entries = [(q.U, q.U.small_op(y, o, transpose=True, bias=q.bias if q.bias is not None else q.U.zero_bias(), residual=res, relu=relu), True)
                     for q, y, o in zip(qls, ys, outs)]
        if launchable(u_entries):
            ops.ortho_apply_ops(u_entries, rows)
            return outs
    if all(blk(q.U, torch.float32) for q in qls):              # bias + residual + ReLU in the second launch of the blocked operator
        outs = []
        for i in range(0, len(qls), 3):
            outs += ops.ortho_blocked_multi([(q.U, y, dict(transpose=True, bias=q.bias, residual=res, relu=relu)) for q, y in zip(qls[i:i + 3], ys[i:i + 3])], x.dtype)
        return outs
    outs = [q.U.apply_rows(y, transpose=True, out_dtype=x.dtype, bias=q.bias) for q, y in zip(qls, ys)]
    if res is not None:
        outs = [o + res for o in outs]
    return [torch.relu(o) for o in outs] if relu else outs


def _chainable(qa, qbs, rows):
    return (_fusable(qa, rows) and all(_fusable(q, rows) for q in qbs) and qa.U.split_ok and qa.U.use_split
            and all(q.V.split_ok and q.V.use_split and q.infeatures == qa.outfeatures for q in qbs)
            and len({(q.infeatures, q.outfeatures, q.bits) for q in qbs}) == 1 and 1 <= len(qbs) <= 3)


def packed_v_stage(qls, x, ln=None):
    """launch 1 of packed_forward_fused on its own: xt_i = V_i (LayerNorm(x) (/) s_i), bf16."""
    rows, d = x.shape[0], qls[0].infeatures
    xts = [torch.empty((rows, d), dtype=torch.bfloat16, device=x.device) for _ in qls]
    x = x.contiguous()
    ops.ortho_apply_ops([(q.V, q.V.small_op(x, xt, colscale=q.inv_scaleWH, ln=_ln_params(ln)), False) for q, xt in zip(qls, xts)], rows)
    return xts


def vgemm_fusable(qls, rows):
    """can launches 1 + 2 run as one (quipamd_dequant_gemm_vop)?  d = 2048 (64 x 32 operator), 2-bit, a few rows."""
    q0 = qls[0]
    return (all(_fusable(q, rows) and q.V.split_ok and q.V.use_split and q.bits == 2 for q in qls) and rows <= 8
            and q0.infeatures == 2048 and (q0.V.p, q0.V.q) == (64, 32) and q0.outfeatures % 32 == 0 and len(qls) <= 3
            and len({(q.infeatures, q.outfeatures) for q in qls}) == 1)


def packed_vgemm_stage(qls, x, ln=None):
    """launches 1 + 2 as ONE: y_i = What_i V_i (LayerNorm(x) (/) s_i), fp32 -- the operator runs in the prologue of every
    workgroup of the dequant-GEMM while its weights are in flight; xt never exists in memory.  Bit-identical to
    packed_v_stage + packed_gemm_stage."""
    rows, m = x.shape[0], qls[0].outfeatures
    x = x.contiguous()
    ys = [torch.empty((rows, m), dtype=torch.float32, device=x.device) for _ in qls]
    vops = [q.V.small_op(x, None, colscale=q.inv_scaleWH, ln=_ln_params(ln), out_dtype=torch.bfloat16) for q in qls]
    ops.dequant_gemm_vop(vops, [q.qweight for q in qls], [q.scales for q in qls], None, ys, rows, m)
    return ys


def packed_gemm_stage(qls, xts):
    """launch 2: y_i = What_i xt_i (grouped fused dequant-GEMM, fp32)."""
    rows, m = xts[0].shape[0], qls[0].outfeatures
    ys = [torch.empty((rows, m), dtype=torch.float32, device=xts[0].device) for _ in qls]
    ops.dequant_gemm_grouped(xts, [q.qweight for q in qls], qls[0].bits, 'b', [q.scales for q in qls], None, ys, m)
    return ys


def packed_u_stage(qls, ys, dtype, residual=None, relu=False):
    """launch 3: out_i = [relu](U_i^T y_i + bias_i + residual)."""
    rows, m = ys[0].shape
    outs = [torch.empty((rows, m), dtype=dtype, device=ys[0].device) for _ in qls]
    res = None if residual is None else residual.contiguous()
    ops.ortho_apply_ops([(q.U, q.U.small_op(y, o, transpose=True, bias=q.bias if q.bias is not None else q.U.zero_bias(), residual=res, relu=relu), True) for q, y, o in zip(qls, ys, outs)], rows)
    return outs


def packed_u_then_v(qa, y, dtype, qbs, residual=None, relu=False, ln=None, store=True):
    """launch 3 of layer `qa` and launch 1 of the layers `qbs` that consume its output, as ONE launch
    (quipamd_ortho_apply_small_chain):  t = [relu](U_a^T y + bias_a + residual);  xt_i = V_i (LayerNorm(t) (/) s_i).
    Returns (t or None when store=False, [xt_i]); bit-identical to packed_u_stage followed by packed_v_stage."""
    rows, m = y.shape
    t = torch.empty((rows, m), dtype=dtype, device=y.device) if store else None
    xts = [torch.empty((rows, m), dtype=torch.bfloat16, device=y.device) for _ in qbs]
    res = None if residual is None else residual.contiguous()
    first = qa.U.small_op(y, t, transpose=True, bias=qa.bias, residual=res, relu=relu, out_dtype=dtype, ld=m)
    seconds = [q.V.small_op(None, xt, colscale=q.inv_scaleWH, ln=_ln_params(ln)) for q, xt in zip(qbs, xts)]
    ops.ortho_small_chain(first, seconds, rows)
    return t, xts


def fused_ok(qls, rows, x_dtype=torch.float16, prev=None, norm=True, residual=True):
    """can `fused_stage` run these layers?  (csrc/decode_fused.hip: 2- / 3- / 4-bit qfn-b layers of one shape sharing their input, Kronecker
    operators of a decode shape on both sides, fp16 activations, a handful of rows; `prev`: the packed layer whose output-side
    operator rides in the prologue -- its U must have the consumers' V shape.  Kernels exist for the combinations a decoder block
    needs: d = 2048 / 4096: (prev + residual, norm or not), (no prev, norm or not); d = 8192: (prev, no residual, no norm),
    (no prev, no norm).)"""
    q0 = qls[0]
    ok = (1 <= len(qls) <= 3 and rows <= ops.FUSED_OPS_MAX_ROWS and x_dtype == torch.float16
          and all(q.bits in (2, 3, 4) and q.qfn == 'b' and q.V is not None and q.V.fused_ok and q.scales.numel() == 1 for q in qls)
          and len({(q.infeatures, q.outfeatures, q.V.p, q.V.q, q.bits) for q in qls}) == 1
          and q0.outfeatures % (32 if (q0.V.p, q0.V.q) == (64, 32) else 16) == 0)
    if ok and prev is not None:
        ok = prev.U is not None and prev.U.fused_ok and (prev.U.p, prev.U.q) == (q0.V.p, q0.V.q)
    if ok:
        big = (q0.V.p, q0.V.q) == (128, 64)
        ok = (not norm and (prev is None or not residual)) if big else (prev is None or residual)
    return ok


def bias16(ql):
    """the layer's bias as fp16 [out] on its device (zeros when it has none), kept on the module: what the fused decode launches read
    (OPT's biases ARE fp16 values, held as fp32 for the K3 kernels: exact)"""
    b = ql.__dict__.get('_bias16')
    dev = ql.qweight.device
    if b is None or b.device != dev:
        b = (ql.bias.to(torch.float16) if ql.bias is not None else torch.zeros(ql.outfeatures, dtype=torch.float16, device=dev)).contiguous()
        ql.__dict__['_bias16'] = b
    return b


def fused_attention(qkv, ys, kcache, vcache, pos, cos_table=None, sin_table=None):
    """decode attention with U_q^T, U_k^T, U_v^T (+ bias) of the three packed projections `qkv` in its prologue; ys: their fp16 GEMM
    outputs from fused_stage(..., y_dtype=torch.float16)"""
    return ops.decode_attention_fused([q.U for q in qkv], ys, [bias16(q) for q in qkv], kcache, vcache, pos, cos_table, sin_table)


def packed_v_stage_gate(ql, gate, up, out_dtype=torch.bfloat16):
    """x~ = V (silu(gate) * up (/) s) in out_dtype: the activation side of Llama's down_proj with the MLP's elementwise product formed on load
    (csrc/ortho_bigp.hip for the 688 x 16 operator; torch ops + the general K3 launch otherwise)"""
    rows = gate.shape[0]
    V = ql.V
    if (V.bigp_ok and not V.small_ok and rows <= ops.BIGP_ROWS and gate.dtype == torch.float16 and up.dtype == torch.float16
            and gate.stride(0) == up.stride(0)):
        xt = torch.empty((rows, ql.infeatures), dtype=out_dtype, device=gate.device)
        cs = ql.inv_scaleWH if ql.inv_scaleWH is not None else V.one_scale()
        ent = [(V, V.small_op(gate.contiguous(), xt, colscale=cs, residual=up.contiguous(), relu=True), False)]
        if all(ops._tile_form(d_) is not None for _, d_, _ in ent):
            ops.ortho_apply_ops(ent, rows)
            return xt
    return V.apply_rows(torch.nn.functional.silu(gate) * up, colscale=ql.inv_scaleWH, out_dtype=out_dtype)


def bigp_tail_ok(ups, down, rows):
    """can `fused_bigp_tail` run this MLP tail?  (csrc/decode_bigp.hip: 1..2 producers whose output-side operators are p x 16 with the
    shape of the consumer's activation-side operator, 2-bit qfn-b consumer with one scale, a handful of rows)"""
    V = down.V
    return (1 <= len(ups) <= 2 and rows <= ops.BIGP_MAX_ROWS and V is not None and V.bigp_fold_ok and down.bits in (2, 3, 4) and down.qfn == 'b'
            and down.scales.numel() == 1 and down.outfeatures % 256 == 0
            and all(q.U is not None and q.U.bigp_fold_ok and (q.U.p, q.U.q) == (V.p, V.q) for q in ups))


def _bigp_tail_tables(ups, down):
    """per (producers, consumer): for every producer the uint16 map image position of U^T's result -> index in the TRANSPOSED input image
    of the consumer's V (b * p + a of the position inv_pin_V[i] natural index i lands on), its bias in image order, and -- on the last
    producer -- 1 / scaleWH of the consumer in image order (the column rescale rides on `up`: silu(g) * (u / s))"""
    cache = down.__dict__.setdefault('_bigp_tail', {})
    key = tuple((id(q), q.__dict__.get('_pack_gen', 0)) for q in ups) + (str(down.V.device),)
    if key not in cache:
        V = down.V
        dev, n, p = V.device, V.n, V.p
        ident = torch.arange(n, device=dev)
        inv_pin_v = ident if V.inv_pin is None else V.inv_pin.long()
        ent = []
        for i, q in enumerate(ups):
            nat = ident if q.U.pin is None else q.U.pin.long()                  # natural index of the element at image position pos
            pos_d = inv_pin_v[nat]
            dest = ((pos_d % 16) * p + pos_d // 16).to(torch.int16).contiguous()   # n <= 16384: the bits of a uint16
            bias_img = None if q.bias is None else bias16(q)[nat].contiguous()
            post = None
            if i == len(ups) - 1 and down.inv_scaleWH is not None:
                post = down.inv_scaleWH.float()[nat].contiguous()
            ent.append((dest, bias_img, post))
        cache[key] = ent
    return cache[key]


def fused_bigp_tail(ups, down, ys, row_tiles_per_wave=0):
    """Llama's MLP tail  down_proj(silu(gate_proj(x)) * up_proj(x))  behind the gate / up GEMM as TWO launches (csrc/decode_bigp.hip):
        [U_gate^T y_gate -> g,  (U_up^T y_up) (/) s_down -> u]      quipamd_decode_bigp_u       (also zeroes the accumulator)
        [x~ = V_down (silu(g) * u),  y_down += What_down x~]         quipamd_decode_bigp_v_gemm  (K-slices meet through fp32 atomics)
    ups = [gate, up] (or one layer: no gating), ys = their fp16 outputs in ZT order (fused_stage with y_dtype=torch.float16).
    Returns y_down fp32 [rows, m] in ZT order of down's U -- what fused_stage(prev=down, y_prev=...) / fused_u_only take."""
    rows = ys[0].shape[0]
    V = down.V
    tabs = _bigp_tail_tables(ups, down)
    dev = ys[0].device
    imgs = torch.empty((len(ups), rows, V.n), dtype=torch.float16, device=dev)
    yd = torch.empty((rows, down.outfeatures), dtype=torch.float32, device=dev)
    if rows >= two_launch_from(down):
        # from TWO_LAUNCH_ROWS rows on: the activation-side pass as its own launch (x~ through a scratch the layer keeps: a hipGraph replays the same
        # pointers), then the ordinary dequant-GEMM -- no atomics, deterministic
        key = ('xt', rows, str(dev))
        ws = down.__dict__.setdefault('_splitk_ws', {})
        if key not in ws:
            ws[key] = torch.empty((rows, V.n), dtype=torch.float16, device=dev)
        ops.decode_bigp_u([(q.U, y, bias_img, post, dest, imgs[i]) for i, (q, y, (dest, bias_img, post)) in enumerate(zip(ups, ys, tabs))], rows)
        ops.decode_bigp_v_gemm(V, imgs[0], imgs[1] if len(ups) == 2 else None, down.decode_qweight(), down.scales, yd, row_tiles_per_wave, bits=down.bits,
                               xt=ws[key])
        return yd
    if DETERMINISTIC_SPLITK:
        # the K-slices of down_proj meet in slice order through a scratch the layer keeps
        key = (rows, str(dev))
        ws = down.__dict__.setdefault('_splitk_ws', {})
        if key not in ws:
            ws[key] = torch.empty((V.p // 16, rows, down.outfeatures), dtype=torch.float32, device=dev)
        ops.decode_bigp_u([(q.U, y, bias_img, post, dest, imgs[i]) for i, (q, y, (dest, bias_img, post)) in enumerate(zip(ups, ys, tabs))], rows)
        ops.decode_bigp_v_gemm(V, imgs[0], imgs[1] if len(ups) == 2 else None, down.decode_qweight(), down.scales, yd, row_tiles_per_wave, bits=down.bits,
                               partials=ws[key])
        return yd
    ops.decode_bigp_u([(q.U, y, bias_img, post, dest, imgs[i]) for i, (q, y, (dest, bias_img, post)) in enumerate(zip(ups, ys, tabs))], rows, clear=yd)
    ops.decode_bigp_v_gemm(V, imgs[0], imgs[1] if len(ups) == 2 else None, down.decode_qweight(), down.scales, yd, row_tiles_per_wave, bits=down.bits)
    return yd


def fused_u_only(ql, y, residual=None, relu=False):
    """out = [relu](U^T y + bias + residual), fp16: the output side of packed layer `ql` on its own; y fp16 in ZT order"""
    return ops.decode_u_only(ql.U, y, bias16(ql), residual=None if residual is None else residual.contiguous(), relu=relu)


def fused_head_ok(ql, rows, ln):
    """can `fused_head` finish a decode step behind packed layer `ql`?  (csrc/decode_head.hip: U 64 x 32 / 64 x 64, a norm, <= 4 rows)"""
    return (ql.U is not None and ql.U.fused_ok and (ql.U.p, ql.U.q) in ((64, 32), (64, 64)) and rows <= ops.FUSED_MAX_ROWS and _ln_params(ln) is not None)


def fused_head(ql, y, residual, ln, W, logits, part_val=None, part_idx=None, pos_inc=None):
    """the end of a decode step as ONE launch (quipamd_decode_head): [U^T y + bias + residual] -> final norm -> logits = W h (+ the
    argmax partials the next step's ops.decode_embed turns into the token, + pos += 1).  y: fp16 / fp32 in ZT order of ql's U."""
    g, b, eps = _ln_params(ln)
    ops.decode_head(W, logits, g, b, eps, U=ql.U.fop(True), u_y=y.contiguous(), u_bias=bias16(ql),
                    u_residual=None if residual is None else residual.contiguous(), part_val=part_val, part_idx=part_idx, pos_inc=pos_inc)
    return logits


def fused_attention_ok(qkv, kcache):
    bs, heads, maxlen, hd = kcache.shape
    shapes = {(q.U.p, q.U.q) for q in qkv if q.U is not None}
    return (len(qkv) == 3 and all(q.U is not None and q.U.fused_ok for q in qkv) and len(shapes) == 1 and kcache.dtype == torch.float16
            and next(iter(shapes)) in (((64, 32),) if hd == 64 else ((64, 64), (64, 32)) if hd == 128 else ())
            and all(q.outfeatures == heads * hd for q in qkv))


def fused_stage(qls, x=None, prev=None, y_prev=None, residual=None, relu=False, ln=None, store=False, y_dtype=torch.float32):
    """ONE launch (quipamd_decode_fused_gemm) for everything between two dequant-GEMMs of a decode step:
        t    = [relu]( U_prev^T y_prev + bias_prev + residual )     when `prev` (the packed layer that produced y_prev) is given,
                                                                    else t = x
        y_i  = What_i V_i ( Norm(t) (/) s_i )                       for the 1..3 layers `qls` sharing t; [rows, m] in y_dtype
    Returns (ys, t): t is the fp16 tensor written by the launch when store=True (the new residual stream), else None.
    ys[i] and y_prev are in ZT ORDER of the producing layer's U (QuantLinear.to_zt / from_zt; the launches read decode_qweight(), the
    codes with both permutations folded in) -- only fused_stage / fused_attention / fused_u_only consume them.
    ln: nn.LayerNorm / an RMSNorm module / None.  y_prev: fp16 (what a fused launch writes with y_dtype=torch.float16; an fp32
    y_prev is rounded here -- the kernel's first step is that rounding anyway).  y_dtype fp16 when the consumer is another fused
    launch, fp32 for the K3 operator kernels."""
    q0 = qls[0]
    rows = (x if prev is None else y_prev).shape[0]
    dev = q0.qweight.device
    m, d = q0.outfeatures, q0.infeatures
    ys = [torch.empty((rows, m), dtype=y_dtype, device=dev) for _ in qls]
    # more rows than the single launch walks in its prologue (ops.FUSED_MAX_ROWS): the same prologue as its own launch, one workgroup per
    # row (ops_only), x~ through global memory, then ONE grouped dequant-GEMM on the same decode-order codes -- the weights stream once for
    # all rows; 2 launches per layer group instead of the 3 of the operator / GEMM / operator form
    two = rows >= two_launch_from(q0)
    xts = [torch.empty((rows, d), dtype=torch.float16, device=dev) for _ in qls] if two else None
    kw = dict(V=[q.V.fop(False) for q in qls], colscale=[q.inv_scaleWH if q.inv_scaleWH is not None else q.V.one_scale() for q in qls],
              qweight=[q.decode_qweight() for q in qls], scale=[q.scales for q in qls], y=xts if two else ys, m=m, bs=rows, bits=q0.bits,
              ops_only=two)
    lnp = _ln_params(ln)
    if lnp is not None:
        g, b, eps = lnp
        kw.update(norm=1 if b is not None else 2, ln_gamma=g, ln_beta=b, ln_eps=eps)
    t = None
    if prev is None:
        kw.update(x=x.contiguous())
    else:
        t = torch.empty((rows, d), dtype=torch.float16, device=dev) if store else None
        if (q0.V.p, q0.V.q) == (128, 64) and len(qls) == 1 and residual is None and lnp is None and rows <= 2 and not store and not two:
            # n = 8192 (OPT fc1 -> fc2): the layer pair's per-lane tables turn gather + scale + scatter into one scatter (decode_fused.hip)
            cache = q0.__dict__.setdefault('_pair_tables', {})
            key = (id(prev), prev.__dict__.get('_pack_gen', 0), str(dev))
            if key not in cache:
                cache[key] = ops.pair_tables(prev.U, q0.V, bias16(prev), kw['colscale'][0])
            kw.update(pair=cache[key])
        # an fp32 y_prev is rounded to fp16 -- inside the launch where a kernel for that exists (the accumulator of fused_bigp_tail
        # feeding Llama's q / k / v), by a cast otherwise
        in_kernel = (y_prev.dtype == torch.float32 and (q0.V.p, q0.V.q) == (64, 64) and residual is not None and lnp is not None and lnp[1] is None
                     and (q0.bits == 2 or two))
        kw.update(U=prev.U.fop(True), u_y=(y_prev if in_kernel else y_prev.to(torch.float16)).contiguous(), u_bias=bias16(prev),
                  u_residual=None if residual is None else residual.contiguous(), u_relu=relu, t_out=t)
    ops.decode_fused_gemm(**kw)
    if two:
        ops.dequant_gemm_grouped(xts, kw['qweight'], q0.bits, 'b', kw['scale'], None, ys, m)
    return ys, t


def stage_operands(qls, prev=None, ln=None):
    """the device tensors the prologue of fused_stage(qls, prev=prev, ln=ln) reads that do NOT depend on the step -- factor fragments, index
    vectors, bias, gains, column scales, grid scales (50-150 KB) -- for ops.decode_prefetch_next (csrc/prefetch.h): in a decode step they are
    cold in L2 once per token, and an earlier launch's spare workgroups can pull them in"""
    out = []
    if prev is not None:
        prev.U.fop(True)
        F0, F1, _, st = prev.U._fops[True][1]
        out += [F0, F1, st, bias16(prev)]
    lnp = _ln_params(ln)
    if lnp is not None:
        out += [lnp[0]] + ([lnp[1]] if lnp[1] is not None else [])
    for q in qls:
        q.V.fop(False)
        F0, F1, ld, _ = q.V._fops[False][1]
        out += [F0, F1, ld, q.inv_scaleWH if q.inv_scaleWH is not None else q.V.one_scale(), q.scales]
    if prev is not None and len(qls) == 1:
        for tabs in qls[0].__dict__.get('_pair_tables', {}).values():     # the fc1 -> fc2 pair launch reads its per-lane tables instead
            out += list(tabs)
    return out


def attention_operands(qkv):
    """the same for fused_attention(qkv, ...): the transposed output-side operators of q / k / v and their biases"""
    out = []
    for l in qkv:
        l.U.fop(True)
        F0, F1, _, st = l.U._fops[True][1]
        out += [F0, F1, st, bias16(l)]
    return out


def save_packed(layers, path):
    """Packed checkpoint: {dotted module name: QuantLinear} -> one torch file of CPU tensors (replaces the dense fp16
    `torch.save(model.state_dict())` of opt.py:644-646 for the quantised Linears; 2 bits/weight + factors)."""
    torch.save({name: ql.packed_state() for name, ql in layers.items()}, path)


def load_packed(path, device):
    """inverse of save_packed: {name: QuantLinear on `device`}, ready for make_quant(model, layers)."""
    # the record holds tensors, lists, tuples, ints, strings and None only: no need for (unsafe) full unpickling
    return {name: QuantLinear.from_packed_state(st, device) for name, st in torch.load(path, map_location='cpu', weights_only=True).items()}


def save_model(model, layers, path):
    """ONE file for a quantised model -- the role of `--save` (opt.py:644-646: torch.save of the dense state dict): the packed records of
    `layers` ({dotted name: QuantLinear}, e.g. decode.collect_packed().named(model)) + every tensor of `model.state_dict()` that does NOT
    belong to one of those modules (embeddings, norms, head, Linears that stayed dense).  Works whether or not `layers` is installed in
    `model` yet."""
    own = tuple(n + "." for n in layers)
    rest = {k: v.detach().cpu() for k, v in model.state_dict().items() if not k.startswith(own)}
    torch.save({"format": "quip_amd.model.v1", "packed": {name: ql.packed_state() for name, ql in layers.items()}, "rest": rest}, path)


def load_model(model, path, device):
    """inverse of save_model onto a fresh skeleton of the same architecture -- the role of `load_quant` (opt.py:350-381: build the model,
    make_quant, load_state_dict): swaps the packed layers in (make_quant), loads the remaining tensors, moves the model to `device`.
    Returns {dotted name: QuantLinear}; `decode.DecodeEngine.from_hf(model)` then serves it."""
    blob = torch.load(path, map_location='cpu', weights_only=True)
    assert blob.get("format") == "quip_amd.model.v1", "not a quip_amd.quant.save_model file"
    layers = {name: QuantLinear.from_packed_state(st, device) for name, st in blob["packed"].items()}
    make_quant(model, layers)
    mods = dict(model.named_modules())
    lost = [n for n, ql in layers.items() if mods.get(n) is not ql]       # make_quant skips names the skeleton does not have
    if lost:
        raise RuntimeError(f"load_model: the skeleton has no module for {len(lost)} packed layer(s) of the file (first: {lost[:3]})")
    missing, unexpected = model.load_state_dict(blob["rest"], strict=False)
    own = tuple(n + "." for n in layers)
    stray = [k for k in missing if not k.startswith(own)]
    if unexpected or stray:
        raise RuntimeError(f"load_model: the skeleton does not match the file (unexpected {list(unexpected)[:3]}, missing {stray[:3]})")
    model.to(device)
    return layers


def make_quant(module, layers, name=''):
    """Swap the named nn.Linear modules for packed QuantLinear layers (the role of make_quant3 / make_quant4,
    quant.py:236-246).  `layers`: {dotted name: QuantLinear already packed}."""
    if isinstance(module, QuantLinear):
        return
    for child_name, child in list(module.named_children()):
        full = f"{name}.{child_name}" if name else child_name
        if full in layers:
            setattr(module, child_name, layers[full])
        else:
            make_quant(child, layers, full)
