"""Thin torch-tensor wrappers over the C ABI (include/quip_amd.h).

torch is used for device memory and the current HIP stream only; every op below launches a hand-written
gfx950 kernel from libquip_amd.so.  Inputs must live on a GPU -- there is no CPU fallback (a CPU tensor
raises), and a missing library raises at first use (quip_amd/_lib.py).
"""
import ctypes

import torch

from . import _lib

LAYOUT_CANONICAL, LAYOUT_STREAM = 0, 1
QFN = {'a': 0, 'b': 1, 'c': 2}
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("quip_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dtype(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}")


def container_bits(bits):
    """storage bits per code in the STREAM layout: 3-bit codes ride in the 4-bit container."""
    return 4 if bits == 3 else bits


def stream_chunk(bits):
    """columns per 16-row STREAM tile (include/quip_amd.h)."""
    return 512 // container_bits(bits)


# ------------------------------------------------------------------------------------------------- K1
def pack(codes, bits, layout=LAYOUT_CANONICAL):
    """codes uint8 [m,d] -> int32 words; CANONICAL returns [d*bits/32, m] (zeroShot/models/quant.py:190-199; bits = 3: the
    reference's 32-codes-in-3-words rule, quant.py:192-220)."""
    _need_gpu(codes)
    assert codes.dtype == torch.uint8 and codes.dim() == 2
    codes = codes.contiguous()
    m, d = codes.shape
    shape = (d * bits // 32, m) if layout == LAYOUT_CANONICAL else (m * d * container_bits(bits) // 32,)
    out = torch.empty(shape, dtype=torch.int32, device=codes.device)
    _lib.call("quipamd_pack", _p(codes), bits, layout, _p(out), m, d, _stream())
    return out


def unpack(packed, bits, layout, m, d):
    _need_gpu(packed)
    assert packed.dtype == torch.int32
    packed = packed.contiguous()
    assert packed.numel() == m * d * (bits if layout == LAYOUT_CANONICAL else container_bits(bits)) // 32
    codes = torch.empty((m, d), dtype=torch.uint8, device=packed.device)
    _lib.call("quipamd_unpack", _p(packed), bits, layout, _p(codes), m, d, _stream())
    return codes


def repack_canonical_to_stream(qweight, bits, m, d):
    """a weight in the reference's CANONICAL packing (int32 [d*bits/32, m]; bits 2, 3 or 4) -> the STREAM layout K2 reads,
    on the device (3-bit codes land in the 4-bit container)."""
    _need_gpu(qweight)
    assert qweight.dtype == torch.int32 and qweight.numel() == m * d * bits // 32
    qweight = qweight.contiguous()
    out = torch.empty(m * d * container_bits(bits) // 32, dtype=torch.int32, device=qweight.device)
    _lib.call("quipamd_repack_canonical_to_stream", _p(qweight), bits, _p(out), m, d, _stream())
    return out


_VQ_WORKSPACES = {}      # mat.data_ptr() -> (weakref to mat, mat._version, bits, workspace): the library repacks once per workspace


def vecquantmatmul(bits, vec, mat, mul, scales, zeros):
    """quant_cuda.vecquant{3,4}matmul(vec, mat, mul, scales, zeros) (quant.py:229, zeroShot/models/quant.py:207): accumulates
    into `mul` (float32 [m], pre-filled with the bias) and returns None like the reference's extension.  Every `mat` keeps its
    own workspace alive for as long as the tensor lives and registers it with quipamd_vecquant_prepare, so the CANONICAL -> STREAM
    repack runs on the first call only (a decode loop calls this once per token per layer); an in-place change of `mat` that torch's
    version counter sees, or a new tensor at the same address, prepares again.  NOT seen: writes through `mat.data` / `.detach()`
    views and raw-pointer writes (they do not bump `_version`) -- call `ops.vecquant_forget(mat)` after those.  The C entry points
    themselves are stateless unless prepared (include/quip_amd.h)."""
    import weakref
    _need_gpu(vec, mat, mul)
    assert bits in (3, 4) and mul.dtype == torch.float32 and mul.is_contiguous() and mat.dtype == torch.int32
    d, m = vec.numel(), mul.numel()
    vec = vec.reshape(-1).to(torch.float32).contiguous()
    sc, zs = _f32vec(scales, mul.device), _f32vec(zeros, mul.device)
    assert sc.numel() == m and zs.numel() == m and mat.numel() == m * d * bits // 32
    lib = _lib.load()
    nbytes = lib.quipamd_vecquant_workspace_bytes(bits, m, d)
    matc = mat if mat.is_contiguous() else mat.contiguous()
    ws = None
    if matc is mat:
        ent = _VQ_WORKSPACES.get(mat.data_ptr())
        if ent is not None and ent[0]() is mat and ent[1] == mat._version and ent[2] == bits and ent[3].numel() == nbytes:
            ws = ent[3]
        else:
            if ent is not None:
                lib.quipamd_vecquant_invalidate(_p(ent[3]))
            for k in [k for k, e in _VQ_WORKSPACES.items() if e[0]() is None]:        # layers that are gone
                lib.quipamd_vecquant_invalidate(_p(_VQ_WORKSPACES.pop(k)[3]))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=mul.device)
            _lib.call("quipamd_vecquant_prepare", bits, _p(matc), m, d, _p(ws), nbytes, _stream())   # opt in: repack once, here
            _VQ_WORKSPACES[mat.data_ptr()] = (weakref.ref(mat), mat._version, bits, ws)
    else:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=mul.device)
        lib.quipamd_vecquant_invalidate(_p(ws))
    _lib.call(f"quipamd_vecquant{bits}matmul", _p(vec), _p(matc), _p(mul), _p(sc), _p(zs), m, d, _p(ws), nbytes, _stream())


def vecquant_forget(mat=None):
    """drop the prepared workspace of `mat` (all of them when None): the next vecquantmatmul repacks from the tensor's current contents"""
    lib = _lib.load()
    keys = list(_VQ_WORKSPACES) if mat is None else [mat.data_ptr()]
    for k in keys:
        ent = _VQ_WORKSPACES.pop(k, None)
        if ent is not None:
            lib.quipamd_vecquant_invalidate(_p(ent[3]))


# ------------------------------------------------------------------------------------------------- K5
def qfnb_scale(W):
    """2.4*rms(W)+1e-16 in W's dtype (quant.py:150); returns a float32 device tensor [1]."""
    _need_gpu(W)
    W = W.contiguous()
    out = torch.empty(1, dtype=torch.float32, device=W.device)
    ws = torch.empty(1, dtype=torch.float64, device=W.device)
    _lib.call("quipamd_qfnb_scale", _p(W), _dtype(W), W.numel(), _p(out), _p(ws), _stream())
    return out


def _f32vec(t, dev):
    return None if t is None else t.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()


def _grid(scale, zero, m, qfn, dev):
    """Grid parameters as the kernels index them: float32, contiguous, scale[1] for qfn b (scalar grid, quant.py:13-14),
    scale[m] / zero[m] per output row for qfn a / c (quant.py:124-126).  A scalar is expanded to m rows; anything else
    (the per-column shapes Quantizer.find_params(weight=False) can produce) is refused instead of being read out of bounds."""
    scale, zero = _f32vec(scale, dev), _f32vec(zero, dev)
    if scale is None:
        raise ValueError("grid scale is None (Quantizer.find_params not called?)")
    if qfn == 'b':
        if scale.numel() != 1:
            raise ValueError(f"qfn b takes one scalar scale, got {scale.numel()} values")
        return scale, zero
    def rows(t, what):
        if t is None:
            raise ValueError(f"qfn {qfn} needs {what}")
        if t.numel() == 1 and m != 1:
            return t.expand(m).contiguous()
        if t.numel() != m:
            raise ValueError(f"{what} has {t.numel()} values for {m} rows: only per-row (or scalar) grids are supported")
        return t
    return rows(scale, "scale"), rows(zero, "zero")


def gridmap(W, qfn, scale, zero, maxq):
    """grid coordinates float32 [m,d] (vector_balance.py:515, 522-524)."""
    _need_gpu(W)
    W = W.contiguous()
    m, d = W.shape
    scale, zero = _grid(scale, zero, m, qfn, W.device)
    out = torch.empty((m, d), dtype=torch.float32, device=W.device)
    _lib.call("quipamd_gridmap", _p(W), _dtype(W), QFN[qfn], _p(scale), _p(zero), int(maxq), _p(out), m, d, _stream())
    return out


def quantize(W, qfn, scale, zero, maxq, want_codes=False):
    """round-to-nearest through the grid (quant.py:6-21).  Returns dequantised W (same dtype) [, codes]."""
    _need_gpu(W)
    W2 = W.contiguous().reshape(W.shape[0], -1)
    m, d = W2.shape
    scale, zero = _grid(scale, zero, m, qfn, W.device)
    out = torch.empty_like(W2)
    codes = torch.empty((m, d), dtype=torch.uint8, device=W.device) if want_codes else None
    _lib.call("quipamd_quantize", _p(W2), _dtype(W2), QFN[qfn], _p(scale), _p(zero), int(maxq), _p(codes), _p(out),
              m, d, _stream())
    out = out.reshape(W.shape)
    return (out, codes) if want_codes else out


def codes_to_weight(codes, qfn, scale, zero, maxq, out_dtype=torch.float16):
    _need_gpu(codes)
    codes = codes.contiguous()
    m, d = codes.shape
    scale, zero = _grid(scale, zero, m, qfn, codes.device)
    out = torch.empty((m, d), dtype=out_dtype, device=codes.device)
    _lib.call("quipamd_codes_to_weight", _p(codes), QFN[qfn], _p(scale), _p(zero), int(maxq), _p(out), _DT[out_dtype],
              m, d, _stream())
    return out


# ------------------------------------------------------------------------------------------------- K2
def dequant_gemm(x, qweight, bits, qfn, scale, zero, bias, out=None, out_dtype=None, accumulate=False,
                 m=None, cfg=None):
    """y[bs,m] = x[bs,d] @ dequant(qweight)^T + bias; qweight in STREAM layout.  x: bf16 or fp16 (the MFMA runs in x's
    dtype); y: x's dtype (default) or fp32.  cfg = (family, p1, p2): force a kernel (include/quip_amd.h), for benchmarks."""
    _need_gpu(x, qweight)
    assert x.dim() == 2 and x.dtype in (torch.bfloat16, torch.float16), "x must be bf16 or fp16"
    x = x.contiguous()
    bs, d = x.shape
    if m is None:
        m = qweight.numel() * 32 // container_bits(bits) // d
    dev = x.device
    scale, zero = _grid(scale, zero, m, qfn, dev)
    bias = _f32vec(bias, dev)
    if bias is not None and bias.numel() != m:
        raise ValueError(f"bias has {bias.numel()} values for {m} rows")
    if out is None:
        assert not accumulate
        out = torch.empty((bs, m), dtype=out_dtype or x.dtype, device=dev)
    assert out.is_contiguous() and out.shape == (bs, m)
    if cfg is None:
        _lib.call("quipamd_dequant_gemm", _p(x), _dtype(x), _p(qweight), bits, LAYOUT_STREAM, QFN[qfn], _p(scale),
                  _p(zero), _p(bias), _p(out), _dtype(out), int(bool(accumulate)), bs, m, d, _stream())
    else:
        c = (ctypes.c_int32 * 4)(*(list(cfg) + [0, 0, 0, 0])[:4])
        _lib.call("quipamd_dequant_gemm_cfg", _p(x), _dtype(x), _p(qweight), bits, LAYOUT_STREAM, QFN[qfn], _p(scale),
                  _p(zero), _p(bias), _p(out), _dtype(out), int(bool(accumulate)), bs, m, d,
                  ctypes.cast(c, ctypes.c_void_p), _stream())
    return out


def dequant_gemm_grouped(xs, qweights, bits, qfn, scales, zeros, outs, m):
    """ngroups <= 4 problems of identical shape in one launch: xs[i] [bs,d] bf16, outs[i] [bs,m] (fp32 or bf16)."""
    n = len(xs)
    bs, d = xs[0].shape
    vp = ctypes.c_void_p
    for t in list(scales) + (list(zeros) if zeros is not None else []):
        _f32ptr(t, "scale / zero")
    arr = lambda ts: (vp * n)(*[vp(0 if t is None else t.data_ptr()) for t in ts])
    zs = arr(zeros) if zeros is not None and zeros[0] is not None else None
    _lib.call("quipamd_dequant_gemm_grouped", n, arr(xs), _dtype(xs[0]), arr(qweights), bits, LAYOUT_STREAM, QFN[qfn], arr(scales),
              zs, None, arr(outs), _dtype(outs[0]), 0, bs, m, d, _stream())
    return outs


def dequant_gemm_grouped_config(form=0):
    """tests / A-B runs: 0 heuristic; 1 | 2 | 4 grouped h kernel row tiles; 74 | 72 | 81 grouped weight-stream kernel, 4 | 7 | 8 row tiles (process-wide)"""
    _lib.load().quipamd_dequant_gemm_grouped_config(int(form))


# ------------------------------------------------------------------------------------------------- K3
class SmallOp(ctypes.Structure):
    """mirror of `quipamd_small_op` (include/quip_amd.h)."""
    _fields_ = [("M0", ctypes.c_void_p), ("M1", ctypes.c_void_p), ("load_idx", ctypes.c_void_p), ("store_idx", ctypes.c_void_p),
                ("p", ctypes.c_int), ("q", ctypes.c_int), ("b_first", ctypes.c_int),
                ("colscale", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("ln_gamma", ctypes.c_void_p), ("ln_beta", ctypes.c_void_p), ("ln_eps", ctypes.c_float), ("ln_dtype", ctypes.c_int),
                ("residual", ctypes.c_void_p), ("res_dtype", ctypes.c_int), ("relu", ctypes.c_int),
                ("x", ctypes.c_void_p), ("x_dtype", ctypes.c_int), ("ldx", ctypes.c_int64),
                ("out", ctypes.c_void_p), ("out_dtype", ctypes.c_int), ("ldo", ctypes.c_int64),
                ("M0_hi", ctypes.c_void_p), ("M0_lo", ctypes.c_void_p), ("M1_hi", ctypes.c_void_p), ("M1_lo", ctypes.c_void_p)]


def _ptr(t):
    return None if t is None else t.data_ptr()


def _f32ptr(t, what):
    """raw float* handed to a kernel: the tensor must BE float32 and contiguous (e.g. model.half() must not have touched
    a packed layer's grid buffers -- QuantLinear._apply keeps them fp32)."""
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise TypeError(f"{what} must be a contiguous float32 tensor, got {t.dtype}, contiguous={t.is_contiguous()}")
    return t.data_ptr()


def ortho_small_ops(op_list, rows):
    """up to 4 fused small-batch operator applications in ONE launch (quipamd_ortho_apply_small_ops)."""
    arr = (SmallOp * len(op_list))(*op_list)
    _lib.call("quipamd_ortho_apply_small_ops", ctypes.cast(arr, ctypes.c_void_p), len(op_list), rows, _stream())


TILE_ROWS = 8          # up to this many rows a Kronecker operator is cut into 16 x 16 output tiles, one workgroup each
BIGP_ROWS = 64         # p x 16 operators (Llama's 11008 = 688 x 16) stay on csrc/ortho_bigp.hip -- one workgroup per (16 of p, row) -- up to this many
                       # rows: the general two-launch kernel gives the 688-deep mix to 16 workgroups per 16 rows (~240 us per application;
                       # profiles/r04l: a Llama-2-7B step went from 3.9 ms at 8 sequences to 27 ms at 16)
USE_TILES = True


def ortho_tile_ops(op_list, inv_list, rows):
    """up to 4 operator applications, every 16 x 16 tile of every output image on its own workgroup (quipamd_ortho_apply_tiles):
    the decode-step form of ortho_small_ops.  inv_list[i]: the inverse of op i's store permutation (int32) or None."""
    n = len(op_list)
    arr = (SmallOp * n)(*op_list)
    inv = (ctypes.c_void_p * n)(*[ctypes.c_void_p(0 if t is None else t.data_ptr()) for t in inv_list])
    _lib.call("quipamd_ortho_apply_tiles", ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(inv, ctypes.c_void_p), n, rows, _stream())


def _tile_form(d):
    """which compiled operand set of csrc/ortho_tile.hip a descriptor has: 'V' (activation side), 'U' (output side) or None"""
    if not d.load_idx or not d.store_idx:
        return None
    if d.x_dtype == _DT[torch.float16] and d.colscale and not d.bias and not d.residual and not d.relu \
            and (not d.ln_gamma or d.ln_dtype == _DT[torch.float16]):
        return ('V', bool(d.ln_gamma))
    if d.x_dtype == _DT[torch.float16] and d.colscale and not d.bias and d.residual and d.relu and not d.ln_gamma \
            and d.res_dtype == _DT[torch.float16]:
        return ('G', False)                  # silu(x) * residual on load: the p x 16 kernel only (Llama's down_proj input)
    if d.x_dtype == _DT[torch.float32] and not d.colscale and d.bias and not d.ln_gamma \
            and (not d.residual or d.res_dtype == _DT[torch.float16]):
        return ('U', bool(d.residual))
    return None


def ortho_bigp_ops(op_list, inv_list, rows):
    """operators p x 16 with a large p (11008 = 688 x 16), one workgroup per 16 image rows (quipamd_ortho_apply_bigp)"""
    n = len(op_list)
    arr = (SmallOp * n)(*op_list)
    inv = (ctypes.c_void_p * n)(*[ctypes.c_void_p(0 if t is None else t.data_ptr()) for t in inv_list])
    _lib.call("quipamd_ortho_apply_bigp", ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(inv, ctypes.c_void_p), n, rows, _stream())


def ortho_apply_ops(entries, rows):
    """entries: [(OrthoOp, SmallOp descriptor, transpose)] sharing p, q and dtypes -> ONE launch: tiled over many workgroups for
    a handful of rows (decode), one workgroup per row otherwise."""
    forms = {_tile_form(d) for _, d, _ in entries}
    if all(o.bigp_ok and not o.small_ok for o, _, _ in entries):
        assert rows <= BIGP_ROWS and len(forms) == 1 and None not in forms and not any(d.ln_gamma for _, d, _ in entries), \
            "p x 16 operators: a handful of rows, activation-side or output-side operand set, no normalisation"
        ortho_bigp_ops([d for _, d, _ in entries], [o.store_inv(t) for o, _, t in entries], rows)
        return
    if USE_TILES and rows <= TILE_ROWS and all(o.tile_ok and o.use_split for o, _, _ in entries) and len(forms) == 1 \
            and all(f is not None and f[0] in 'VU' for f in forms):
        ortho_tile_ops([d for _, d, _ in entries], [o.store_inv(t) for o, _, t in entries], rows)
    else:
        assert all(f is None or f[0] != 'G' for f in forms), "the silu-gate input form exists on the p x 16 kernel only"
        ortho_small_ops([d for _, d, _ in entries], rows)


def ortho_small_chain(first, seconds, rows):
    """`first` then each of `seconds` (1..3 SmallOp sharing its result) in ONE launch (quipamd_ortho_apply_small_chain)."""
    one = (SmallOp * 1)(first)
    arr = (SmallOp * len(seconds))(*seconds)
    _lib.call("quipamd_ortho_apply_small_chain", ctypes.cast(one, ctypes.c_void_p), ctypes.cast(arr, ctypes.c_void_p),
              len(seconds), rows, _stream())


def dequant_gemm_vop(vops, qweights, scales, biases, ys, bs, m, bits=2):
    """V-side operator + grouped 2-bit dequant-GEMM in ONE launch (quipamd_dequant_gemm_vop; d = 2048, bs <= 8)."""
    n = len(vops)
    arr = (SmallOp * n)(*vops)
    for t in list(scales) + (list(biases) if biases is not None else []):
        _f32ptr(t, "scale / bias")
    vp = lambda ts: (ctypes.c_void_p * n)(*[0 if t is None else t.data_ptr() for t in ts])
    _lib.call("quipamd_dequant_gemm_vop", ctypes.cast(arr, ctypes.c_void_p), vp(qweights), vp(scales),
              vp(biases if biases is not None else [None] * n), vp(ys), n, bits, bs, m, _stream())


# ---- decode: fused launches (csrc/decode_fused.hip) ---------------------------------------------------------------------------
class Fop(ctypes.Structure):
    """mirror of `quipamd_fop` (include/quip_amd.h)"""
    _fields_ = [("F0", ctypes.c_void_p), ("F1", ctypes.c_void_p), ("load_idx", ctypes.c_void_p), ("store_idx", ctypes.c_void_p),
                ("p", ctypes.c_int), ("q", ctypes.c_int)]


class FusedGemmArgs(ctypes.Structure):
    """mirror of `quipamd_fused_gemm_args`"""
    _fields_ = [("act_dtype", ctypes.c_int), ("bits", ctypes.c_int), ("has_u", ctypes.c_int), ("U", Fop),
                ("u_y", ctypes.c_void_p), ("u_bias", ctypes.c_void_p), ("u_residual", ctypes.c_void_p), ("ld_residual", ctypes.c_int64),
                ("u_relu", ctypes.c_int), ("t_out", ctypes.c_void_p), ("ld_t", ctypes.c_int64),
                ("x", ctypes.c_void_p), ("ldx", ctypes.c_int64),
                ("norm", ctypes.c_int), ("ln_gamma", ctypes.c_void_p), ("ln_beta", ctypes.c_void_p), ("ln_eps", ctypes.c_float),
                ("ngroups", ctypes.c_int), ("V", Fop * 3), ("colscale", ctypes.c_void_p * 3), ("qweight", ctypes.c_void_p * 3),
                ("scale", ctypes.c_void_p * 3), ("y", ctypes.c_void_p * 3), ("y_dtype", ctypes.c_int), ("bs", ctypes.c_int64),
                ("m", ctypes.c_int64), ("pair_sig", ctypes.c_void_p), ("pair_bias", ctypes.c_void_p), ("pair_cs", ctypes.c_void_p),
                ("u_y_dtype", ctypes.c_int), ("ops_only", ctypes.c_int)]


FUSED_SHAPES = ((64, 32), (64, 64), (128, 64))
FUSED_MAX_ROWS = 4          # rows the single fused launch walks in its prologue (csrc/decode_fused.hip FG_MAXBS)
FUSED_OPS_MAX_ROWS = 16     # rows of the two-launch form (prologue-only launch, one workgroup per row, + grouped dequant-GEMM)


def _f16_b_frags(M):
    """M [P, P] fp32 (out index i, in index k) -> fp16 [P/16, P/32, 64, 8] in v_mfma_f32_16x16x32_f16 B-fragment order
    (include/quip_amd.h quipamd_fop): element [t][S][lane][e] = M[16 t + lane % 16][32 S + 8 (lane // 16) + e]"""
    P = M.shape[0]
    F = M.to(torch.float16).view(P // 16, 16, P // 32, 4, 8)          # [t, j, S, g, e]
    return F.permute(0, 2, 3, 1, 4).contiguous().reshape(-1)          # [t, S, g, j, e]: lane = 16 g + j


def _f16_b_frags_padded(M):
    """_f16_b_frags for a p that is a multiple of 16 only: the k index zero padded to ks = ceil(p / 32) steps
    (include/quip_amd.h quipamd_decode_bigp_u): fp16 [p/16, ks, 64, 8]"""
    P = M.shape[0]
    ks = (P + 31) // 32
    Mp = torch.zeros((P, 32 * ks), dtype=torch.float16, device=M.device)
    Mp[:, :P] = M.to(torch.float16)
    F = Mp.view(P // 16, 16, ks, 4, 8)                                 # [t, j, S, g, e]
    return F.permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


class BigpUOp(ctypes.Structure):
    """quipamd_bigp_u_op"""
    _fields_ = [("F0", ctypes.c_void_p), ("M1", ctypes.c_void_p), ("y", ctypes.c_void_p), ("bias_img", ctypes.c_void_p),
                ("post_img", ctypes.c_void_p), ("dest", ctypes.c_void_p), ("out", ctypes.c_void_p), ("ld_out", ctypes.c_int64)]


class BigpVGemmArgs(ctypes.Structure):
    """quipamd_bigp_v_gemm_args"""
    _fields_ = [("F0", ctypes.c_void_p), ("M1", ctypes.c_void_p), ("gate", ctypes.c_void_p), ("up", ctypes.c_void_p), ("ldx", ctypes.c_int64),
                ("qweight", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("bits", ctypes.c_int), ("y", ctypes.c_void_p), ("m", ctypes.c_int64),
                ("p", ctypes.c_int), ("rows", ctypes.c_int64), ("row_tiles_per_wave", ctypes.c_int), ("partials", ctypes.c_void_p),
                ("xt", ctypes.c_void_p)]


BIGP_MAX_ROWS = 16         # csrc/decode_bigp.hip: 1..4 rows per workgroup pass, up to 16 per launch (round 5)


def decode_bigp_u(entries, rows, clear=None):
    """ONE launch for the output-side p x 16 operators of up to three layers (quipamd_decode_bigp_u):
    entries = [(U, y, bias_img | None, post_img | None, dest, out)]: U an OrthoOp with fold_ok (applied TRANSPOSED), y fp16 [rows, n] in ZT
    order of U, bias_img fp16 [n] / post_img fp32 [n] in image order, dest int16-as-uint16 [n] (image position -> index of `out`), out fp16
    [rows, >= n].  clear: an fp32 tensor the launch zeroes (the accumulator of the decode_bigp_v_gemm that follows)."""
    U0 = entries[0][0]
    _need_gpu(entries[0][1])
    arr = (BigpUOp * len(entries))()
    keep = []
    for i, (U, y, bias_img, post_img, dest, out) in enumerate(entries):
        assert U.bigp_fold_ok and U.p == U0.p and y.dtype == torch.float16 and y.shape == (rows, U.n) and y.is_contiguous()
        assert out.dtype == torch.float16 and out.stride(1) == 1 and out.shape[0] == rows and dest.dtype == torch.int16 and dest.numel() == U.n
        assert bias_img is None or (bias_img.dtype == torch.float16 and bias_img.numel() == U.n)
        assert post_img is None or (post_img.dtype == torch.float32 and post_img.numel() == U.n)
        F0, M1 = U.bigp_frags(True)
        keep.append((F0, M1))
        arr[i] = BigpUOp(_p(F0), _p(M1), _p(y), _p(bias_img), _p(post_img), _p(dest), _p(out), out.stride(0))
    if clear is not None:
        assert clear.dtype == torch.float32 and clear.is_contiguous() and clear.numel() % 4 == 0
    _lib.call("quipamd_decode_bigp_u", arr, len(entries), U0.p, rows, _p(clear), 0 if clear is None else clear.numel(), _stream())


def decode_bigp_v_gemm(V, gate, up, qweight_d, scale, y, row_tiles_per_wave=0, bits=2, partials=None, xt=None):
    """y += What V(silu(gate) * up) for a 2-bit qfn-b layer whose activation-side operator V is p x 16 (quipamd_decode_bigp_v_gemm):
    gate / up fp16 [rows, n] as the transposed image of V's input (up None: the input is `gate` itself), qweight_d the codes with their
    columns in image order of V (QuantLinear.decode_qweight()), y fp32 [rows, m] ACCUMULATED with atomics (zero it first).
    partials fp32 [p / 16, rows, m]: the K-slices meet in a fixed order instead (stores + a second small launch) -- y is STORED, runs are
    bit-identical (quant.DETERMINISTIC_SPLITK).  xt fp16 [rows, n] scratch: the two-launch form -- the operator pass alone, then the
    dequant-GEMM on x~ -- y STORED, deterministic; what quant.fused_bigp_tail uses from 5 rows on (the one-launch kernel repeats the whole
    activation-side pass in every workgroup: 49 us at 16 rows against ~15)."""
    _need_gpu(gate)
    rows, m = y.shape
    assert V.bigp_fold_ok and gate.dtype == torch.float16 and gate.shape == (rows, V.n) and gate.stride(1) == 1
    assert up is None or (up.dtype == torch.float16 and up.shape == gate.shape and up.stride() == gate.stride())
    assert y.dtype == torch.float32 and y.is_contiguous() and scale.dtype == torch.float32 and scale.numel() == 1
    F0, M1 = V.bigp_frags(False)
    if partials is not None:
        assert partials.dtype == torch.float32 and partials.is_contiguous() and partials.numel() >= (V.p // 16) * rows * m
    if xt is not None:
        assert xt.dtype == torch.float16 and xt.is_contiguous() and xt.shape == (rows, V.n)
    a = BigpVGemmArgs(_p(F0), _p(M1), _p(gate), _p(up), gate.stride(0), _p(qweight_d), _p(scale), int(bits), _p(y), m, V.p, rows, int(row_tiles_per_wave),
                      _p(partials), _p(xt))
    _lib.call("quipamd_decode_bigp_v_gemm", ctypes.byref(a), _stream())


def pair_tables(Uop, Vop, bias16, colscale):
    """per-lane tables of a layer PAIR for the n = 8192 fused launch (include/quip_amd.h pair_sig / pair_bias / pair_cs): for the element
    (a, b) of U^T's image that lane (w, lane) holds in register 4 i + reg after stage 2 -- where it goes in V's input image, and the bias
    and 1 / s that go with it.  Uop: the previous layer's OrthoOp (applied transposed), Vop: the consumer's (forward)."""
    p, q, dev = Uop.p, Uop.q, Uop.device
    assert (Vop.p, Vop.q) == (p, q) and bias16.dtype == torch.float16
    w = torch.arange(16, device=dev).view(16, 1, 1, 1)
    lane = torch.arange(64, device=dev).view(1, 64, 1, 1)
    i = torch.arange(2, device=dev).view(1, 1, 2, 1)
    reg = torch.arange(4, device=dev).view(1, 1, 1, 4)
    tile = w + 16 * i
    bt, at = tile % (q // 16), tile // (q // 16)
    a = 16 * at + 4 * (lane // 16) + reg
    b = 16 * bt + lane % 16
    posU = (a * q + b).reshape(-1)                                           # [16 * 64 * 8] in table order
    n = p * q
    ident = torch.arange(n, device=dev)
    pinU = Uop._p_in.to(dev) if Uop.pin is not None else ident               # U^T's store_idx is inv_pin: element k sits at inv_pin[k] -> k = pin[pos]
    k = pinU[posU]
    inv_pin_V = torch.argsort(Vop._p_in.to(dev)) if Vop.pin is not None else ident
    pv = inv_pin_V[k]                                                        # V's load_idx
    sig = ((pv % q) * (p + 8) + pv // q).to(torch.int16).contiguous()
    return sig, bias16.reshape(-1)[k].contiguous(), colscale.reshape(-1)[k].to(torch.float16).contiguous()


def decode_fused_gemm(*, V, colscale, qweight, scale, y, m, bs, x=None, U=None, u_y=None, u_bias=None, u_residual=None, u_relu=False,
                      t_out=None, norm=0, ln_gamma=None, ln_beta=None, ln_eps=0.0, pair=None, bits=2, ops_only=False):
    """one launch of quipamd_decode_fused_gemm.  V / colscale / qweight / scale / y: lists (1..3 groups); V, U: Fop records
    (OrthoOp.fop); 16-bit tensors fp16.  See include/quip_amd.h for the contract.  ops_only: y[i] = x~_i fp16 [bs, n] (qweight / scale
    may be None)."""
    a = FusedGemmArgs()
    a.ops_only = int(bool(ops_only))
    a.act_dtype, a.bits = _DT[torch.float16], int(bits)
    a.has_u = int(U is not None)
    if U is not None:
        a.U = U
        a.u_y, a.u_bias = _ptr(u_y), _ptr(u_bias)
        a.u_residual, a.ld_residual = _ptr(u_residual), 0 if u_residual is None else u_residual.stride(0)
        a.u_relu = int(bool(u_relu))
        a.t_out, a.ld_t = _ptr(t_out), 0 if t_out is None else t_out.stride(0)
        assert u_y.dtype in (torch.float16, torch.float32) and u_y.is_contiguous() and u_bias.dtype == torch.float16 and u_bias.is_contiguous()
        a.u_y_dtype = _DT[u_y.dtype]
        assert u_residual is None or (u_residual.dtype == torch.float16 and u_residual.stride(1) == 1)
        assert t_out is None or (t_out.dtype == torch.float16 and t_out.stride(1) == 1)
    else:
        assert x.dtype == torch.float16 and x.stride(1) == 1
        a.x, a.ldx = _ptr(x), x.stride(0)
    a.norm, a.ln_eps = int(norm), float(ln_eps)
    if norm:
        assert ln_gamma.dtype == torch.float16 and (norm != 1 or ln_beta.dtype == torch.float16)
        a.ln_gamma, a.ln_beta = _ptr(ln_gamma), _ptr(ln_beta)
    n = len(V)
    a.ngroups = n
    for i in range(n):
        a.V[i] = V[i]
        a.colscale[i] = _f32ptr(colscale[i], "colscale")
        a.qweight[i] = None if (ops_only and qweight is None) else qweight[i].data_ptr()
        a.scale[i] = None if (ops_only and scale is None) else _f32ptr(scale[i], "scale")
        assert y[i].dtype == y[0].dtype and y[i].dtype in (torch.float32, torch.float16) and y[i].is_contiguous()
        a.y[i] = y[i].data_ptr()
    a.y_dtype = _DT[y[0].dtype]
    a.bs, a.m = int(bs), int(m)
    if pair is not None:
        a.pair_sig, a.pair_bias, a.pair_cs = (t.data_ptr() for t in pair)
    _lib.call("quipamd_decode_fused_gemm", ctypes.byref(a), _stream())


def decode_prefetch_next(tensors):
    """attach the byte ranges of `tensors` (device tensors, contiguous) to the NEXT fused decode launch of this thread: 8 spare workgroups of
    that launch pull them into every XCD's L2 (quipamd_decode_prefetch_next, csrc/prefetch.h).  A hint: never changes a result."""
    ts = [t for t in tensors if t is not None and t.numel()]
    n = len(ts)
    if n == 0:
        return
    assert n <= 40, "decode_prefetch_next: at most 40 ranges"
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    sizes = (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t in ts])
    _lib.call("quipamd_decode_prefetch_next", ptrs, sizes, n)


def ortho_blocked_multi(entries, out_dtype):
    """1..3 blocked operators of one shape in the SAME two launches (quipamd_ortho_blocked_rows_multi: q / k / v, gate / up).
    entries = [(OrthoOp, x, kwargs of OrthoOp.blk_desc)]; returns the outputs [rows, n] in out_dtype."""
    op0, x0, _ = entries[0]
    rows, n = x0.shape
    outs = [torch.empty((rows, n), dtype=out_dtype, device=x0.device) for _ in entries]
    arr = (BlkOp * len(entries))()
    keep = []
    for i, ((op, x, kw), out) in enumerate(zip(entries, outs)):
        arr[i], k = op.blk_desc(x, out, **kw)
        keep.append(k)
    ws = torch.empty(len(entries) * rows * n, dtype=torch.float32, device=x0.device)
    _lib.call("quipamd_ortho_blocked_rows_multi", arr, len(entries), _p(ws), _stream())
    return outs


BLK_FUSED_N = 2048                 # csrc/ortho_blk.hip's default: one launch per blocked operator up to this n = p q


BLK_FUSED_ROWS = 4                 # ... and up to this many rows


def decode_attention_config(one_group_from=0, three_heads_from=-1):
    """csrc/decode_attn.hip, forms of the fused attention launch by the number of (sequence, head) pairs: `three_heads_from` (default 257, -1 = the
    default, 0 = never): a workgroup serves three heads; `one_group_from` (default 0 = never): the 4-wave form (measured slower).  A/B runs, tests."""
    _lib.load().quipamd_decode_attention_config(int(one_group_from), int(three_heads_from))


def ortho_blocked_config(max_fused_n=BLK_FUSED_N, max_fused_rows=BLK_FUSED_ROWS):
    """csrc/ortho_blk.hip: one launch per operator for n = p q up to `max_fused_n` and `max_fused_rows` rows (where the input rows fit LDS),
    two stage launches beyond; 0 / False = always two, True = wherever it fits (A/B runs, tests)"""
    if max_fused_n is True:
        max_fused_n, max_fused_rows = 1 << 20, 64
    _lib.load().quipamd_ortho_blocked_config(int(max_fused_n), int(max_fused_rows))


class BlkOp(ctypes.Structure):
    """mirror of `quipamd_blk_op` (include/quip_amd.h): the blocked butterfly on a handful of rows (csrc/ortho_blk.hip)"""
    _fields_ = [("F_first", ctypes.c_void_p), ("F_second", ctypes.c_void_p), ("first_mixes_a", ctypes.c_int), ("p", ctypes.c_int), ("q", ctypes.c_int),
                ("in_idx", ctypes.c_void_p), ("out_idx", ctypes.c_void_p), ("x", ctypes.c_void_p), ("x_dtype", ctypes.c_int), ("ld_x", ctypes.c_int64),
                ("out", ctypes.c_void_p), ("out_dtype", ctypes.c_int), ("ld_out", ctypes.c_int64), ("rows", ctypes.c_int64), ("gate_up", ctypes.c_void_p),
                ("norm", ctypes.c_int), ("ln_gamma", ctypes.c_void_p), ("ln_beta", ctypes.c_void_p), ("ln_eps", ctypes.c_float),
                ("colscale", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("residual_dtype", ctypes.c_int),
                ("ld_residual", ctypes.c_int64), ("relu", ctypes.c_int)]


BLK_MAX_ROWS = 64      # csrc/ortho_blk.hip: 16 rows per workgroup (the MFMA's columns), groups of 16 side by side in the same launch


def _mfma_b_frags(M):
    """M [C, P, P] (out index i, in index k) -> float [C, NT, NT, 64, 4] in v_mfma_f32_16x16x4_f32 B-fragment order
    (include/quip_amd.h): element [c][nt][S][lane][s] = M[c][16 nt + (lane & 15)][16 S + 4 (lane >> 4) + s]."""
    C, P, _ = M.shape
    NT = (P + 15) // 16
    Mp = torch.zeros((C, 16 * NT, 16 * NT), dtype=torch.float32, device=M.device)
    Mp[:, :P, :P] = M
    F = Mp.view(C, NT, 16, NT, 4, 4)               # [c, nt, j, S, g, s]
    return F.permute(0, 1, 3, 4, 2, 5).contiguous().reshape(-1)   # [c, nt, S, g, j, s] : lane = 16 g + j


class OrthoOp:
    """Device-resident structured orthogonal operator built from the reference's (B, p_in, p_out) tuple
    (method.py:34-43).  apply_rows(x) computes Q x_r for every row x_r; transpose=True applies Q^T."""

    def __init__(self, Bpp, device):
        (B, p_in, p_out) = Bpp
        B0, B1 = B[0].to(torch.float32), B[1].to(torch.float32)
        B0 = B0.reshape(-1, B0.shape[-1], B0.shape[-1]).to(device)
        B1 = B1.reshape(-1, B1.shape[-1], B1.shape[-1]).to(device)
        self.p, self.q = B0.shape[-1], B1.shape[-1]
        self.n = self.p * self.q
        self.blocked = int(B0.shape[0] > 1 or B1.shape[0] > 1)
        if self.blocked:
            assert B0.shape[0] == self.q and B1.shape[0] == self.p
        self._B0, self._B1 = B0, B1
        self._frags = {}
        p_in = torch.as_tensor(p_in).to(torch.int64).cpu()
        p_out = torch.as_tensor(p_out).to(torch.int64).cpu()
        self._p_in, self._p_out = p_in, p_out
        ident = torch.arange(self.n)
        i32 = lambda t: t.to(torch.int32).to(device)
        self.pin = None if torch.equal(p_in, ident) else i32(p_in)
        self.inv_pin = None if torch.equal(p_in, ident) else i32(torch.argsort(p_in))
        self.pout = None if torch.equal(p_out, ident) else i32(p_out)
        self.inv_pout = None if torch.equal(p_out, ident) else i32(torch.argsort(p_out))
        self.device = device
        # single-launch small-batch path (quipamd_ortho_apply_small): Kronecker factors that fit one workgroup's LDS
        lds = (self.p * (self.p + 4) + self.q * (self.q + 4) + 2 * self.p * (self.q + 4) + 16) * 4
        self.split_ok = False
        self.small_ok = (not self.blocked) and self.p % 16 == 0 and self.q % 16 == 0 and lds <= 160 * 1024 and self.n <= 16384 and (self.q & (self.q - 1)) == 0
        # p x 16 with a large p (Llama's 11008 = 688 x 16): csrc/ortho_bigp.hip for a handful of rows
        self.bigp_ok = (not self.blocked) and self.q == 16 and self.p % 16 == 0 and 64 <= self.p and self.n <= 12288
        if self.small_ok or self.bigp_ok:
            self._M = {False: (B0[0].contiguous(), B1[0].contiguous()),
                       True: (B0[0].t().contiguous(), B1[0].t().contiguous())}
        if self.small_ok:
            # split-bf16 copies of the factors (hi + lo) for the bf16-pipe variant of the small-batch kernel
            split_lds = 2 * (2 * (self.p * (self.p + 8) + self.q * (self.q + 8)) + 2 * self.q * (self.p + 8) + 2 * self.p * (self.q + 8)) + 64
            self.split_ok = self.p % 32 == 0 and self.q % 32 == 0 and 2 * self.q >= self.p and split_lds <= 160 * 1024
            if self.split_ok:
                def hl(M):
                    hi = M.to(torch.bfloat16)
                    return hi.contiguous(), (M - hi.float()).to(torch.bfloat16).contiguous()
                self._Msplit = {k: hl(m0) + hl(m1) for k, (m0, m1) in self._M.items()}
        # csrc/ortho_blk.hip: the BLOCKED butterfly on a handful of rows (decode of a model quantised by the shipped --incoh_processing)
        self.blk_ok = bool(self.blocked and self.p % 16 == 0 and self.q % 16 == 0 and self.p <= 768 and self.q <= 768)
        # csrc/ortho_tile.hip: one workgroup per 16 x 16 output tile for a handful of rows (decode)
        self.tile_ok = self.split_ok and (self.p, self.q) in ((64, 32), (64, 64), (128, 64))
        self.tile_supported = self.split_ok and (self.p, self.q) in ((64, 32), (64, 64), (128, 64))

    def zero_bias(self):
        """n float32 zeros: what a bias-free layer (Llama) hands the output-side kernels as `bias`"""
        if getattr(self, '_zero_bias', None) is None:
            self._zero_bias = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        return self._zero_bias

    def one_scale(self):
        if getattr(self, '_one_scale', None) is None:
            self._one_scale = torch.ones(self.n, dtype=torch.float32, device=self.device)
        return self._one_scale

    def fop(self, transpose):
        """the operator prepared for csrc/decode_fused.hip (quipamd_fop): fp16 factors in MFMA B-fragment order + uint16 index
        vectors; built once per orientation and kept on the device"""
        cache = self.__dict__.setdefault('_fops', {})
        key = bool(transpose)
        if key not in cache:
            assert self.fused_ok
            M0, M1 = self._M[key]
            ident = torch.arange(self.n, device=self.device, dtype=torch.int32)
            ld, st = (self.pout, self.inv_pin) if key else (self.inv_pin, self.pout)
            i16 = lambda t: (ident if t is None else t).to(torch.int16).contiguous()       # n <= 8192: the bits of a uint16
            keep = (_f16_b_frags(M0), _f16_b_frags(M1), i16(ld), i16(st))
            cache[key] = (Fop(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), self.p, self.q), keep)
        return cache[key][0]

    def zt_rows(self):
        """int64 [n] on the device: zt[i] = position of element i of a vector that U^T (this operator, transposed) is applied to in
        "ZT order" -- the transposed image row-major, b * p + a for load position pout[i] = a * q + b (include/quip_amd.h)"""
        if getattr(self, '_zt_rows', None) is None:
            pos = self._p_out.to(self.device)
            self._zt_rows = (pos % self.q) * self.p + pos // self.q
        return self._zt_rows

    def image_cols(self):
        """int64 [n] on the device: img[k] = image position the (forward) operator's output element k is read from, pout[k]"""
        if getattr(self, '_img_cols', None) is None:
            self._img_cols = self._p_out.to(self.device).clone()
        return self._img_cols

    @property
    def fused_ok(self):
        return (not self.blocked) and (self.p, self.q) in FUSED_SHAPES and (self.small_ok or self.bigp_ok)

    @property
    def bigp_fold_ok(self):
        """p x 16 with a large p (Llama's 688 x 16): served by csrc/decode_bigp.hip"""
        return bool(self.bigp_ok and (self.p, self.q) not in FUSED_SHAPES and self.p <= 1024)      # = quipamd_decode_bigp_supported

    @property
    def fold_ok(self):
        """does a decode launch exist that takes this operator's permutations folded into the packing (QuantLinear.decode_qweight)?"""
        return self.fused_ok or self.bigp_fold_ok

    def bigp_frags(self, transpose):
        """(F0, M1) for csrc/decode_bigp.hip: M0 of the wanted orientation as zero-padded fp16 B fragments, M1 fp32 [16, 16]"""
        cache = self.__dict__.setdefault('_bigp_frags', {})
        key = bool(transpose)
        if key not in cache:
            M0, M1 = self._M[key]
            cache[key] = (_f16_b_frags_padded(M0), M1.to(torch.float32).contiguous())
        return cache[key]

    def blk_factors(self, transpose):
        """(F_first, F_second, first_mixes_a) for csrc/ortho_blk.hip: fp16 [G][P][P] (out index, in index) of the stage that runs first /
        second.  Q: B0 [q, p, p] (mixes a) then B1 [p, q, q]; Q^T: B1^T then B0^T.  Built once per orientation, kept on the device."""
        cache = self.__dict__.setdefault('_blk_factors', {})
        key = bool(transpose)
        if key not in cache:
            assert self.blk_ok
            h = lambda t: t.to(torch.float16).contiguous()
            if key:
                cache[key] = (h(self._B1.transpose(1, 2)), h(self._B0.transpose(1, 2)), 0)
            else:
                cache[key] = (h(self._B0), h(self._B1), 1)
        return cache[key]

    def blk_desc(self, x, out, transpose=False, colscale=None, bias=None, ln=None, residual=None, relu=False, gate_up=None):
        """(BlkOp record, tensors it points at) for apply_rows_blocked / ops.ortho_blocked_multi"""
        rows = x.shape[0]
        assert self.blk_ok and x.dim() == 2 and x.shape[1] == self.n and x.stride(1) == 1 and rows <= BLK_MAX_ROWS and x.dtype in _DT
        F1, F2, first_a = self.blk_factors(transpose)
        gather, scatter = (self.inv_pout, self.pin) if transpose else (self.pin, self.inv_pout)
        a = BlkOp()
        a.F_first, a.F_second, a.first_mixes_a, a.p, a.q = F1.data_ptr(), F2.data_ptr(), first_a, self.p, self.q
        a.in_idx, a.out_idx = _ptr(gather), _ptr(scatter)
        a.x, a.x_dtype, a.ld_x = x.data_ptr(), _dtype(x), x.stride(0)
        a.out, a.out_dtype, a.ld_out, a.rows = out.data_ptr(), _dtype(out), out.stride(0), rows
        if gate_up is not None:
            assert gate_up.dtype == x.dtype and gate_up.shape == x.shape and gate_up.stride() == x.stride()
            a.gate_up = gate_up.data_ptr()
        if ln is not None:
            g, b, eps = ln
            assert g.dtype == torch.float16 and (b is None or b.dtype == torch.float16) and g.numel() == self.n
            a.norm, a.ln_gamma, a.ln_beta, a.ln_eps = (1 if b is not None else 2), g.data_ptr(), _ptr(b), float(eps)
        cs, bs_ = _f32vec(colscale, x.device), _f32vec(bias, x.device)
        a.colscale, a.bias = _ptr(cs), _ptr(bs_)
        if residual is not None:
            assert residual.shape == (rows, self.n) and residual.stride(1) == 1 and residual.dtype in _DT
            a.residual, a.residual_dtype, a.ld_residual = residual.data_ptr(), _dtype(residual), residual.stride(0)
        a.relu = int(bool(relu))
        return a, (cs, bs_, F1, F2)

    def apply_rows_blocked(self, x, transpose=False, colscale=None, out_dtype=None, bias=None, ln=None, residual=None, relu=False, gate_up=None):
        """out = [relu]( Q ( colscale * Norm( silu(x) * gate_up | x ) ) + bias + residual ) for <= 8 rows of a BLOCKED operator, two launches
        (quipamd_ortho_blocked_rows; fp16 factors: the fused decode launches' tolerance class).  ln = (gamma, beta | None, eps) with fp16
        gamma / beta; colscale / bias fp32 [n]; residual [rows, n] fp16 / bf16 / fp32; gate_up like x."""
        _need_gpu(x)
        return ortho_blocked_multi([(self, x, dict(transpose=transpose, colscale=colscale, bias=bias, ln=ln, residual=residual, relu=relu,
                                                   gate_up=gate_up))], out_dtype or x.dtype)[0]

    def store_inv(self, transpose):
        """image position -> output index: the inverse of the `store_idx` small_op() hands to the kernels"""
        return self.pin if transpose else self.inv_pout

    def state(self):
        """the reference-style generator tuple ([B0, B1], p_in, p_out) on the CPU -- what a packed checkpoint stores."""
        B0 = self._B0.cpu() if self.blocked else self._B0[0].cpu()
        B1 = self._B1.cpu() if self.blocked else self._B1[0].cpu()
        return ([B0, B1], self._p_in.clone(), self._p_out.clone())

    def _stage_frags(self, transpose):
        """(first, second) stage matrices in B-fragment order; built once per orientation, kept on the device."""
        if transpose not in self._frags:
            if transpose:
                self._frags[True] = (_mfma_b_frags(self._B1.transpose(1, 2)), _mfma_b_frags(self._B0.transpose(1, 2)))
            else:
                self._frags[False] = (_mfma_b_frags(self._B0), _mfma_b_frags(self._B1))
        return self._frags[transpose]

    SMALL_ROWS = 64
    use_split = True      # small-batch path: split-bf16 factors on the bf16 matrix pipe (~1e-5 rel.) when the shape allows

    def apply_rows(self, x, transpose=False, colscale=None, out_dtype=None, bias=None, fast16=False):
        """out[r] = Q x[r] (Q^T if transpose) with x[r] multiplied elementwise by colscale first and bias added last.
        fast16: the caller is a decode step (a handful of activation rows): a BLOCKED operator may then take csrc/ortho_blk.hip (fp16
        factors, ~3e-4 per stage) instead of the general fp32 launches; quantisation (W, H) never passes it."""
        _need_gpu(x)
        assert x.dim() == 2 and x.shape[1] == self.n and x.stride(1) == 1
        rows = x.shape[0]
        if fast16 and self.blk_ok and 0 < rows <= BLK_MAX_ROWS and x.dtype in _DT:
            return self.apply_rows_blocked(x, transpose=transpose, colscale=colscale, out_dtype=out_dtype, bias=bias)
        out = torch.empty((rows, self.n), dtype=out_dtype or x.dtype, device=x.device)
        cs = _f32vec(colscale, x.device)
        if self.small_ok and x.stride(0) % 4 == 0:
            # Kronecker operator: the single-launch kernel for any number of rows (its workgroups walk the rows with the
            # factors resident in LDS).  Few rows (decode) and 16-bit activations (prefill) take the split-bf16 arithmetic
            # (~1e-5); many fp32 rows are the weight side (W, H in preproc / postproc) and stay on the exact fp32 MFMA.
            split = self.use_split
            if rows > self.SMALL_ROWS and x.dtype == torch.float32:
                self.use_split = False
            try:
                ortho_apply_ops([(self, self.small_op(x, out, transpose=transpose, colscale=cs, bias=_f32vec(bias, x.device)), bool(transpose))], rows)
            finally:
                self.use_split = split
            return out
        if self.bigp_ok and rows <= BIGP_ROWS and x.stride(0) % 4 == 0 and x.dtype in (torch.float16, torch.float32) \
                and self.pin is not None and self.pout is not None:
            if x.dtype == torch.float16:            # activation-side operand set: (x f16, colscale); a bias is added afterwards
                ones = cs if cs is not None else self.one_scale()
                ortho_bigp_ops([self.small_op(x, out, transpose=transpose, colscale=ones)], [self.store_inv(transpose)], rows)
                if bias is not None:
                    out += _f32vec(bias, x.device).to(out.dtype)
                return out
            if cs is None:                           # output-side operand set: (x f32, bias)
                b = _f32vec(bias, x.device) if bias is not None else self.zero_bias()
                ortho_bigp_ops([self.small_op(x, out, transpose=transpose, bias=b)], [self.store_inv(transpose)], rows)
                return out
        ws = torch.empty((16 * ((rows + 15) // 16), self.n), dtype=torch.float32, device=x.device)
        f1, f2 = self._stage_frags(bool(transpose))
        gather, scatter = (self.inv_pout, self.pin) if transpose else (self.pin, self.inv_pout)
        _lib.call("quipamd_ortho_apply_rows", _p(f1), _p(f2), self.blocked, _p(gather), _p(scatter),
                  self.p, self.q, int(bool(transpose)), _p(cs), _p(x), _dtype(x), x.stride(0), _p(out), _dtype(out),
                  out.stride(0), rows, _p(ws), _stream())
        if bias is not None:
            out += _f32vec(bias, x.device).to(out.dtype)
        return out

    def small_op(self, x, out, transpose=False, colscale=None, bias=None, ln=None, residual=None, relu=False,
                 out_dtype=None, ld=None):
        """descriptor of  out = [relu](Q (colscale * [LayerNorm](x)) + bias + residual)  for ortho_small_ops().
        ln = (gamma, beta, eps) tensors on the device; every tensor argument must outlive the launch call.
        For ortho_small_chain: x may be None (a second op) and out may be None with out_dtype / ld given (a first op whose
        result is only handed over)."""
        assert (self.small_ok or self.bigp_ok) and (x is None or x.stride(1) == 1) and (out is None or out.stride(1) == 1)
        _f32ptr(colscale, "colscale"), _f32ptr(bias, "bias")          # read as float* by the kernel
        if x is None or out is None:
            n = self.p * self.q
            M0, M1 = self._M[bool(transpose)]
            ldv, st = (self.pout, self.inv_pin) if transpose else (self.inv_pin, self.pout)
            g, b, eps = ln if ln is not None else (None, None, 0.0)
            sp = self._Msplit[bool(transpose)] if (self.split_ok and self.use_split) else (None, None, None, None)
            odt = _dtype(out) if out is not None else _DT[out_dtype]
            return SmallOp(_ptr(M0), _ptr(M1), _ptr(ldv), _ptr(st), self.p, self.q, int(bool(transpose)),
                           _ptr(colscale), _ptr(bias), _ptr(g), _ptr(b), float(eps), 0 if g is None else _dtype(g),
                           _ptr(residual), 0 if residual is None else _dtype(residual), int(bool(relu)),
                           _ptr(x), 0 if x is None else _dtype(x), n if x is None else x.stride(0),
                           _ptr(out), odt, (ld or n) if out is None else out.stride(0),
                           _ptr(sp[0]), _ptr(sp[1]), _ptr(sp[2]), _ptr(sp[3]))
        M0, M1 = self._M[bool(transpose)]
        ld, st = (self.pout, self.inv_pin) if transpose else (self.inv_pin, self.pout)
        g, b, eps = ln if ln is not None else (None, None, 0.0)
        sp = self._Msplit[bool(transpose)] if (self.split_ok and self.use_split) else (None, None, None, None)
        return SmallOp(_ptr(M0), _ptr(M1), _ptr(ld), _ptr(st), self.p, self.q, int(bool(transpose)),
                       _ptr(colscale), _ptr(bias), _ptr(g), _ptr(b), float(eps), 0 if g is None else _dtype(g),
                       _ptr(residual), 0 if residual is None else _dtype(residual), int(bool(relu)),
                       _ptr(x), _dtype(x), x.stride(0), _ptr(out), _dtype(out), out.stride(0),
                       _ptr(sp[0]), _ptr(sp[1]), _ptr(sp[2]), _ptr(sp[3]))

    def apply_cols(self, x, transpose=False):
        """Q @ x for x [n, c] (the reference's mul_ortho_butterfly orientation)."""
        return self.apply_rows(x.t().contiguous(), transpose=transpose).t().contiguous()


# ------------------------------------------------------------------------------------------------- K4
def unit_lower_t(C):
    """LT[c][j] = C[j][c]/C[c][c] (j>c) from the lower Cholesky factor C (vector_balance.py:171-173)."""
    _need_gpu(C)
    assert C.dtype == torch.float32 and C.dim() == 2 and C.shape[0] == C.shape[1]
    C = C.contiguous()
    LT = torch.empty_like(C)
    _lib.call("quipamd_unit_lower_t", _p(C), _p(LT), C.shape[0], _stream())
    return LT


def ldlq_greedy_pass(wr, sH, negH_upper, hdiag):
    """one greedy pass (vector_balance.py:186-196): returns (wr_new float [m,d] unclamped, eps float [m,d])."""
    _need_gpu(wr, sH, negH_upper, hdiag)
    m, d = wr.shape
    assert wr.dtype == sH.dtype == negH_upper.dtype == hdiag.dtype == torch.float32
    assert sH.shape == (m, d) and negH_upper.shape == (d, d) and hdiag.shape == (d,)
    out = torch.empty_like(wr)
    eps = torch.empty_like(wr)
    _lib.call("quipamd_ldlq_greedy_pass", _p(wr.contiguous()), _p(sH.contiguous()), _p(negH_upper.contiguous()),
              _p(hdiag.contiguous()), _p(out), _p(eps), m, d, _stream())
    return out, eps


def gptq_feedback_matrix(Hinv):
    """FT for gptq_round from the upper Cholesky factor of H^-1 (gptq.py:51-54): reversed, transposed, each row of Hinv
    divided by its diagonal, negated, strictly upper (include/quip_amd.h)."""
    Hn = Hinv / Hinv.diagonal()[:, None]
    return torch.triu((-Hn.t()).flip(0, 1), diagonal=1).contiguous()


def gptq_feedback(H, check=True):
    """FT of the OPTQ sweep from the (damped) Hessian itself, on our own kernels: flip, K8 Cholesky, one unit-triangular
    inverse (csrc/trinv.hip) -- stands in for gptq.py:51-54's cholesky / cholesky_inverse / cholesky and
    gptq_feedback_matrix.  Raises torch.linalg.LinAlgError like torch.linalg.cholesky when H is not positive definite."""
    _need_gpu(H)
    assert H.dtype == torch.float32 and H.dim() == 2 and H.shape[0] == H.shape[1]
    d = H.shape[0]
    H = H.contiguous()
    FT = torch.empty_like(H)
    work = torch.empty((2, d, d), dtype=torch.float32, device=H.device)
    info = torch.zeros(1, dtype=torch.int32, device=H.device)
    _lib.call("quipamd_gptq_feedback", _p(H), _p(FT), _p(work), d, _p(info), _stream())
    if check and int(info.item()):
        raise torch.linalg.LinAlgError("quip_amd.gptq_feedback: the Hessian is not positive-definite")
    return FT


def unit_upper_inverse(N):
    """(I + triu(N, 1))^-1, upper triangular (csrc/trinv.hip)."""
    _need_gpu(N)
    assert N.dtype == torch.float32 and N.dim() == 2 and N.shape[0] == N.shape[1]
    N = N.contiguous()
    X = torch.zeros_like(N)
    work = torch.empty_like(N)
    _lib.call("quipamd_unit_upper_inverse", _p(N), _p(X), _p(work), N.shape[0], _stream())
    return X


def gptq_round(Wgrid, Hinv, bits, return_err=False, FT=None):
    """OPTQ codes uint8 [m,d] of grid coordinates Wgrid given Hinv = chol(H^-1, upper) (gptq.py:56-93, groupsize -1) or the
    prepared feedback matrix FT (gptq_feedback)."""
    _need_gpu(Wgrid, Hinv, FT)
    assert Wgrid.dtype == torch.float32
    m, d = Wgrid.shape
    if FT is None:
        assert Hinv.dtype == torch.float32 and Hinv.shape == (d, d)
        FT = gptq_feedback_matrix(Hinv)
    assert FT.shape == (d, d) and FT.dtype == torch.float32 and FT.is_contiguous()
    wrev = Wgrid.flip(1).contiguous()
    codes = torch.empty((m, d), dtype=torch.uint8, device=Wgrid.device)
    err = torch.empty((m, d), dtype=torch.float32, device=Wgrid.device)
    _lib.call("quipamd_gptq_round", _p(wrev), _p(FT), bits, _p(codes), _p(err), m, d, _stream())
    codes = codes.flip(1).contiguous()
    return (codes, err.flip(1).contiguous()) if return_err else codes


def gptq_round_groups(W, Hinv, bits, groupsize=-1, sym=False, qfn='a', scale=None, zero=None, return_codes=False, FT=None):
    """OPTQ in weight units with the reference quantiser in the loop (gptq.py:60-87 + quant.py:6-21): `groupsize` 16/32/64/128
    (the kernel finds each group's (scale, zero) like Quantizer.find_params_qfna, perchannel) or -1 with per-row scale/zero
    given; qfn 'a' or 'c'.  Returns (Q float32 [m,d], scale, zero[, codes]); with groups scale/zero are [m, d/groupsize]."""
    _need_gpu(W, Hinv, FT)
    assert W.dtype == torch.float32 and qfn in ('a', 'c')
    m, d = W.shape
    dev = W.device
    if groupsize > 0:
        if groupsize not in (16, 32, 64, 128) or d % groupsize:
            raise ValueError(f"gptq_round_groups: groupsize {groupsize} must be 16, 32, 64 or 128 and divide {d}")
        scale = torch.empty((m, d // groupsize), dtype=torch.float32, device=dev)
        zero = torch.empty_like(scale)
    else:
        scale, zero = _grid(scale, zero, m, 'a', dev)
        scale, zero = scale.clone(), zero.clone()
    if FT is None:
        assert Hinv.dtype == torch.float32 and Hinv.shape == (d, d)
        FT = gptq_feedback_matrix(Hinv)
    assert FT.shape == (d, d) and FT.dtype == torch.float32 and FT.is_contiguous()
    wrev = W.flip(1).contiguous()
    Q = torch.empty((m, d), dtype=torch.float32, device=dev)
    codes = torch.empty((m, d), dtype=torch.uint8, device=dev) if return_codes else None
    err = torch.empty((m, d), dtype=torch.float32, device=dev)
    _lib.call("quipamd_gptq_round_groups", _p(wrev), _p(FT), bits, int(groupsize), int(bool(sym)), int(qfn == 'c'), _p(scale), _p(zero),
              _p(Q), _p(codes), _p(err), m, d, _stream())
    out = (Q.flip(1).contiguous(), scale, zero)
    return out + (codes.flip(1).contiguous(),) if return_codes else out


def preproc_rescale(w, H):
    """the diagonal rescale of QuantMethod.preproc (method.py:140-156) in three launches (csrc/preproc.hip):
    H /= max|H|;  s = clamp((clamp(diag H) / clamp(diag W^T W)) ** 0.25);  W <- W s (columns, rounded to W's dtype);  H <- H / s_j / s_i.
    w [m, d] (fp16 / bf16 / fp32) and H [d, d] fp32 are rewritten IN PLACE; returns s (fp32 [d])."""
    _need_gpu(w, H)
    m, d = w.shape
    assert H.dtype == torch.float32 and H.shape == (d, d) and H.is_contiguous() and w.is_contiguous()
    s = torch.empty(d, dtype=torch.float32, device=w.device)
    ws = torch.empty(int(_lib.load().quipamd_preproc_workspace_bytes(m, d)), dtype=torch.uint8, device=w.device)
    _lib.call("quipamd_preproc_rescale", _p(H), _p(w), _dtype(w), m, d, _p(s), _p(ws), _stream())
    return s


def preproc_trace_ridge(H, ridge):
    """H <- H * (n / (trace(H) + 1e-8)) + ridge * I in place (method.py:165), two launches"""
    _need_gpu(H)
    assert H.dtype == torch.float32 and H.dim() == 2 and H.shape[0] == H.shape[1] and H.is_contiguous()
    ws = torch.empty(64, dtype=torch.uint8, device=H.device)
    _lib.call("quipamd_preproc_trace_ridge", _p(H), H.shape[0], float(ridge), _p(ws), _stream())
    return H


def gptq_round_qfnb(W, FT, bits):
    """OPTQ with the qfn-b quantiser in the loop (gptq.py:56-93 + quant.py:10-15,158-160: every column on its own scale, recomputed from all
    rows of the updated column) as csrc/gptq_qfnb.hip runs it.  W float32 [m, d], FT = gptq_feedback(H).  Returns (Q float32 [m, d], the
    per-column scales float32 [d])."""
    _need_gpu(W, FT)
    assert W.dtype == torch.float32 and W.dim() == 2
    m, d = W.shape
    assert FT.shape == (d, d) and FT.dtype == torch.float32 and FT.is_contiguous()
    wt = W.flip(1).t().contiguous()                                   # [d, m], columns reversed
    qt = torch.empty_like(wt)
    cs = torch.empty(d, dtype=torch.float32, device=W.device)
    lib = _lib.load()
    ws = torch.empty(int(lib.quipamd_gptq_qfnb_workspace_bytes(m, d)), dtype=torch.uint8, device=W.device)
    _lib.call("quipamd_gptq_round_qfnb", _p(wt), _p(FT), int(bits), _p(qt), _p(cs), _p(ws), m, d, _stream())
    # the sweep's workgroups wait for each other behind a BOUNDED poll (include/quip_amd.h): the abort word is read once the stream has
    # drained -- the callers synchronise right behind this call anyway (gptq.py: torch.cuda.synchronize() closes fasterquant's timer)
    off = int(lib.quipamd_gptq_qfnb_info_offset(m, d))
    code = int(ws[off:off + 4].view(torch.int32).item()) if m and d else 0
    if code == 2:
        # the one-XCD form's roll call failed before anything was written (fewer workgroups than needed became resident on that XCD):
        # the same sweep once more with the exchange across the XCDs
        short, spin, rows = _GB_DEBUG
        lib.quipamd_gptq_qfnb_debug(short, spin, 1)
        try:
            _lib.call("quipamd_gptq_round_qfnb", _p(wt), _p(FT), int(bits), _p(qt), _p(cs), _p(ws), m, d, _stream())
            code = int(ws[off:off + 4].view(torch.int32).item())
        finally:
            lib.quipamd_gptq_qfnb_debug(short, spin, rows)
    if code != 0:
        raise _lib.QuipAmdError("quipamd_gptq_round_qfnb: the sweep was abandoned -- its workgroups were not all co-resident "
                                "(is another grid / an RCCL collective holding compute units of this device?)")
    return qt.t().flip(1).contiguous(), cs.flip(0).contiguous()


_GB_DEBUG = [0, 0, 0]


def gptq_qfnb_debug(short_grid=0, spin_limit=0, force_rows=0):
    """test / lab hook of csrc/gptq_qfnb.hip (quipamd_gptq_qfnb_debug): launch `short_grid` workgroups too few, give up after `spin_limit`
    polls, `force_rows` rows per workgroup"""
    _GB_DEBUG[:] = [int(short_grid), int(spin_limit), int(force_rows)]
    _lib.load().quipamd_gptq_qfnb_debug(int(short_grid), int(spin_limit), int(force_rows))


def cholesky_lt(H, check=True):
    """LT = D^-1 U strictly upper with H = U^T U: the unit-lower LDL factor of vector_balance.py:171-173, transposed,
    by the blocked fp32 factorisation of quip_amd/csrc/cholesky.hip (K8).  Raises torch.linalg.LinAlgError like
    torch.linalg.cholesky when a pivot is not positive (check=False skips the device read-back)."""
    _need_gpu(H)
    assert H.dtype == torch.float32 and H.dim() == 2 and H.shape[0] == H.shape[1]
    H = H.contiguous()
    LT = torch.empty_like(H)
    info = torch.zeros(1, dtype=torch.int32, device=H.device)
    _lib.call("quipamd_cholesky_lt", _p(H), _p(LT), H.shape[0], _p(info), _stream())
    if check:
        bad = int(info.item())
        if bad:
            raise torch.linalg.LinAlgError(
                f"quip_amd.cholesky_lt: the leading minor of order {bad} is not positive-definite")
    return LT


def cholesky_config(old_syrk=False, lookahead=None, unblocked_diag=False):
    """tests / A-B measurements: K8 with the guarded round-1 trailing-update kernel, the unblocked 64-step factorisation of the diagonal
    block (rounds 1-2) and / or the two-stream look-ahead schedule forced on (True: from d = 1024) or off (False); None = the default
    (from d = 12288).  Process-wide switch."""
    _lib.load().quipamd_cholesky_config(int(bool(old_syrk)) | (2 if unblocked_diag else 0), -1 if lookahead is None else int(bool(lookahead)))


def ldlq_config(row_groups=0):
    """tests / A-B runs: K4 with 1 or 2 groups of 16 rows per workgroup forced (0 = by the row count).  Process-wide switch."""
    _lib.load().quipamd_ldlq_config(int(row_groups))


def ldlq_round(Wgrid, LT, bits, eta=None, return_err=False):
    """LDLQ codes uint8 [m,d] (vector_balance.py:155-199 / 218-257)."""
    _need_gpu(Wgrid, LT)
    assert Wgrid.dtype == torch.float32 and LT.dtype == torch.float32
    Wgrid, LT = Wgrid.contiguous(), LT.contiguous()
    m, d = Wgrid.shape
    assert LT.shape == (d, d)
    if eta is not None:
        eta = eta.to(torch.float32).contiguous()
    codes = torch.empty((m, d), dtype=torch.uint8, device=Wgrid.device)
    err = torch.empty((m, d), dtype=torch.float32, device=Wgrid.device)
    _lib.call("quipamd_ldlq_round", _p(Wgrid), _p(LT), _p(eta), bits, _p(codes), _p(err), m, d, _stream())
    return (codes, err) if return_err else codes


# ------------------------------------------------------------------------------------------------- K7
def hessian_accum(Hacc, x, fast=False):
    """Hacc += x^T x in fp64 for the block-lower triangle (method.py:98-120).  x: [tokens, d] f16/bf16/f32, last dim
    contiguous; Hacc: float64 [d, d] (zero-initialised by the caller, finished by hessian_finish).
    fast=True (f16 / bf16 only): exact products on the 16-bit matrix pipe, fp32 runs of 128 tokens summed in fp64 --
    ~10x faster, error ~1e-9 of sqrt(H_ii H_jj) instead of fp64 round-off (include/quip_amd.h)."""
    _need_gpu(Hacc, x)
    assert Hacc.dtype == torch.float64 and Hacc.dim() == 2 and Hacc.shape[0] == Hacc.shape[1] and Hacc.is_contiguous()
    assert x.dim() == 2 and x.shape[1] == Hacc.shape[0]
    if x.numel() and x.stride(1) != 1:
        x = x.contiguous()
    ldx = x.stride(0) if x.shape[0] > 1 else x.shape[1]
    if ldx < x.shape[1]:                       # expanded / overlapping rows
        x = x.contiguous()
        ldx = x.shape[1]
    if fast and x.dtype in (torch.float16, torch.bfloat16) and x.numel():
        n = _lib.load().quipamd_hessian_fast_workspace(x.shape[0], x.shape[1])
        ws = torch.empty(n, dtype=x.dtype, device=x.device)
        _lib.call("quipamd_hessian_accum_fast", _p(x), _dtype(x), ldx, x.shape[0], x.shape[1], _p(Hacc), _p(ws), _stream())
        return Hacc
    _lib.call("quipamd_hessian_accum", _p(x), _dtype(x), ldx, x.shape[0], x.shape[1], _p(Hacc), _stream())
    return Hacc


def hessian_finish(Hacc, nsamples):
    """fp32 [d,d] = mirror(Hacc) / nsamples  (post_batch, method.py:122-123)."""
    _need_gpu(Hacc)
    assert Hacc.dtype == torch.float64 and Hacc.dim() == 2 and Hacc.shape[0] == Hacc.shape[1] and Hacc.is_contiguous()
    H = torch.empty(Hacc.shape, dtype=torch.float32, device=Hacc.device)
    _lib.call("quipamd_hessian_finish", _p(Hacc), float(nsamples), _p(H), Hacc.shape[0], _stream())
    return H


# ------------------------------------------------------------------------------------------------- decode attention
def rope_inplace(q, k, cos_table, sin_table, pos, heads, kv_heads=None):
    """rotary embedding of one decode step, in place on q [bs, heads*hd] and k [bs, kv_heads*hd] at position pos (int64 [1] on
    the device); cos_table / sin_table float32 [maxpos, hd] in HF's duplicated-halves layout (quipamd_rope_inplace)."""
    _need_gpu(q, k, cos_table, sin_table, pos)
    kv_heads = kv_heads or heads
    bs = q.shape[0]
    hd = q.shape[1] // heads
    assert q.dtype == k.dtype and q.stride(1) == 1 and k.stride(1) == 1 and k.shape[1] == kv_heads * hd
    assert cos_table.dtype == torch.float32 and sin_table.dtype == torch.float32 and cos_table.shape[1] == hd and cos_table.is_contiguous()
    assert pos.dtype == torch.int64 and pos.numel() == 1 and sin_table.shape == cos_table.shape and sin_table.is_contiguous()
    _lib.call("quipamd_rope_inplace", _p(q), _p(k), _p(cos_table), _p(sin_table), cos_table.shape[0], _p(pos), _dtype(q), bs, heads, kv_heads, hd,
              q.stride(0), k.stride(0), _stream())


def decode_attention(q, k, v, kcache, vcache, pos, scale=None):
    """one decode step of causal attention with a static KV cache, one launch (quip_amd/csrc/decode_attn.hip).
    q, k, v: [bs, heads*hd] f16/bf16; kcache, vcache: [bs, heads, maxlen, hd] contiguous (updated in place at *pos);
    pos: int64 [1] ON THE DEVICE.  Returns out [bs, heads*hd]."""
    _need_gpu(q, k, v, kcache, vcache, pos)
    bs, heads, maxlen, hd = kcache.shape
    assert q.shape == (bs, heads * hd) and k.shape == q.shape and v.shape == q.shape
    assert kcache.is_contiguous() and vcache.is_contiguous() and vcache.shape == kcache.shape
    assert pos.dtype == torch.int64 and pos.numel() == 1
    assert q.dtype == k.dtype == v.dtype == kcache.dtype == vcache.dtype
    ld = q.stride(0) if bs > 1 else heads * hd
    assert all(t.stride(-1) == 1 and (bs == 1 or t.stride(0) == ld) for t in (q, k, v))
    out = torch.empty((bs, heads * hd), dtype=q.dtype, device=q.device)
    if bs > 1 and ld != heads * hd:
        out = torch.empty((bs, ld), dtype=q.dtype, device=q.device)[:, :heads * hd]
    sc = float(scale) if scale is not None else 1.0 / (hd ** 0.5)
    _lib.call("quipamd_decode_attention", _p(q), _p(k), _p(v), _p(kcache), _p(vcache), _p(pos), _p(out), _dtype(q), bs,
              heads, hd, maxlen, ctypes.c_float(sc), ld, _stream())
    return out


def decode_attention_fused(Uops, ys, biases, kcache, vcache, pos, cos_table=None, sin_table=None, scale=None):
    """decode attention with the output-side operators of q / k / v in its prologue (quipamd_decode_attention_fused):
    Uops: the three OrthoOp of the projections (applied transposed); ys: their GEMM outputs in the projected basis, fp16 [bs, heads*hd];
    biases: fp16 [heads*hd] each; rotary tables float32 [rows, hd] (Llama) or None.  Returns out fp16 [bs, heads*hd]."""
    _need_gpu(kcache, vcache, pos, *ys)
    bs, heads, maxlen, hd = kcache.shape
    n = heads * hd
    assert len(Uops) == len(ys) == len(biases) == 3 and kcache.dtype == vcache.dtype == torch.float16
    assert kcache.is_contiguous() and vcache.is_contiguous() and pos.dtype == torch.int64 and pos.numel() == 1
    for y, bv in zip(ys, biases):
        assert y.dtype == torch.float16 and y.shape == (bs, n) and y.is_contiguous() and bv.dtype == torch.float16 and bv.numel() == n
    fops = (Fop * 3)(*[o.fop(True) for o in Uops])
    vp = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    out = torch.empty((bs, n), dtype=torch.float16, device=kcache.device)
    rows = 0
    if cos_table is not None:
        assert cos_table.dtype == sin_table.dtype == torch.float32 and cos_table.shape == sin_table.shape and cos_table.shape[1] == hd
        assert cos_table.is_contiguous() and sin_table.is_contiguous()
        rows = cos_table.shape[0]
    sc = float(scale) if scale is not None else 1.0 / (hd ** 0.5)
    _lib.call("quipamd_decode_attention_fused", ctypes.cast(fops, ctypes.c_void_p), ctypes.cast(vp(ys), ctypes.c_void_p),
              ctypes.cast(vp(biases), ctypes.c_void_p), _p(kcache), _p(vcache), _p(pos), _p(out), _p(cos_table), _p(sin_table), rows, bs,
              heads, hd, maxlen, ctypes.c_float(sc), n, _stream())
    return out


def argmax_rows(x, out=None):
    """out[r] = argmax x[r, :] as int64 on the device (first index among equal maxima), one launch (quipamd_argmax_rows)"""
    _need_gpu(x)
    assert x.dim() == 2 and x.stride(1) == 1
    if out is None:
        out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    assert out.dtype == torch.int64 and out.numel() == x.shape[0] and out.is_contiguous()
    _lib.call("quipamd_argmax_rows", _p(x), _dtype(x), x.shape[0], x.shape[1], x.stride(0), _p(out), _stream())
    return out


class HeadArgs(ctypes.Structure):
    """mirror of `quipamd_head_args`"""
    _fields_ = [("has_u", ctypes.c_int), ("U", Fop), ("u_y", ctypes.c_void_p), ("u_y_dtype", ctypes.c_int), ("u_bias", ctypes.c_void_p),
                ("u_residual", ctypes.c_void_p), ("ld_residual", ctypes.c_int64), ("x", ctypes.c_void_p), ("ldx", ctypes.c_int64),
                ("norm", ctypes.c_int), ("ln_gamma", ctypes.c_void_p), ("ln_beta", ctypes.c_void_p), ("ln_eps", ctypes.c_float),
                ("n", ctypes.c_int64), ("bs", ctypes.c_int64), ("W", ctypes.c_void_p), ("vocab", ctypes.c_int64), ("logits", ctypes.c_void_p),
                ("ld_logits", ctypes.c_int64), ("part_val", ctypes.c_void_p), ("part_idx", ctypes.c_void_p), ("nparts", ctypes.c_int),
                ("pos_inc", ctypes.c_void_p)]


HEAD_PARTS = 256          # workgroups of the head launch = entries per row of the argmax partials


def decode_head(W, logits, ln_gamma, ln_beta, ln_eps, x=None, U=None, u_y=None, u_bias=None, u_residual=None, part_val=None, part_idx=None,
                pos_inc=None):
    """ONE launch for the end of a decode step (quipamd_decode_head): t = U^T u_y + u_bias + u_residual (or t = x), the final LayerNorm
    (ln_beta given) / RMSNorm (ln_beta None), logits = W h into `logits` (fp16 [bs, vocab]), per-workgroup argmax partials
    (part_val fp32 / part_idx int32 [bs, HEAD_PARTS]), pos_inc (int64 [1]) += 1.  U: OrthoOp.fop(True); u_y fp16 or fp32 in ZT order."""
    _need_gpu(W)
    vocab, n = W.shape
    a = HeadArgs()
    a.has_u = int(U is not None)
    if U is not None:
        bs = u_y.shape[0]
        assert u_y.dtype in (torch.float16, torch.float32) and u_y.is_contiguous() and u_y.shape[1] == n
        assert u_bias.dtype == torch.float16 and u_bias.numel() == n
        assert u_residual is None or (u_residual.dtype == torch.float16 and u_residual.stride(1) == 1)
        a.U, a.u_y, a.u_y_dtype, a.u_bias = U, _ptr(u_y), _DT[u_y.dtype], _ptr(u_bias)
        a.u_residual, a.ld_residual = _ptr(u_residual), 0 if u_residual is None else u_residual.stride(0)
    else:
        bs = x.shape[0]
        assert x.dtype == torch.float16 and x.stride(1) == 1 and x.shape[1] == n
        a.x, a.ldx = _ptr(x), x.stride(0)
    assert W.dtype == torch.float16 and W.is_contiguous() and logits.dtype == torch.float16 and logits.stride(1) == 1 and logits.shape == (bs, vocab)
    assert ln_gamma.dtype == torch.float16 and (ln_beta is None or ln_beta.dtype == torch.float16)
    a.norm, a.ln_gamma, a.ln_beta, a.ln_eps = (1 if ln_beta is not None else 2), _ptr(ln_gamma), _ptr(ln_beta), float(ln_eps)
    a.n, a.bs, a.W, a.vocab, a.logits, a.ld_logits = n, bs, _ptr(W), vocab, _ptr(logits), logits.stride(0)
    a.nparts = HEAD_PARTS
    if part_val is not None:
        assert part_val.dtype == torch.float32 and part_idx.dtype == torch.int32 and part_val.shape == part_idx.shape == (bs, HEAD_PARTS)
        assert part_val.is_contiguous() and part_idx.is_contiguous()
        a.part_val, a.part_idx = _ptr(part_val), _ptr(part_idx)
    if pos_inc is not None:
        assert pos_inc.dtype == torch.int64 and pos_inc.numel() == 1
        a.pos_inc = _ptr(pos_inc)
    _lib.call("quipamd_decode_head", ctypes.byref(a), _stream())


def decode_embed(tok_table, ids, out, pos_table=None, pos=None, pos_offset=0, part_val=None, part_idx=None):
    """ONE launch for the start of a decode step (quipamd_decode_embed): ids[r] = argmax over the previous head launch's partials (when any
    is valid), out[r] = tok_table[ids[r]] + pos_table[pos + pos_offset] (fp16)."""
    _need_gpu(tok_table)
    vocab, n = tok_table.shape
    bs = ids.numel()
    assert tok_table.dtype == torch.float16 and tok_table.is_contiguous() and ids.dtype == torch.int64 and ids.is_contiguous()
    assert out.dtype == torch.float16 and out.stride(1) == 1 and out.shape == (bs, n)
    assert pos_table is None or (pos_table.dtype == torch.float16 and pos_table.is_contiguous() and pos_table.shape[1] == n and pos.dtype == torch.int64)
    if part_val is not None:
        assert part_val.dtype == torch.float32 and part_idx.dtype == torch.int32 and part_val.shape == part_idx.shape and part_val.shape[0] == bs
        assert part_val.is_contiguous() and part_idx.is_contiguous()
    _lib.call("quipamd_decode_embed", _ptr(tok_table), vocab, _ptr(pos_table), 0 if pos_table is None else pos_table.shape[0], int(pos_offset),
              _ptr(pos), _ptr(ids), _ptr(part_val), _ptr(part_idx), 0 if part_val is None else part_val.shape[1], n, _ptr(out), out.stride(0), bs,
              _stream())


def decode_u_only(U, y, bias16, residual=None, relu=False):
    """out = [relu](U^T y + bias + residual) as one small launch (quipamd_decode_u_only): y fp16 [bs, n] in ZT order of U"""
    _need_gpu(y)
    n = U.n
    assert y.dtype == torch.float16 and y.shape[1] == n and y.is_contiguous() and bias16.dtype == torch.float16 and bias16.numel() == n
    assert residual is None or (residual.dtype == torch.float16 and residual.stride(1) == 1)
    out = torch.empty((y.shape[0], n), dtype=torch.float16, device=y.device)
    fop = U.fop(True)
    _lib.call("quipamd_decode_u_only", ctypes.byref(fop), _p(y), _p(bias16), _p(residual), 0 if residual is None else residual.stride(0),
              int(bool(relu)), _p(out), n, y.shape[0], _stream())
    return out
