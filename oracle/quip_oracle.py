"""oracle/quip_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy + the small C file ldlq_oracle.c) of the reference's
low-bit linear hot path, Cornell-RelaxML/QuIP.  Each function cites the
reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import this module, and only as the checker;
nothing under quip_amd/ imports it.

Pinning: tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which tests/golden/make_golden.py produced by running the
reference itself (imported from /root/reference) in the authoring container.

Floating-point conventions
  * torch CPU arithmetic on fp16 tensors computes each elementwise op in fp32
    and rounds the result to fp16; numpy's float16 does the same, so `np.float16`
    arrays below reproduce the reference's fp16 intermediate roundings
    (SURVEY.md section 2 #9 "dtype trap").
  * integer work (pack/unpack) is bit-exact.
"""
import ctypes
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    """ldlq_oracle.c compiled by oracle/build.py (gcc)."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            from . import build as _b
            _b.build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------- grids
def quantize_qfna(x, scale, zero, maxq):
    """quant.py:6-8."""
    q = np.clip(np.round(x / scale) + zero, 0, maxq)
    return scale * (q - zero)


def quantize_qfnb(x, scale, maxq):
    """quant.py:10-15 (x, scale in the same dtype; maxq an integer)."""
    dt = x.dtype.type
    q = x / scale
    q = np.clip(np.round(((q + dt(1)) / dt(2)) * dt(maxq)), 0, maxq).astype(x.dtype)
    q = (q / dt(maxq)) * dt(2) - dt(1)
    return (q * scale).astype(x.dtype)


def quantize_qfnc(x, scale, zero, maxq):
    """quant.py:17-21."""
    q = np.clip((x / scale) + zero, 0, maxq)
    q = np.round(q)
    return scale * (q - zero)


def qfnb_scale(x):
    """quant.py:150 / vector_balance.py:522: 2.4*sqrt(mean(x^2)) + 1e-16, evaluated
    in the dtype of x.  For fp16 x: squares are rounded to fp16, the mean is
    accumulated wide and rounded to fp16, sqrt and the *2.4 each round to fp16
    (the multiply takes the python scalar in fp32), and +1e-16 is absorbed."""
    if x.dtype == np.float16:
        sq = (x * x)                                         # fp16 squares
        mean = np.float16(np.sum(sq.astype(np.float64)) / sq.size)
        root = np.float16(np.sqrt(np.float32(mean)))
        s = np.float16(np.float32(root) * np.float32(2.4))
        return np.float16(np.float32(s) + np.float32(1e-16))
    sq = x.astype(np.float32) * x.astype(np.float32)
    mean = np.float32(np.sum(sq.astype(np.float64)) / sq.size)
    return np.float32(np.float32(2.4) * np.sqrt(mean) + np.float32(1e-16))


def find_params_qfna(x, bits, perchannel=True, sym=False):
    """quant.py:57-136 with weight=True, mse=False: returns (scale, zero) shaped [m,1]
    in fp32 (the reference promotes through an fp32 `tmp`, quant.py:76-78)."""
    maxq = np.float32(2 ** bits - 1)
    shape = x.shape
    xf = x.reshape(shape[0], -1).astype(np.float32) if perchannel else x.reshape(1, -1).astype(np.float32)
    xmin = np.minimum(xf.min(1), np.float32(0))
    xmax = np.maximum(xf.max(1), np.float32(0))
    if sym:
        xmax = np.maximum(np.abs(xmin), xmax)
        neg = xmin < 0
        xmin = np.where(neg, -xmax, xmin)
    both0 = (xmin == 0) & (xmax == 0)
    xmin = np.where(both0, np.float32(-1), xmin)
    xmax = np.where(both0, np.float32(1), xmax)
    scale = ((xmax - xmin) / maxq).astype(np.float32)
    if sym:
        zero = np.full_like(scale, (maxq + 1) / 2)
    else:
        zero = np.round(-xmin / scale).astype(np.float32)
    if not perchannel:
        scale = np.repeat(scale, shape[0])
        zero = np.repeat(zero, shape[0])
    return scale.reshape(-1, 1), zero.reshape(-1, 1)


def gridmap_qfnb(w, maxq):
    """vector_balance.py:522-524: returns (scale, grid coordinates) in w's dtype."""
    dt = w.dtype.type
    scale = qfnb_scale(w)
    wr = w / scale
    wr = np.clip(((wr + dt(1)) / dt(2)) * dt(maxq), 0, maxq).astype(w.dtype)
    return scale, wr


def gridmap_qfna(w, scale, zero, maxq):
    """vector_balance.py:515 (fp32: scale/zero are fp32 [m,1])."""
    return np.clip((w.astype(np.float32) / scale) + zero, 0, maxq).astype(np.float32)


def codes_to_weight_qfnb(codes, scale, maxq):
    """vector_balance.py:528-530: fp32 ops then .half()."""
    wr = codes.astype(np.float32)
    wr = (wr / np.float32(maxq)) * np.float32(2) - np.float32(1)
    wr = wr * np.float32(scale)
    return wr.astype(np.float16)


def codes_to_weight_qfna(codes, scale, zero):
    """vector_balance.py:519-520."""
    return (scale * (codes.astype(np.float32) - zero)).astype(np.float16)


# --------------------------------------------------------------------------- pack / unpack
def pack_canonical(codes, bits):
    """zeroShot/models/quant.py:190-199 (int4 rule) generalised to bits in {2,4,8}:
    intweight = codes.T [d,m]; qweight[i // per] |= intweight[i] << (bits * (i % per)).
    Returns int32 [d/per, m]."""
    m, d = codes.shape
    per = 32 // bits
    assert d % per == 0
    iw = codes.T.astype(np.uint32)                            # [d, m]
    q = np.zeros((d // per, m), dtype=np.uint32)
    for i in range(d):
        q[i // per] |= iw[i] << np.uint32(bits * (i % per))
    return q.view(np.int32)


def unpack_canonical(qweight, bits, d):
    per = 32 // bits
    q = qweight.view(np.uint32)
    m = q.shape[1]
    codes = np.zeros((m, d), dtype=np.uint8)
    mask = np.uint32((1 << bits) - 1)
    for i in range(d):
        codes[:, i] = ((q[i // per] >> np.uint32(bits * (i % per))) & mask).astype(np.uint8)
    return codes


def pack3(codes):
    """quant.py:192-220 (Quant3Linear.pack).  The reference's three-phase loop is a
    little-endian bit stream: within each run of 32 input columns, code j occupies stream
    bits [3j, 3j+3) of a 96-bit group stored as 3 consecutive int32 rows; 1024 columns ->
    96 rows.  Restated here as exactly that bit stream."""
    m, d = codes.shape
    assert d % 1024 == 0
    c = codes.T.astype(np.uint64).reshape(d // 32, 32, m)        # [group, j, m]
    lo = np.zeros((d // 32, m), dtype=np.uint64)                 # stream bits 0..63
    hi = np.zeros((d // 32, m), dtype=np.uint64)                 # stream bits 64..95
    for j in range(32):
        pos = 3 * j
        if pos + 3 <= 64:
            lo |= c[:, j] << np.uint64(pos)
        elif pos >= 64:
            hi |= c[:, j] << np.uint64(pos - 64)
        else:                                                    # j = 21 straddles bit 64
            lo |= (c[:, j] << np.uint64(pos)) & np.uint64(0xFFFFFFFFFFFFFFFF)
            hi |= c[:, j] >> np.uint64(64 - pos)
    q = np.empty((d // 32, 3, m), dtype=np.uint32)
    q[:, 0] = (lo & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    q[:, 1] = (lo >> np.uint64(32)).astype(np.uint32)
    q[:, 2] = (hi & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return q.reshape(d // 32 * 3, m).view(np.int32)


# "stream" layout: NOT in the reference (it has no int2 packer and no matmul kernel
# source).  It is the declared permutation of the canonical
# layout that quip_amd/csrc/dqgemm.hip streams; this function is its specification.
STREAM_ROWS = 16


def stream_chunk(bits):
    """columns covered by one 16-row x 1 KiB tile: 16 B per lane = 128 / bits codes per lane,
    4 lane groups -> 512 / bits columns."""
    assert bits in (2, 4)
    return 512 // bits


def pack_stream(codes, bits):
    """Specification of layout 1 ("stream").  Tile = 16 rows x KC columns
    (KC = 256 for 2 bit, 128 for 4 bit) = 64 lanes x 4 dwords, stored
    [row_tile][k_chunk][lane][dword].  Lane l = 16*g + j holds row 16*rt + j.
    Inside the tile MFMA step t covers k in [32t, 32t+32) and lane group g the 8
    columns k = 32t + 8g + e, e = 0..7.
      2 bit: dword u holds steps t = 2u, 2u+1; field i = 4*(t&1) + e//2 sits at bits
             [2i, 2i+2) for even e and [16+2i, 16+2i+2) for odd e.
      4 bit: dword u holds step t = u; field i = e//2 at bits [4i,4i+4) (even e) and
             [16+4i, 16+4i+4) (odd e)."""
    m, d = codes.shape
    KC = stream_chunk(bits)
    assert m % STREAM_ROWS == 0 and d % KC == 0
    nt = KC // 32
    out = np.zeros((m // 16, d // KC, 64, 4), dtype=np.uint32)
    c = codes.reshape(m // 16, 16, d // KC, nt, 4, 8).astype(np.uint32)   # [rt, j, kc, t, g, e]
    for t in range(nt):
        for e in range(8):
            if bits == 2:
                u, i = t // 2, 4 * (t & 1) + e // 2
            else:
                u, i = t, e // 2
            sh = bits * i + (16 if (e & 1) else 0)
            v = c[:, :, :, t, :, e]                                        # [rt, j, kc, g]
            v = np.transpose(v, (0, 2, 3, 1)).reshape(m // 16, d // KC, 64)  # lane = 16g + j
            out[:, :, :, u] |= v << np.uint32(sh)
    return out.reshape(-1).view(np.int32)


def unpack_stream(packed, bits, m, d):
    KC = stream_chunk(bits)
    nt = KC // 32
    p = packed.view(np.uint32).reshape(m // 16, d // KC, 64, 4)
    codes = np.zeros((m // 16, 16, d // KC, nt, 4, 8), dtype=np.uint8)
    mask = np.uint32((1 << bits) - 1)
    for t in range(nt):
        for e in range(8):
            if bits == 2:
                u, i = t // 2, 4 * (t & 1) + e // 2
            else:
                u, i = t, e // 2
            sh = bits * i + (16 if (e & 1) else 0)
            v = (p[:, :, :, u] >> np.uint32(sh)) & mask                    # [rt, kc, lane]
            v = v.reshape(m // 16, d // KC, 4, 16)                         # [rt, kc, g, j]
            codes[:, :, :, t, :, e] = np.transpose(v, (0, 3, 1, 2))
    return codes.reshape(m, d)


# --------------------------------------------------------------------------- packed linear
def bf16_round(x):
    """round-to-nearest-even fp32 -> bf16 -> fp32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def dequant_weight(codes, qfn, scale, zero, maxq):
    """Dense fp64 weights from integer codes: qfn b quant.py:13-14 ((q/maxq)*2-1)*s;
    qfn a quant.py:8 scale*(q-zero)."""
    c = codes.astype(np.float64)
    if qfn == 'b':
        return ((c / maxq) * 2.0 - 1.0) * float(scale)
    return np.asarray(scale, dtype=np.float64).reshape(-1, 1) * (c - np.asarray(zero, dtype=np.float64).reshape(-1, 1))


def dequant_linear(x, codes, qfn, scale, zero, maxq, bias=None):
    """The packed layer's forward, y = x @ What^T + bias (quant.py:222-233 semantics,
    SURVEY.md 8(c) K2 oracle): x already rounded to the activation dtype, fp64 math."""
    W = dequant_weight(codes, qfn, scale, zero, maxq)
    y = x.astype(np.float64) @ W.T
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float64)
    return y


def packed_matmul_c(x, qweight, y, scales, zeros, bits):
    """C restatement of quant_cuda.vecquant{3,4}matmul's contract (ldlq_oracle.c)."""
    bsz, d = x.shape
    m = qweight.shape[1]
    x = np.ascontiguousarray(x, np.float32)
    qweight = np.ascontiguousarray(qweight, np.int32)
    sc = np.ascontiguousarray(scales, np.float32).reshape(-1)
    zs = np.ascontiguousarray(zeros, np.float32).reshape(-1)
    assert y.dtype == np.float32 and y.flags.c_contiguous
    _lib().oracle_packed_matmul(_fp(x), _fp(qweight), _fp(y), _fp(sc), _fp(zs),
                                ctypes.c_int64(bsz), ctypes.c_int64(m), ctypes.c_int64(d),
                                ctypes.c_int(bits))
    return y


# --------------------------------------------------------------------------- butterfly
def prime_factors(n):
    """ascending prime factors (what primefac.primefac yields for these n; method.py:17)."""
    out, f = [], 2
    while f * f <= n:
        while n % f == 0:
            out.append(f)
            n //= f
        f += 1 if f == 2 else 2
    if n > 1:
        out.append(n)
    return out


def butterfly_factors(n):
    """method.py:16-18."""
    pf = prime_factors(n)
    return (math.prod(pf[0::2]), math.prod(pf[1::2]))


def mul_ortho_butterfly(Bpp, x, transpose=False):
    """method.py:46-67 in index form.  B0: [q,p,p] or [1,p,p] /
    [p,p]; B1: [p,q,q] or [1,q,q] / [q,q].  transpose=True applies the inverse."""
    (B, p_in, p_out) = Bpp
    x = np.asarray(x)
    one_d = x.ndim == 1
    if one_d:
        x = x.reshape(-1, 1)
    n, c = x.shape
    p, q = butterfly_factors(n)
    B0 = np.asarray(B[0], dtype=x.dtype).reshape(-1, p, p)
    B1 = np.asarray(B[1], dtype=x.dtype).reshape(-1, q, q)
    B0 = np.broadcast_to(B0, (q, p, p)) if B0.shape[0] == 1 else B0
    B1 = np.broadcast_to(B1, (p, q, q)) if B1.shape[0] == 1 else B1
    p_in = np.asarray(p_in)
    p_out = np.asarray(p_out)
    if not transpose:
        z = x[p_in].reshape(p, q, c)
        z = np.einsum('bac,cbk->abk', B0, z)
        z = np.einsum('abc,ack->abk', B1, z)
        y = z.reshape(n, c)[p_out]
    else:
        z = np.empty_like(x)
        z[p_out] = x
        z = z.reshape(p, q, c)
        z = np.einsum('acb,ack->abk', B1, z)
        z = np.einsum('bca,cbk->abk', B0, z)
        y = np.empty_like(x)
        y[p_in] = z.reshape(n, c)
    return y.reshape(n) if one_d else y


# --------------------------------------------------------------------------- LDLQ
def ldl_factor(H):
    """vector_balance.py:171-173: L = chol(H); L = L @ diag(1/diag L); L -= I (fp32)."""
    L = np.linalg.cholesky(H.astype(np.float64)).astype(np.float32)
    L = (L * (np.float32(1) / np.diag(L))[None, :]).astype(np.float32)
    return (L - np.eye(H.shape[0], dtype=np.float32)).astype(np.float32)


def round_ldl(w, H, nbits, eta=None, L=None):
    """vector_balance.py:155-199 with n_greedy_passes=0.  Returns fp32 integer codes."""
    w = np.ascontiguousarray(w, np.float32)
    m, d = w.shape
    L = np.ascontiguousarray(ldl_factor(H) if L is None else L, np.float32)
    out = np.empty_like(w)
    e = None if eta is None else np.ascontiguousarray(eta, np.float32)
    _lib().oracle_round_ldl(_fp(w), _fp(L), _fp(e) if e is not None else None, _fp(out),
                            ctypes.c_int64(m), ctypes.c_int64(d), ctypes.c_int(nbits))
    return out


def greedy_passes(w, w_hat, H, nbits, n_greedy_passes):
    """vector_balance.py:182-196: coordinate descent over the integer grid after round_ldl.  fp32 like the reference;
    np.round and torch.round both round half to even.  Returns the final codes (fp32)."""
    w = np.asarray(w, np.float32)
    w_hat = np.array(w_hat, np.float32)
    d = w.shape[1]
    wr = w_hat.copy()
    s = (w_hat - w).astype(np.float32)
    H = (np.asarray(H, np.float32) / np.float32(np.diag(H).max())).astype(np.float32)
    for _ in range(n_greedy_passes):
        for i in range(d - 1, -1, -1):
            Hs = (s @ H[:, i]).astype(np.float32)
            eps = (wr[:, i] - np.round(wr[:, i] - Hs / H[i, i])).astype(np.float32)
            wr[:, i] -= eps
            s[:, i] -= eps
        wr = np.clip(wr, 0, 2 ** nbits - 1).astype(np.float32)
        if (w_hat == wr).all():
            break
        w_hat = wr.copy()
    return wr


def round_ldl_gptqequiv(w, H, nbits, eta=None):
    """vector_balance.py:381-422."""
    w = np.ascontiguousarray(w, np.float32)
    m, d = w.shape
    Hf = H[::-1, ::-1]
    L = np.linalg.cholesky(Hf.astype(np.float64)).astype(np.float32)
    L = L[::-1, ::-1]
    L = (L * (np.float32(1) / np.diag(L))[None, :]).astype(np.float32)
    L = np.ascontiguousarray(L - np.eye(d, dtype=np.float32), np.float32)
    out = np.empty_like(w)
    e = None if eta is None else np.ascontiguousarray(eta, np.float32)
    _lib().oracle_round_ldl_forward(_fp(w), _fp(L), _fp(e) if e is not None else None, _fp(out),
                                    ctypes.c_int64(m), ctypes.c_int64(d), ctypes.c_int(nbits))
    return out


def round_ldl_kernel_order(w, LT, nbits, eta=None, blocksize=128):
    """Same algorithm as round_ldl_block (vector_balance.py:218-257) in the HIP
    kernel's documented evaluation order (ldlq_oracle.c); LT[c][j] = L[j][c]."""
    w = np.ascontiguousarray(w, np.float32)
    LT = np.ascontiguousarray(LT, np.float32)
    m, d = w.shape
    codes = np.empty((m, d), dtype=np.uint8)
    e = None if eta is None else np.ascontiguousarray(eta, np.float32)
    _lib().oracle_round_ldl_kernel_order(_fp(w), _fp(LT), _fp(e) if e is not None else None,
                                         _fp(codes), ctypes.c_int64(m), ctypes.c_int64(d),
                                         ctypes.c_int(nbits), ctypes.c_int(blocksize))
    return codes


def proxy_loss(dw, H):
    """vector_balance.py:14-15 / method.py:228-231: tr(dW H dW^T), fp64 here."""
    dw = dw.astype(np.float64)
    return float(np.einsum('ij,jk,ik->', dw, H.astype(np.float64), dw))


def quantize_weight_vecbal(w, H, nbits, scale, zero, qfn):
    """vector_balance.py:500-532 for qmethod='ldlq', npasses=0, unbiased=False."""
    maxq = 2 ** nbits - 1
    if qfn == 'a':
        wr = gridmap_qfna(w, scale, zero, maxq)
        codes = round_ldl(wr, H, nbits)
        return codes_to_weight_qfna(codes, scale, zero), codes
    s, wr = gridmap_qfnb(w, maxq)
    codes = round_ldl(wr.astype(np.float32), H, nbits)
    return codes_to_weight_qfnb(codes, s, maxq), codes


# --------------------------------------------------------------------------- OPTQ / GPTQ
def gptq_round(W, H, scale, zero, maxq, blocksize=128):
    """gptq.py:51-93 (groupsize -1, qfn a): upper Cholesky factor of H^-1, then column by column
    q = quantize_qfna(w); e = (w - q) / Hinv[i][i]; W[:, i:] -= e Hinv[i][i:], lazily across 128-column blocks.
    fp32 like the reference.  Returns (Q fp32 [m,d], codes)."""
    W = np.array(W, np.float32)
    m, d = W.shape
    Hinv = np.linalg.cholesky(np.linalg.inv(np.asarray(H, np.float64))).T.astype(np.float32)   # upper: H^-1 = Hinv^T Hinv
    Q = np.zeros_like(W)
    codes = np.zeros((m, d), np.float32)
    s, z = np.asarray(scale, np.float32).reshape(-1, 1), np.asarray(zero, np.float32).reshape(-1, 1)
    for i1 in range(0, d, blocksize):
        i2 = min(i1 + blocksize, d)
        Wb = W[:, i1:i2].copy()
        Eb = np.zeros_like(Wb)
        Hb = Hinv[i1:i2, i1:i2]
        for i in range(i2 - i1):
            col = Wb[:, i:i + 1]
            c = np.clip(np.round(col / s) + z, 0, maxq).astype(np.float32)          # quant.py:6-8 (np.round: half to even)
            q = (s * (c - z)).astype(np.float32)
            Q[:, i1 + i], codes[:, i1 + i] = q[:, 0], c[:, 0]
            e = ((col - q) / Hb[i, i]).astype(np.float32)
            Wb[:, i:] -= e * Hb[i:i + 1, i:]
            Eb[:, i] = e[:, 0]
        W[:, i2:] -= Eb @ Hinv[i1:i2, i2:]
    return Q, codes


# --------------------------------------------------------------------------- Hessian accumulation
def hessian_add_batch(H, inp):
    """method.py:98-120 for nn.Linear / Conv1D layers: a 2-D input counts as one call, a 3-D input as inp.shape[0];
    tokens are flattened, widened to fp64 and `H += inp^T inp` (the reference forms inp.t() first and multiplies
    inp.matmul(inp.t())).  H: float64 [d,d], updated in place.  Returns the increment of nsamples."""
    inp = np.asarray(inp)
    if inp.ndim == 2:
        inp = inp[None]
    n_calls = inp.shape[0]
    x = inp.reshape(-1, inp.shape[-1]).astype(np.float64)
    H += x.T @ x
    return n_calls


def hessian_post_batch(H, nsamples):
    """method.py:122-123: fp64 division, then narrowing to fp32."""
    return (np.asarray(H, np.float64) / nsamples).astype(np.float32)


# --------------------------------------------------------------------------- preproc / postproc
def preproc(W, H, layer_dtype, rescale, proj, gptqH, percdamp=0.01, U=None, V=None):
    """method.py:125-193 with the orthogonal operators injected as (factors, p_in, p_out)
    tuples U (rows) and V (columns).  W is re-rounded to `layer_dtype` after every stage
    exactly where the reference does (method.py:155,179,191).  Returns (W, H, scaleWH)."""
    W = np.asarray(W, layer_dtype)
    H = np.asarray(H, np.float32)
    scaleWH = None
    if rescale:                                               # method.py:139-156
        w = W.astype(np.float32)
        Hs = H / np.abs(H).max()
        diagH = np.clip(np.diag(Hs), 1e-8, None)
        diagW2 = np.clip((w.astype(np.float64) ** 2).sum(0).astype(np.float32), 1e-8, None)
        scaleWH = np.sqrt(np.sqrt(diagH / diagW2)).astype(np.float32)
        scaleWH = np.clip(scaleWH, 1e-8, None)
        w = w * scaleWH[None, :]
        Hs = Hs / scaleWH[None, :]
        Hs = Hs / scaleWH[:, None]
        W = w.astype(layer_dtype)
        H = Hs.astype(np.float32)
    if proj:                                                  # method.py:157-180
        w = W.astype(np.float32)
        n = H.shape[0]
        Hn = (H * np.float32(n / (np.trace(H) + 1e-8)) + np.float32(1e-2) * np.eye(n, dtype=np.float32)).astype(np.float32)
        w = mul_ortho_butterfly(U, w)                        # U @ w
        w = mul_ortho_butterfly(V, w.T.copy()).T              # (V @ w^T)^T = w V^T
        Hn = mul_ortho_butterfly(V, Hn)                       # V @ H
        Hn = mul_ortho_butterfly(V, Hn.T.copy()).T            # V H V^T
        W = w.astype(layer_dtype)
        H = Hn.astype(np.float32)
    if gptqH:                                                 # method.py:182-192
        w = W.copy()
        Hd = H.copy()
        dead = np.diag(Hd) == 0
        Hd[dead, dead] = 1
        w[:, dead] = 0
        damp = np.float32(percdamp) * np.mean(np.diag(Hd), dtype=np.float32)
        idx = np.arange(Hd.shape[0])
        Hd[idx, idx] += damp
        W, H = w, Hd.astype(np.float32)
    return W, H, scaleWH


def postproc(W, H, layer_dtype, rescale, proj, scaleWH=None, U=None, V=None):
    """method.py:195-214."""
    W = np.asarray(W, layer_dtype)
    H = np.asarray(H, np.float32)
    if proj:
        w = W.astype(np.float32)
        w = mul_ortho_butterfly(U, w, transpose=True)                     # U^T w
        w = mul_ortho_butterfly(V, w.T.copy(), transpose=True).T          # w V
        Hn = mul_ortho_butterfly(V, H, transpose=True)
        Hn = mul_ortho_butterfly(V, Hn.T.copy(), transpose=True).T
        W = w.astype(layer_dtype)
        H = Hn.astype(np.float32)
    if rescale:
        w = W.astype(np.float32) / scaleWH[None, :].astype(np.float32)       # fp16 tensor / fp32 tensor -> fp32
        Hn = H * scaleWH[:, None]
        Hn = Hn * scaleWH[None, :]
        W = w.astype(layer_dtype)
        H = Hn.astype(np.float32)
    return W, H
