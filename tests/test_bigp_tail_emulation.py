"""CPU: lane-level emulation of csrc/decode_bigp.hip (the p x 16 operators around Llama's down_proj in a decode step).

As in test_fused_pass_emulation.py, what is under test on the CPU is the HOST side of the contract -- the zero-padded B fragments
(ops._f16_b_frags_padded), the uint16 tables quant._bigp_tail_tables composes from the three layers' permutations, the row / column
orders QuantLinear.decode_qweight folds into the packing -- and the index formulas the two kernels were written from, restated one array
element per (wave, lane, register) with the MFMA operand maps of the guide.  The kernels themselves are checked on the GPU
(tests/test_gpu_decode_bigp.py)."""
import types

import numpy as np
import pytest
import torch

from test_fused_pass_emulation import mfma_16x16x32


def mfma_16x16x4_f32(a, b, acc):
    """one v_mfma_f32_16x16x4_f32: a, b [64 lanes] (A: row = lane % 16, k = lane / 16; B: col = lane % 16, k = lane / 16)"""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    for lane in range(64):
        A[lane % 16, lane // 16] = a[lane]
        B[lane // 16, lane % 16] = b[lane]
    D = A @ B
    out = acc.copy()
    for lane in range(64):
        for reg in range(4):
            out[lane, reg] += D[4 * (lane // 16) + reg, lane % 16]
    return out


def mix_a_tile(zt_rows, F0, at, p, ks):
    """the kernels' first phase for image rows 16 at .. 16 at + 15: zt_rows [16][p] (row b of the transposed image) ->
    T in D layout [64 lanes, 4] (lane (j, g): T[b = 4g + reg][a' = j]), K split over waves two steps each, partials summed"""
    nw = (ks + 1) // 2
    lanes = np.arange(64)
    j, g = lanes % 16, lanes // 16
    F0 = F0.reshape(p // 16, ks, 64, 8).astype(np.float64)
    T = np.zeros((64, 4))
    for wave in range(nw):
        acc = np.zeros((64, 4))
        for S in (2 * wave, 2 * wave + 1):
            if S >= ks:
                continue                                               # the kernel zeroes the fragment instead
            A = np.zeros((64, 8))
            for l in range(64):
                ka = 32 * S + 8 * g[l]
                ka = ka if ka < p else 0                               # clamp: the fragment is zero there
                A[l] = zt_rows[j[l], ka:ka + 8]
            acc = mfma_16x16x32(A, F0[at, S], acc)
        T += acc.astype(np.float32)
    return T


def mix_b(T, M1):
    """bg_mix_b: z2 in D layout, lane (j, g) reg: z2[a' = j][b' = 4g + reg]"""
    lanes = np.arange(64)
    j, g = lanes % 16, lanes // 16
    acc = np.zeros((64, 4))
    for s in range(4):
        a = np.array([M1[j[l], 4 * g[l] + s] for l in range(64)], np.float64)
        acc = mfma_16x16x4_f32(a, T[:, s].astype(np.float64), acc)
    return acc


def emulate_bigp_u(y_zt, F0, M1, bias_img, post_img, dest, p, out):
    ks = (p + 31) // 32
    zt = y_zt.reshape(16, p).astype(np.float64)
    for at in range(p // 16):
        z2 = mix_b(mix_a_tile(zt, F0, at, p, ks), M1)
        for lane in range(64):
            j, g = lane % 16, lane // 16
            pos0 = (16 * at + j) * 16 + 4 * g
            for reg in range(4):
                v = z2[lane, reg] + (0.0 if bias_img is None else float(bias_img[pos0 + reg]))
                v *= 1.0 if post_img is None else float(post_img[pos0 + reg])
                out[int(dest[pos0 + reg]) & 0xffff] = np.float16(v)


def silu16(gv, uv):
    gv, uv = gv.astype(np.float32), uv.astype(np.float32)
    sl = (gv / (1.0 + np.exp(-gv))).astype(np.float16).astype(np.float32)
    return (sl * uv).astype(np.float16)


def emulate_bigp_v_gemm(gate_img, up_img, F0, M1, Wd_cols, p):
    """-> (x~ in image order [n] fp16, y [m] = sum over the K-slices of Wd_cols[:, slice] x~[slice])"""
    ks = (p + 31) // 32
    t = (silu16(gate_img, up_img) if up_img is not None else gate_img).reshape(16, p).astype(np.float64)
    n = 16 * p
    xt = np.zeros(n, np.float16)
    y = np.zeros(Wd_cols.shape[0])
    for at in range(p // 16):
        z2 = mix_b(mix_a_tile(t, F0, at, p, ks), M1)
        sl = np.zeros(256, np.float16)
        for lane in range(64):
            j, g = lane % 16, lane // 16
            sl[16 * j + 4 * g: 16 * j + 4 * g + 4] = z2[lane].astype(np.float16)     # XT[k = 16 j + 4 g + reg]
        xt[256 * at: 256 * at + 256] = sl
        y += Wd_cols[:, 256 * at: 256 * at + 256].astype(np.float64) @ sl.astype(np.float64)
    return xt, y


@pytest.mark.parametrize("p", [80, 96])
def test_bigp_tail_emulation_matches_the_dense_chain(p):
    from quip_amd import ops, quant
    rng = np.random.default_rng(p)
    torch.manual_seed(p)
    n, m = 16 * p, 32

    def gen():
        B0 = np.linalg.qr(rng.standard_normal((p, p)))[0].astype(np.float32)
        B1 = np.linalg.qr(rng.standard_normal((16, 16)))[0].astype(np.float32)
        return ([torch.from_numpy(B0)[None], torch.from_numpy(B1)[None]], torch.from_numpy(rng.permutation(n)), torch.from_numpy(rng.permutation(n)))

    dev = torch.device("cpu")
    Ug, Uu, Vd = (ops.OrthoOp(gen(), dev) for _ in range(3))
    assert Ug.bigp_fold_ok and Ug.fold_ok and not Ug.fused_ok
    bias_g = torch.from_numpy(rng.standard_normal(n).astype(np.float16))
    s_inv = torch.from_numpy((0.5 + rng.random(n)).astype(np.float32))
    gate = types.SimpleNamespace(U=Ug, bias=bias_g.float(), qweight=torch.zeros(1), outfeatures=n)
    up = types.SimpleNamespace(U=Uu, bias=None)
    down = types.SimpleNamespace(V=Vd, inv_scaleWH=s_inv)
    tabs = quant._bigp_tail_tables([gate, up], down)

    # ---- dense chain, fp64, from the generator tuples (method.py:46-67) ------------------------------------------------------------------
    def dense(op, x, transpose):
        B0, B1 = op._B0[0].double().numpy(), op._B1[0].double().numpy()
        pin, pout = op._p_in.numpy(), op._p_out.numpy()
        if not transpose:
            return (B0 @ x[pin].reshape(p, 16) @ B1.T).reshape(-1)[pout]
        z = np.zeros(n)
        z[pout] = x
        out = np.zeros(n)
        out[pin] = (B0.T @ z.reshape(p, 16) @ B1).reshape(-1)
        return out

    yg = rng.standard_normal(n).astype(np.float16)
    yu = rng.standard_normal(n).astype(np.float16)
    g_want = dense(Ug, yg.astype(np.float64), True) + bias_g.double().numpy()
    u_want = dense(Uu, yu.astype(np.float64), True) * s_inv.double().numpy()

    # ---- launch 1 -------------------------------------------------------------------------------------------------------------------------
    imgs = np.zeros((2, n), np.float16)
    for i, (U, y_nat) in enumerate(((Ug, yg), (Uu, yu))):
        y_zt = np.zeros(n, np.float16)
        y_zt[U.zt_rows().numpy()] = y_nat                              # QuantLinear.to_zt: what the producing GEMM writes
        F0, M1 = U.bigp_frags(True)
        dest, bias_img, post = tabs[i]
        emulate_bigp_u(y_zt, F0.numpy(), M1.numpy(), None if bias_img is None else bias_img.numpy(), None if post is None else post.numpy(),
                       dest.numpy().view(np.uint16), p, imgs[i])
    # the images are the transposed input image of V_down: element with natural index i at (b p + a), (a, b) = inv_pin_V[i]
    inv_pin = torch.argsort(Vd._p_in).numpy()
    timg = (inv_pin % 16) * p + inv_pin // 16
    assert np.abs(imgs[0][timg].astype(np.float64) - g_want).max() <= 4e-3 * np.abs(g_want).max()
    assert np.abs(imgs[1][timg].astype(np.float64) - u_want).max() <= 4e-3 * np.abs(u_want).max()

    # ---- launch 2 -------------------------------------------------------------------------------------------------------------------------
    W = rng.integers(0, 4, (m, n)).astype(np.float64) - 1.5           # the layer in its natural column order
    perm = np.empty(n, np.int64)
    perm[Vd.image_cols().numpy()] = np.arange(n)                       # QuantLinear.decode_qweight: columns in image order
    F0, M1 = Vd.bigp_frags(False)
    xt, y = emulate_bigp_v_gemm(imgs[0], imgs[1], F0.numpy(), M1.numpy(), W[:, perm], p)
    t_nat = silu16(imgs[0][timg], imgs[1][timg]).astype(np.float64)
    x_want = dense(Vd, t_nat, False)                                   # natural order of V's output = the layer's columns
    assert np.abs(xt.astype(np.float64)[Vd.image_cols().numpy()] - x_want).max() <= 4e-3 * np.abs(x_want).max()
    y_want = W @ x_want
    assert np.linalg.norm(y - y_want) <= 3e-3 * np.linalg.norm(y_want)
