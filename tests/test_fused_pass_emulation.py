"""CPU: lane-level emulation of the operator pass of quip_amd/csrc/decode_fused.hip.

The kernel's index arithmetic (scatter into the transposed image, the two MFMA stages with the host-prepared B fragments, the
8-byte hand-over between the stages, the gather) is restated here in numpy, one array element per (wave, lane, register), with
the v_mfma_f32_16x16x32_f16 operand maps of the guide (A: row = lane % 16, k = 8 (lane / 16) + e; B: col = lane % 16, same k;
D: col = lane % 16, row = 4 (lane / 16) + reg) -- and checked against the dense operator.  What is under test on the CPU is the
HOST side of the contract (ops._f16_b_frags, the uint16 index vectors of OrthoOp.fop's layout) and the formulas the kernel was
written from; the kernel itself is checked on the GPU (tests/test_gpu_decode_fused.py)."""
import numpy as np
import pytest
import torch

NW = 16


def mfma_16x16x32(A_frag, B_frag, acc):
    """A_frag, B_frag [64 lanes, 8]; acc [64 lanes, 4] -> acc + A B with the gfx950 lane maps"""
    A = np.zeros((16, 32), np.float64)
    B = np.zeros((32, 16), np.float64)
    for lane in range(64):
        j, g = lane % 16, lane // 16
        A[j, 8 * g:8 * g + 8] = A_frag[lane]
        B[8 * g:8 * g + 8, j] = B_frag[lane]
    D = A @ B
    out = acc.copy()
    for lane in range(64):
        j, g = lane % 16, lane // 16
        for reg in range(4):
            out[lane, reg] += D[4 * g + reg, j]
    return out


def emulate_pass(x, M0, M1, load_idx, store_idx, P, Q, F0, F1):
    """the kernel's pass on one row: scatter4 -> mix_stages -> gather4 (fp16 images, fp32 accumulation)"""
    n = P * Q
    PS, QS, QF = P + 8, Q + 8, Q + 4
    S0, S1 = P // 32, Q // 32
    NT = (P // 16) * (Q // 16)
    TPW = (NT + NW - 1) // NW
    qsh = int(np.log2(Q))
    ZT = np.zeros((Q, PS), np.float16)
    Z1 = np.zeros((P, QS), np.float16)
    ZF = np.zeros((P, QF), np.float32)
    F0 = F0.reshape(P // 16, S0, 64, 8)
    F1 = F1.reshape(Q // 16, S1, 64, 8)
    # scatter4: element i -> image position pos = load_idx[i] = (a, b) -> ZT[b][a]
    for i in range(n):
        pos = int(load_idx[i])
        ZT[pos & (Q - 1), pos >> qsh] = np.float16(x[i])
    lanes = np.arange(64)
    j, g = lanes % 16, lanes // 16
    for wave in range(NW):                                   # stage 1
        for it in range(TPW):
            tile = wave + NW * it
            if tile >= NT:
                continue
            at, bt = tile % (P // 16), tile // (P // 16)
            acc = np.zeros((64, 4))
            for S in range(S0):
                A = np.stack([ZT[16 * bt + j[l], 8 * g[l] + 32 * S: 8 * g[l] + 32 * S + 8] for l in range(64)]).astype(np.float64)
                acc = mfma_16x16x32(A, F0[at, S].astype(np.float64), acc)
            for l in range(64):
                Z1[16 * at + j[l], 16 * bt + 4 * g[l]: 16 * bt + 4 * g[l] + 4] = acc[l].astype(np.float32).astype(np.float16)
    for wave in range(NW):                                   # stage 2
        for it in range(TPW):
            tile = wave + NW * it
            if tile >= NT:
                continue
            bt, at = tile % (Q // 16), tile // (Q // 16)
            acc = np.zeros((64, 4))
            for S in range(S1):
                A = np.stack([Z1[16 * at + j[l], 8 * g[l] + 32 * S: 8 * g[l] + 32 * S + 8] for l in range(64)]).astype(np.float64)
                acc = mfma_16x16x32(A, F1[bt, S].astype(np.float64), acc)
            for l in range(64):
                for reg in range(4):
                    ZF[16 * at + 4 * g[l] + reg, 16 * bt + j[l]] = acc[l, reg]
    out = np.zeros(n, np.float32)
    for i in range(n):
        pos = int(store_idx[i])
        out[i] = ZF[pos >> qsh, pos & (Q - 1)]
    return out


@pytest.mark.parametrize("P,Q", [(64, 32), (64, 64), (128, 64)])
@pytest.mark.parametrize("transpose", [False, True])
def test_pass_emulation_matches_the_dense_operator(P, Q, transpose):
    from quip_amd import ops
    rng = np.random.default_rng(P + Q + int(transpose))
    n = P * Q
    B0 = np.linalg.qr(rng.standard_normal((P, P)))[0].astype(np.float32)
    B1 = np.linalg.qr(rng.standard_normal((Q, Q)))[0].astype(np.float32)
    p_in, p_out = rng.permutation(n), rng.permutation(n)
    # the reference operator (method.py:46-67 for the Kronecker generator): y = P_out (B0 (x) B1) P_in x, i.e. y = z2.flat[p_out]
    # with z = x[p_in].reshape(P, Q), z2 = B0 z B1^T;  transpose: x = scatter / the inverse chain
    x = rng.standard_normal(n).astype(np.float32)
    if not transpose:
        want = (B0.astype(np.float64) @ x[p_in].reshape(P, Q).astype(np.float64) @ B1.T.astype(np.float64)).reshape(-1)[p_out]
        M0, M1 = B0, B1
        inv_pin = np.argsort(p_in)
        load_idx, store_idx = inv_pin, p_out                 # OrthoOp.small_op: (inv_pin, pout)
    else:
        z = np.zeros(n)
        z[p_out] = x
        z2 = (B0.T.astype(np.float64) @ z.reshape(P, Q) @ B1.astype(np.float64)).reshape(-1)
        want = np.zeros(n)
        want[p_in] = z2
        M0, M1 = B0.T.copy(), B1.T.copy()
        load_idx, store_idx = p_out, np.argsort(p_in)        # (pout, inv_pin)
    F0 = ops._f16_b_frags(torch.from_numpy(M0)).numpy()
    F1 = ops._f16_b_frags(torch.from_numpy(M1)).numpy()
    # fragment order as the header states it
    lane, e = 37, 5
    assert F0.reshape(P // 16, P // 32, 64, 8)[1, 1, lane, e] == np.float16(M0[16 + lane % 16, 32 + 8 * (lane // 16) + e])
    got = emulate_pass(x, M0, M1, load_idx.astype(np.uint16), store_idx.astype(np.uint16), P, Q, F0, F1)
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rel <= 1.5e-3, rel                                # fp16 images and factors: ~3e-4 per stage + input rounding
