"""CPU test: the index algebra of dq_mb_kernel<..., T32> (csrc/dqgemm_v2.h) -- how the 16-row STREAM tiles become the 32-row A operand of
v_mfma_f32_32x32x16 by a lane permutation through LDS, which x chunk a lane reads, and where a lane's 16 results go.  The operand layouts of the
two MFMA shapes are the ones the GPU parity tests pin (tests/test_gpu_dqgemm_v2.py runs configuration 47); this file restates the derivation so
that a change to one side of it fails here, without a GPU."""
import itertools


def stream_lane_k(lane, t):
    """STREAM tile (= the A fragment of 16x16x32): lane -> (row, the 8 k of step t)"""
    return lane % 16, [32 * t + 8 * (lane // 16) + e for e in range(8)]


def a32_wants(L, s):
    """A operand of 32x32x16, step s of a 256-k tile: lane -> (row, the 8 k)"""
    return L % 32, [16 * s + 8 * (L // 32) + e for e in range(8)]


def test_the_32_row_operand_is_a_lane_permutation_of_two_stream_tiles():
    for L, s in itertools.product(range(64), range(16)):
        l32, kg = L % 32, L // 32
        tile = l32 // 16                                    # which tile of the pair
        old_lane = (l32 % 16) + 16 * (2 * (s & 1) + kg)     # wE (even s) at + 16 kg, wO (odd s) 32 lanes = 512 bytes further
        row, ks = stream_lane_k(old_lane, s // 2)           # the fragment frag(w, t = s / 2) dequantises
        want_row, want_ks = a32_wants(L, s)
        assert (16 * tile + row, ks) == (want_row, want_ks)
        assert old_lane * 16 == ((l32 & 15) + 16 * kg) * 16 + (512 if s & 1 else 0)     # the two ds_read_b128 addresses inside a tile


def test_x_chunk_of_a_step():
    # column block cb = 64 k = 8 chunks of 8 k; step s = 4 cb + q reads chunk 2 q + L / 32
    for L, s in itertools.product(range(64), range(16)):
        cb, q = s // 4, s % 4
        ks = [64 * cb + 8 * (2 * q + L // 32) + e for e in range(8)]
        assert ks == a32_wants(L, s)[1]                     # B operand: same k as the A operand of the lane's k-group


def test_results_go_out_as_quads_of_consecutive_weight_rows():
    # D of 32x32: register r of lane L = row 8 (r / 4) + 4 (L / 32) + r % 4, column L % 32; the epilogue stores quad rg = r / 4 of pair p at
    # row tile 2 p + (rg >> 1), in-tile row 8 (rg & 1) + 4 (L / 32)
    for L, r in itertools.product(range(64), range(16)):
        kg, rg = L // 32, r // 4
        d_row = 8 * rg + 4 * kg + r % 4
        tile, in_tile = rg >> 1, 8 * (rg & 1) + 4 * kg + r % 4
        assert 16 * tile + in_tile == d_row


def test_one_product_gives_both_row_sums():
    # 16x16x32: D[i][j] = sum_k A[i][k] B[k][j]; A row 0 = ones, row 1 = the offsets, others zero -> D[0][j] = S_1, D[1][j] = S_off, and the
    # kernel reads registers 0 and 1 of lanes 0..15 (D row = 4 (lane / 16) + register, column = lane % 16)
    import numpy as np
    rng = np.random.default_rng(0)
    x = rng.standard_normal((32, 16))                       # B: k x batch column
    off = rng.choice([4.0, 16.0, 64.0], size=32)
    A = np.zeros((16, 32)); A[0] = 1.0; A[1] = off
    D = A @ x
    for lane in range(16):
        g, j = lane // 16, lane % 16
        assert np.isclose(D[4 * g + 0, j], x[:, j].sum()) and np.isclose(D[4 * g + 1, j], (off * x[:, j]).sum())
    assert not D[2:].any()
