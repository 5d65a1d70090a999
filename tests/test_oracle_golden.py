"""Pins oracle/quip_oracle.py against the fixtures the REFERENCE produced
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import quip_oracle as O
from conftest import load_golden, f16


# ------------------------------------------------------------------ grids (quant.py:6-163)
@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("tag", ["f32", "f16"])
def test_qfna_params_and_grid(bits, tag):
    g = load_golden("grids")
    W = g["W32"] if tag == "f32" else f16(g["W16"])
    scale, zero = O.find_params_qfna(W, bits)
    np.testing.assert_array_equal(scale, g[f"a{bits}_{tag}_scale"])
    np.testing.assert_array_equal(zero, g[f"a{bits}_{tag}_zero"])
    maxq = 2 ** bits - 1
    out = O.quantize_qfna(W.astype(np.float32), scale, zero, maxq)
    np.testing.assert_array_equal(out.astype(np.float32), g[f"a{bits}_{tag}_out"])
    outc = O.quantize_qfnc(W.astype(np.float32), scale, zero, maxq)
    np.testing.assert_array_equal(outc.astype(np.float32), g[f"c{bits}_{tag}_out"])


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("tag", ["f32", "f16"])
def test_qfnb_scale_and_grid(bits, tag):
    g = load_golden("grids")
    W = g["W32"] if tag == "f32" else f16(g["W16"])
    s = O.qfnb_scale(W)
    assert np.float32(s) == g[f"b{bits}_{tag}_scale"][0]
    out = O.quantize_qfnb(W, s, 2 ** bits - 1)
    np.testing.assert_array_equal(out.astype(np.float32), g[f"b{bits}_{tag}_out"])


def test_qfna_sym_per_tensor():
    g = load_golden("grids")
    scale, zero = O.find_params_qfna(g["W32"], 4, perchannel=False, sym=True)
    np.testing.assert_array_equal(scale, g["a4_sym_tensor_scale"])
    np.testing.assert_array_equal(zero, g["a4_sym_tensor_zero"])
    np.testing.assert_array_equal(O.quantize_qfna(g["W32"], scale, zero, 15), g["a4_sym_tensor_out"])


# ------------------------------------------------------------------ packers
def test_pack4_matches_reference_bit_exact():
    g = load_golden("pack")
    q = O.pack_canonical(g["p4_codes"], 4)
    np.testing.assert_array_equal(q, g["p4_qweight"])
    np.testing.assert_array_equal(O.unpack_canonical(q, 4, g["p4_codes"].shape[1]), g["p4_codes"])


def test_pack3_matches_reference_bit_exact():
    g = load_golden("pack")
    np.testing.assert_array_equal(O.pack3(g["p3_codes"]), g["p3_qweight"])


@pytest.mark.parametrize("bits", [2, 4])
def test_pack_roundtrips(bits):
    rng = np.random.default_rng(bits)
    codes = rng.integers(0, 2 ** bits, size=(48, 512), dtype=np.uint8)
    codes[0, :16] = 2 ** bits - 1          # top field set -> negative int32 words
    q = O.pack_canonical(codes, bits)
    assert q.shape == (512 * bits // 32, 48) and q.dtype == np.int32
    np.testing.assert_array_equal(O.unpack_canonical(q, bits, 512), codes)
    s = O.pack_stream(codes, bits)
    assert s.size == 48 * 512 * bits // 32
    np.testing.assert_array_equal(O.unpack_stream(s, bits, 48, 512), codes)


def test_packed_matmul_contract_matches_dense():
    g = load_golden("pack")
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 64)).astype(np.float32)
    y = g["p4_bias"].astype(np.float32).reshape(1, -1).copy()
    O.packed_matmul_c(x, g["p4_qweight"], y, g["p4_scales"], g["p4_zeros"], 4)
    ref = x.astype(np.float64) @ g["p4_W"].astype(np.float64).T + g["p4_bias"]
    np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ butterfly (method.py:16-78)
def test_butterfly_factors():
    g = load_golden("butterfly")
    for n in (2, 6, 40, 64, 192, 768, 2048, 3072, 4096, 7168, 8192, 11008, 28672):
        assert tuple(g[f"factors_{n}"]) == O.butterfly_factors(n)


@pytest.mark.parametrize("n,gname", [(6, "blocked"), (6, "noblock"), (40, "blocked"), (40, "nopermute"),
                                     (64, "blocked"), (64, "noblock"), (64, "nopermute"),
                                     (192, "blocked"), (192, "noblock"), (768, "blocked")])
def test_mul_ortho_butterfly(n, gname):
    g = load_golden("butterfly")
    k = f"n{n}_{gname}"
    Bpp = ([g[k + "_B0"], g[k + "_B1"]], g[k + "_pin"], g[k + "_pout"])
    Y = O.mul_ortho_butterfly(Bpp, g[k + "_X"])
    np.testing.assert_allclose(Y, g[k + "_Y"], rtol=0, atol=2e-6)
    y1 = O.mul_ortho_butterfly(Bpp, g[k + "_X"][:, 0])
    np.testing.assert_allclose(y1, g[k + "_y1"], rtol=0, atol=2e-6)
    # transpose is the inverse (orthogonality) and matches dense U^T
    back = O.mul_ortho_butterfly(Bpp, Y, transpose=True)
    np.testing.assert_allclose(back, g[k + "_X"], rtol=0, atol=1e-5)
    if k + "_dense" in g:
        U = g[k + "_dense"]
        np.testing.assert_allclose(O.mul_ortho_butterfly(Bpp, np.eye(n, dtype=np.float32)), U, atol=2e-6)
        np.testing.assert_allclose(O.mul_ortho_butterfly(Bpp, g[k + "_X"], transpose=True),
                                   U.T @ g[k + "_X"], atol=1e-5)
        np.testing.assert_allclose(U @ U.T, np.eye(n), atol=1e-5)


# ------------------------------------------------------------------ LDLQ (vector_balance.py:155-291,381-422)
def _mismatch(a, b):
    return float(np.mean(a != b))


@pytest.mark.parametrize("bits", [2, 4])
def test_round_ldl_matches_reference(bits):
    g = load_golden("ldlq")
    W, H = g[f"W{bits}"], g["H"]
    got = O.round_ldl(W, H, bits)
    ref = g[f"ldl{bits}"]
    # statistical gate: summation order differs from torch's matvec
    assert _mismatch(got, ref) <= 1e-3
    p_got, p_ref = O.proxy_loss(got - W, H), O.proxy_loss(ref - W, H)
    assert abs(p_got - p_ref) <= 1e-3 * p_ref
    assert abs(p_ref - float(g[f"proxy_ldl{bits}"])) <= 1e-3 * p_ref
    assert p_got < 0.5 * float(g[f"proxy_near{bits}"])            # LDLQ is doing real work
    assert len(np.unique(got)) <= 2 ** bits                       # check_nbits, vector_balance.py:8-11
    # the reference's own blocked variant sits within the same noise band
    assert _mismatch(g[f"ldlblock{bits}"], ref) <= 1e-3


@pytest.mark.parametrize("bits", [2, 4])
def test_round_ldl_unbiased_matches_reference(bits):
    g = load_golden("ldlq")
    got = O.round_ldl(g[f"W{bits}"], g["H"], bits, eta=g[f"eta{bits}"])
    assert _mismatch(got, g[f"ldl{bits}_unbiased"]) <= 1e-3


@pytest.mark.parametrize("bits", [2, 4])
def test_round_ldl_gptqequiv_matches_reference(bits):
    g = load_golden("ldlq")
    got = O.round_ldl_gptqequiv(g[f"W{bits}"], g["H"], bits)
    assert _mismatch(got, g[f"gptqequiv{bits}"]) <= 1e-3


@pytest.mark.parametrize("bits", [2, 4])
def test_kernel_order_variant_is_the_same_algorithm(bits):
    g = load_golden("ldlq")
    W, H = g[f"W{bits}"], g["H"]
    L = O.ldl_factor(H)
    LT = np.ascontiguousarray(np.tril(L, -1).T)
    got = O.round_ldl_kernel_order(W, LT, bits).astype(np.float32)
    ref = g[f"ldlblock{bits}"]
    assert _mismatch(got, ref) <= 1e-3
    p_got, p_ref = O.proxy_loss(got - W, H), O.proxy_loss(ref - W, H)
    assert abs(p_got - p_ref) <= 1e-3 * p_ref


@pytest.mark.parametrize("bits", [2, 4])
@pytest.mark.parametrize("tag", ["f32", "f16"])
@pytest.mark.parametrize("qfn", ["a", "b"])
def test_quantize_weight_vecbal(bits, tag, qfn):
    g = load_golden("ldlq")
    W = g["Wf32"] if tag == "f32" else f16(g["Wf16"])
    scale, zero = O.find_params_qfna(W, bits)
    out, _ = O.quantize_weight_vecbal(W, g["H"], bits, scale, zero, qfn)
    ref = g[f"vecbal_{qfn}{bits}_{tag}_lazy0"]
    assert _mismatch(out.astype(np.float32), ref) <= 2e-3
    ref_lazy = g[f"vecbal_{qfn}{bits}_{tag}_lazy1"]
    assert _mismatch(out.astype(np.float32), ref_lazy) <= 2e-3


def test_counter_example_losses():
    """optq_counter.py:7-31."""
    g = load_golden("counter")
    for n in (64, 256):
        c = 0.01
        H = np.ones((n, n), np.float32) + np.eye(n, dtype=np.float32)
        H[n - 1, n - 1] = 1.0
        H[0, 1:n - 1] += 2 * c
        H[1:n - 1, 0] += 2 * c
        H[0, n - 1] += c
        H[n - 1, 0] += c
        H[0, 0] += 4 * c + n * (c ** 2)
        w = (0.499 * np.ones((n, n)) + 0.002 * (np.arange(n) % 2)).astype(np.float32)
        w_ldl = O.round_ldl_gptqequiv(w, H, 2)
        ldl_loss = O.proxy_loss(w_ldl - w, H)
        near_loss = O.proxy_loss(np.round(w) - w, H)
        assert abs(ldl_loss - float(g[f"n{n}_ldl_loss"])) <= 1e-3 * ldl_loss
        assert abs(near_loss - float(g[f"n{n}_near_loss"])) <= 1e-3 * near_loss


# ------------------------------------------------------------------ QuantMethod (method.py:98-233)
def _bpp(g, case, side):
    return ([g[f"{case}_{side}_B0"], g[f"{case}_{side}_B1"]], g[f"{case}_{side}_pin"], g[f"{case}_{side}_pout"])


def test_hessian_accumulation():
    g = load_golden("method")
    X = f16(g["X"])                                            # [6, 64, d] fp16, one add_batch call per sample
    H = np.zeros((X.shape[-1],) * 2, np.float64)
    n = sum(O.hessian_add_batch(H, x[None]) for x in X)        # method.py:98-120
    assert n == X.shape[0]
    np.testing.assert_allclose(H, g["H64"], rtol=1e-12)
    np.testing.assert_allclose(O.hessian_post_batch(H, n), g["Hraw"], rtol=1e-6)
    H2 = np.zeros_like(H)
    assert O.hessian_add_batch(H2, X) == X.shape[0]            # one 3-D call counts its batch dimension
    np.testing.assert_allclose(H2, H, rtol=1e-12)


@pytest.mark.parametrize("case", ["incoh_w2", "incoh_w4_noblock_lazy"])
def test_preproc_incoherence(case):
    g = load_golden("method")
    U, V = _bpp(g, case, "U"), _bpp(g, case, "V")
    W, H, s = O.preproc(f16(g["W0"]), g["Hraw"], np.float16, True, True, True, U=U, V=V)
    np.testing.assert_allclose(s, g[case + "_scaleWH"], rtol=2e-6)
    np.testing.assert_allclose(H, g[case + "_Hpre"], rtol=0, atol=2e-5 * np.abs(g[case + "_Hpre"]).max())
    ref = f16(g[case + "_Wpre"])
    # W is re-rounded to fp16 after each stage (method.py:155,179): a single 1-ulp flip after the
    # rescale (fp32 summation order of diag(W^T W)) is spread by the dense rotation over the whole
    # matrix, so individual fp16 values may differ by one ulp; the north-star gate is 1e-3 relative.
    a, b = W.astype(np.float32), ref.astype(np.float32)
    assert np.linalg.norm(a - b) / np.linalg.norm(b) <= 1e-3
    assert np.abs(a - b).max() <= 2.0 ** -10 * np.abs(b).max()       # never more than one fp16 ulp
    # dense U stored by the reference equals the structured operator
    np.testing.assert_allclose(O.mul_ortho_butterfly(U, np.eye(W.shape[0], dtype=np.float32)),
                               g[case + "_projU"], atol=2e-6)


def test_preproc_gptqH_only():
    g = load_golden("method")
    W, H, _ = O.preproc(f16(g["W0"]), g["Hraw"], np.float16, False, False, True)
    np.testing.assert_array_equal(W, f16(g["plain_w4_qfna_Wpre"]))
    np.testing.assert_allclose(H, g["plain_w4_qfna_Hpre"], rtol=1e-6)


@pytest.mark.parametrize("case,bits,qfn", [("incoh_w2", 2, "b"), ("incoh_w4_noblock_lazy", 4, "b"),
                                           ("plain_w4_qfna", 4, "a")])
def test_balance_fasterquant_end_to_end(case, bits, qfn):
    """bal.py:21-48: grid map -> LDLQ -> codes->weights(.half()) -> postproc, from the
    reference's own pre-processed (W,H) so only the rounding path is under test."""
    g = load_golden("method")
    Wpre, Hpre = f16(g[case + "_Wpre"]), g[case + "_Hpre"]
    scale, zero = O.find_params_qfna(Wpre, bits)
    wq, _ = O.quantize_weight_vecbal(Wpre, Hpre, bits, scale, zero, qfn)
    incoh = qfn == "b"
    if incoh:
        U, V = _bpp(g, case, "U"), _bpp(g, case, "V")
        Wout, Hpost = O.postproc(wq, Hpre, np.float16, True, True, g[case + "_scaleWH"], U, V)
    else:
        Wout, Hpost = O.postproc(wq, Hpre, np.float16, False, False)
    ref = f16(g[case + "_Wq"]).astype(np.float32)
    got = Wout.astype(np.float32)
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert rel <= 2e-2, rel            # a flipped code moves one weight by a grid step
    assert np.mean(np.abs(got - ref) > 1e-3 * np.abs(ref).max()) <= 5e-3
    # bal.py:48 quirk: error_compute(w, quant_w) uses pre-postproc weights with post-postproc H
    err = O.proxy_loss(Wpre.astype(np.float32) - wq.astype(np.float32), Hpost)
    assert abs(err - float(g[case + "_error"])) <= 2e-2 * abs(float(g[case + "_error"]))
    if case == "incoh_w2":
        np.testing.assert_allclose(Hpost, g[case + "_Hpost"], atol=2e-5 * np.abs(g[case + "_Hpost"]).max())


# ----------------------------------------------------------------------------- greedy passes, LDLQ-RG, GPTQ
def test_greedy_passes_match_reference():
    """oracle greedy_passes (vector_balance.py:182-196) after oracle round_ldl == the reference's round_ldl / round_ldl_block
    with n_greedy_passes = 3 (tests/golden/rounders.npz)."""
    g = load_golden("rounders")
    start = O.round_ldl(g["W2"], g["H"], 2)
    got = O.greedy_passes(g["W2"], start, g["H"], 2, 3)
    assert _mismatch(got, g["ldl2_greedy3"]) <= 2e-3
    assert _mismatch(got, g["ldlblock2_greedy3"]) <= 2e-3


def test_ldlqRG_matches_reference():
    g = load_golden("rounders")
    H, W = g["H"], g["W2"]
    p = np.argsort(np.diag(H), kind="stable")                   # vector_balance.py:147 torch.argsort
    Hp, Wp = H[p][:, p], W[:, p]
    got = np.zeros_like(W)
    got[:, p] = O.greedy_passes(Wp, O.round_ldl(Wp, Hp, 2), Hp, 2, 2)
    assert _mismatch(got, g["ldlqRG2_greedy2"]) <= 5e-3


def test_gptq_round_matches_reference():
    """oracle gptq_round (gptq.py:51-93) against the reference's GPTQ.fasterquant run on CPU (rounders.npz)."""
    g = load_golden("rounders")
    Q, codes = O.gptq_round(g["gptq_W0"], g["gptq_Hdamped"], g["gptq_w4_scale"], g["gptq_w4_zero"], 15)
    assert np.mean(Q != g["gptq_w4_Q"]) <= 2e-3
    dw = (Q - g["gptq_W0"]).astype(np.float64)
    err = float(((dw @ g["gptq_Hdamped"].astype(np.float64)) * dw).sum())
    assert abs(err - float(g["gptq_w4_error"])) <= 1e-3 * float(g["gptq_w4_error"])


def test_oracle_round_ldl_vs_reference_block_sizes():
    """the reference's round_ldl_block at blocksize 32 / 64 / 1000 (tests/golden/ldl_blocksizes.npz, vector_balance.py:218-257) gives the codes
    of its plain round_ldl up to fp32 summation order: the oracle's restatement of :155-199 is within 1e-3 of every one of them"""
    g, gb = load_golden("ldlq"), load_golden("ldl_blocksizes")
    for bits in (2, 4):
        W, H = g[f"W{bits}"], g["H"]
        mine = O.round_ldl(W, H, bits)
        for bsz in (32, 64, 1000):
            ref = gb[f"ldlblock{bits}_bs{bsz}"]
            assert ref.shape == mine.shape and np.mean(mine != ref) <= 1e-3, (bits, bsz)


def test_round_ldl_block_rejects_a_block_size_the_reference_loop_cannot_step():
    import pytest
    import torch
    from quip_amd import vector_balance as vb
    for bad in (0, -128, 64.0, None):
        with pytest.raises(ValueError):
            vb.round_ldl_block(torch.zeros(4, 16), torch.eye(16), 2, blocksize=bad, n_greedy_passes=0)
