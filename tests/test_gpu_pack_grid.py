"""-m gpu: K1 (pack/unpack) and K5 (grid map) through the C ABI vs the oracle and the reference goldens."""
import numpy as np
import pytest
import torch

from conftest import load_golden, f16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from quip_amd import ops as o
    return o


@pytest.fixture(scope="module")
def O():
    from oracle import quip_oracle
    return quip_oracle


# --------------------------------------------------------------------------------------------- K1
@pytest.mark.parametrize("bits", [2, 4])
@pytest.mark.parametrize("m,d", [(16, 256), (48, 512), (40, 192), (33, 48), (256, 4096)])
def test_pack_canonical_bit_exact(ops, O, bits, m, d):
    rng = np.random.default_rng(m * d + bits)
    codes = rng.integers(0, 2 ** bits, size=(m, d), dtype=np.uint8)
    codes[0, :32] = 2 ** bits - 1                                   # bit 31 set -> negative int32 words
    q = ops.pack(torch.from_numpy(codes).to(DEV), bits, ops.LAYOUT_CANONICAL)
    assert q.shape == (d * bits // 32, m) and q.dtype == torch.int32
    np.testing.assert_array_equal(q.cpu().numpy(), O.pack_canonical(codes, bits))
    back = ops.unpack(q, bits, ops.LAYOUT_CANONICAL, m, d)
    np.testing.assert_array_equal(back.cpu().numpy(), codes)


def test_pack4_reproduces_reference_qweight(ops):
    g = load_golden("pack")                                          # zeroShot/models/quant.py:190-199
    q = ops.pack(torch.from_numpy(g["p4_codes"]).to(DEV), 4, ops.LAYOUT_CANONICAL)
    np.testing.assert_array_equal(q.cpu().numpy(), g["p4_qweight"])


@pytest.mark.parametrize("bits", [2, 4])
@pytest.mark.parametrize("m,d", [(16, 512), (48, 1024), (4096, 4096)])
def test_pack_stream_is_the_declared_permutation(ops, O, bits, m, d):
    rng = np.random.default_rng(7 + bits)
    codes = rng.integers(0, 2 ** bits, size=(m, d), dtype=np.uint8)
    cd = torch.from_numpy(codes).to(DEV)
    s = ops.pack(cd, bits, ops.LAYOUT_STREAM)
    if m * d <= 1 << 20:
        np.testing.assert_array_equal(s.cpu().numpy(), O.pack_stream(codes, bits))
    # round trips: stream -> codes, and canonical -> codes -> stream == direct stream (permutation proof)
    assert torch.equal(ops.unpack(s, bits, ops.LAYOUT_STREAM, m, d), cd)
    c = ops.pack(cd, bits, ops.LAYOUT_CANONICAL)
    s2 = ops.pack(ops.unpack(c, bits, ops.LAYOUT_CANONICAL, m, d), bits, ops.LAYOUT_STREAM)
    assert torch.equal(s, s2)
    # same multiset of bits: population count is invariant under the permutation
    pc = lambda t: int(np.unpackbits(t.cpu().numpy().view(np.uint8)).sum())
    if m * d <= 1 << 20:
        assert pc(s) == pc(c)


def test_pack_empty_and_errors(ops):
    z = ops.pack(torch.zeros((0, 256), dtype=torch.uint8, device=DEV), 2, ops.LAYOUT_CANONICAL)
    assert z.numel() == 0
    from quip_amd._lib import QuipAmdError
    with pytest.raises(QuipAmdError):
        ops.pack(torch.zeros((16, 24), dtype=torch.uint8, device=DEV), 2, ops.LAYOUT_CANONICAL)   # d % 16
    with pytest.raises(QuipAmdError):
        ops.pack(torch.zeros((8, 256), dtype=torch.uint8, device=DEV), 2, ops.LAYOUT_STREAM)      # m % 16


# --------------------------------------------------------------------------------------------- K5
@pytest.mark.parametrize("tag", ["f32", "f16"])
def test_qfnb_scale_matches_reference(ops, O, tag):
    g = load_golden("grids")
    W = g["W32"] if tag == "f32" else f16(g["W16"])
    s = ops.qfnb_scale(torch.from_numpy(W.copy()).to(DEV)).cpu().numpy()[0]
    assert s == g[f"b2_{tag}_scale"][0]                              # quant.py:150, bit-exact
    rng = np.random.default_rng(3)
    big = (0.02 * rng.standard_normal((512, 1024))).astype(W.dtype)
    assert ops.qfnb_scale(torch.from_numpy(big).to(DEV)).cpu().numpy()[0] == np.float32(O.qfnb_scale(big))


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("tag", ["f32", "f16"])
def test_quantize_matches_reference_grids(ops, O, bits, tag):
    g = load_golden("grids")
    W = g["W32"] if tag == "f32" else f16(g["W16"])
    Wd = torch.from_numpy(W.copy()).to(DEV)
    maxq = 2 ** bits - 1
    scale, zero = g[f"a{bits}_{tag}_scale"], g[f"a{bits}_{tag}_zero"]
    for qfn in ("a", "c"):
        out = ops.quantize(Wd, qfn, torch.from_numpy(scale), torch.from_numpy(zero), maxq)
        ref = g[f"{qfn}{bits}_{tag}_out"]                            # fp32 (promoted) in the reference
        np.testing.assert_array_equal(out.float().cpu().numpy(), ref.astype(W.dtype).astype(np.float32))
    s = ops.qfnb_scale(Wd)
    out, codes = ops.quantize(Wd, "b", s, None, maxq, want_codes=True)
    np.testing.assert_array_equal(out.float().cpu().numpy(), g[f"b{bits}_{tag}_out"])
    assert int(codes.max()) <= maxq


@pytest.mark.parametrize("tag", ["f32", "f16"])
@pytest.mark.parametrize("bits", [2, 4])
def test_gridmap_and_codes_to_weight(ops, O, tag, bits):
    g = load_golden("ldlq")
    W = g["Wf32"] if tag == "f32" else f16(g["Wf16"])
    Wd = torch.from_numpy(W.copy()).to(DEV)
    maxq = 2 ** bits - 1
    s_ref, wr_ref = O.gridmap_qfnb(W, maxq)
    s = ops.qfnb_scale(Wd)
    assert s.cpu().numpy()[0] == np.float32(s_ref)
    wr = ops.gridmap(Wd, "b", s, None, maxq)
    np.testing.assert_array_equal(wr.cpu().numpy(), wr_ref.astype(np.float32))
    scale, zero = O.find_params_qfna(W, bits)
    wa = ops.gridmap(Wd, "a", torch.from_numpy(scale), torch.from_numpy(zero), maxq)
    np.testing.assert_array_equal(wa.cpu().numpy(), O.gridmap_qfna(W, scale, zero, maxq))
    rng = np.random.default_rng(0)
    codes = rng.integers(0, maxq + 1, size=W.shape, dtype=np.uint8)
    cd = torch.from_numpy(codes).to(DEV)
    wb = ops.codes_to_weight(cd, "b", s, None, maxq)
    np.testing.assert_array_equal(wb.cpu().numpy(), O.codes_to_weight_qfnb(codes, s_ref, maxq))
    wq = ops.codes_to_weight(cd, "a", torch.from_numpy(scale), torch.from_numpy(zero), maxq)
    np.testing.assert_array_equal(wq.cpu().numpy(), O.codes_to_weight_qfna(codes, scale, zero))


@pytest.mark.parametrize("bits", [2, 4])
def test_stream_packing_of_row_chunks_concatenates(ops, bits):
    """quip_amd/shard.py gathers per-rank STREAM-packed code chunks: packing row chunks (multiples of 16 rows)
    separately and concatenating must equal packing the whole matrix (STREAM is row-tile-major)."""
    m, d = 208, 1024
    g = torch.Generator().manual_seed(bits)
    codes = torch.randint(0, 2 ** bits, (m, d), generator=g, dtype=torch.uint8).to(DEV)
    whole = ops.pack(codes, bits, ops.LAYOUT_STREAM)
    for cuts in ([0, 112, 208], [0, 16, 48, 208], [0, 64, 128, 192, 208]):
        parts = [ops.pack(codes[a:b].contiguous(), bits, ops.LAYOUT_STREAM) for a, b in zip(cuts[:-1], cuts[1:])]
        assert torch.equal(torch.cat(parts), whole)
    from quip_amd import shard
    assert torch.equal(shard._unpack_all(torch.cat(parts), bits, m, d), codes)


def test_degenerate_and_rejected_calls(ops):
    """empty batches / zero rows are no-ops, unsupported shapes fail loudly (no silent fallback)."""
    from quip_amd import method
    from quip_amd._lib import QuipAmdError
    codes = torch.randint(0, 4, (32, 512), dtype=torch.uint8).to(DEV)
    qs = ops.pack(codes, 2, ops.LAYOUT_STREAM)
    sc = torch.tensor([0.05])
    y = ops.dequant_gemm(torch.empty(0, 512, dtype=torch.bfloat16, device=DEV), qs, 2, 'b', sc, None, None, m=32)
    assert y.shape == (0, 32)
    with pytest.raises(QuipAmdError):                                  # d not a multiple of the 2-bit chunk (256)
        ops.dequant_gemm(torch.zeros(4, 384, dtype=torch.bfloat16, device=DEV), qs, 2, 'b', sc, None, None, m=32)
    with pytest.raises(AssertionError):                                # fp32 activations must be narrowed by the caller
        ops.dequant_gemm(torch.zeros(4, 512, dtype=torch.float32, device=DEV), qs, 2, 'b', sc, None, None, m=32)
    with pytest.raises(ValueError):                                    # per-column grids are refused, not read out of bounds
        ops.quantize(torch.zeros(32, 512, device=DEV), 'a', torch.ones(1, 512), torch.zeros(1, 512), 3)
    with pytest.raises(ValueError):
        ops.dequant_gemm(torch.zeros(4, 512, dtype=torch.bfloat16, device=DEV), qs, 2, 'a', torch.ones(7), torch.zeros(7), None, m=32)
    # a scalar per-row grid is expanded
    w = torch.randn(32, 512, device=DEV)
    assert torch.equal(ops.quantize(w, 'a', torch.tensor(0.5), torch.tensor(2.0), 3),
                       ops.quantize(w, 'a', torch.full((32,), 0.5), torch.full((32, 1), 2.0), 3))
    np.random.seed(0)
    torch.manual_seed(0)
    op = ops.OrthoOp(method.gen_rand_ortho_butterfly(40), DEV)
    assert op.apply_rows(torch.empty(0, 40, device=DEV)).shape == (0, 40)
    H = torch.eye(64, device=DEV)
    LT = ops.unit_lower_t(torch.linalg.cholesky(H))
    assert ops.ldlq_round(torch.empty(0, 64, device=DEV), LT, 2).shape == (0, 64)
    with pytest.raises(QuipAmdError):                                  # d must be a multiple of 16
        ops.ldlq_round(torch.zeros(4, 40, device=DEV), torch.zeros(40, 40, device=DEV), 2)
    # identity Hessian: LDLQ degenerates to round-to-nearest (no error feedback)
    W = (torch.rand(8, 64, device=DEV) * 3.6 - 0.3).clamp(0, 3)
    assert torch.equal(ops.ldlq_round(W, LT, 2).float(), torch.clamp(torch.floor(W + 0.5), 0, 3))
