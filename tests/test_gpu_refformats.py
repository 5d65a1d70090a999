"""-m gpu: the reference's OWN packed formats on the GPU (VERDICT r1 missing #3 / next #8):
  * Quant3Linear.pack's 32-codes-in-3-words rule (quant.py:192-220) as quipamd_pack / quipamd_unpack with bits = 3,
    bit-exact vs the oracle restatement (itself pinned to the reference's output in tests/golden/pack.npz);
  * CANONICAL -> STREAM repack on the device for 2 / 3 / 4 bits, bit-exact vs packing the codes directly;
  * quipamd_vecquant{3,4}matmul: the reference extension's call by name and argument meaning (quant.py:229,
    zeroShot/models/quant.py:207), vs fp64 of the formula  mul += sum (scales q - zeros) vec;
  * a checkpoint in the reference's Quant3Linear format loads into quip_amd.quant.Quant3Linear."""
import numpy as np
import pytest
import torch

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


@pytest.mark.parametrize("m,d", [(16, 1024), (48, 2048), (5, 32), (40, 96)])
def test_three_bit_canonical_pack_is_the_reference_rule(ops, O, m, d):
    rng = np.random.default_rng(m + d)
    codes = rng.integers(0, 8, size=(m, d), dtype=np.uint8)
    want = O.pack3(codes) if d % 1024 == 0 else None
    got = ops.pack(torch.from_numpy(codes).to(DEV), 3, ops.LAYOUT_CANONICAL)
    assert got.shape == (d // 32 * 3, m) and got.dtype == torch.int32
    if want is not None:
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    back = ops.unpack(got, 3, ops.LAYOUT_CANONICAL, m, d)
    np.testing.assert_array_equal(back.cpu().numpy(), codes)


@pytest.mark.parametrize("bits,m,d", [(2, 32, 512), (4, 48, 256), (3, 64, 1024), (3, 16, 128), (2, 4096, 4096), (4, 1024, 11008)])
def test_repack_canonical_to_stream_is_bit_exact(ops, bits, m, d):
    g = torch.Generator().manual_seed(bits + m)
    codes = torch.randint(0, 2 ** bits, (m, d), generator=g, dtype=torch.uint8).to(DEV)
    canon = ops.pack(codes, bits, ops.LAYOUT_CANONICAL)
    want = ops.pack(codes, bits, ops.LAYOUT_STREAM)
    got = ops.repack_canonical_to_stream(canon, bits, m, d)
    assert torch.equal(got, want)


@pytest.mark.parametrize("bits,m,d", [(4, 64, 256), (4, 4096, 4096), (3, 48, 1024), (3, 2048, 2048)])
def test_vecquant_matmul_by_the_references_signature(ops, O, bits, m, d):
    rng = np.random.default_rng(bits * 7 + m)
    maxq = 2 ** bits - 1
    W = (0.02 * rng.standard_normal((m, d))).astype(np.float32)
    scale, zero = O.find_params_qfna(W, bits)
    codes = np.clip(np.round(W / scale) + zero, 0, maxq).astype(np.uint8)
    vec = rng.standard_normal(d).astype(np.float32)
    bias = rng.standard_normal(m).astype(np.float32)
    scales = scale.reshape(m, 1).astype(np.float32)
    zeros = (zero.reshape(m, 1) * scales).astype(np.float32)            # quant.py:186: self.zeros = zeros * scales
    want = bias.astype(np.float64) + (scales.astype(np.float64) * codes - zeros.astype(np.float64)) @ vec.astype(np.float64)
    mat = ops.pack(torch.from_numpy(codes).to(DEV), bits, ops.LAYOUT_CANONICAL)     # the reference's packing
    mul = torch.from_numpy(bias.copy()).to(DEV)                         # y = self.bias.clone(); vecquant(x, qweight, y, ...)
    ops.vecquantmatmul(bits, torch.from_numpy(vec).to(DEV), mat, mul, torch.from_numpy(scales).to(DEV), torch.from_numpy(zeros).to(DEV))
    got = mul.cpu().numpy().astype(np.float64)
    assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-4     # x kept to 2^-16 (hi + lo bf16 terms), fp32 accumulate
    # second token, same layer: the repack inside the library is skipped (workspace remembered per `mat`), same answer bit for bit
    mul2 = torch.from_numpy(bias.copy()).to(DEV)
    ops.vecquantmatmul(bits, torch.from_numpy(vec).to(DEV), mat, mul2, torch.from_numpy(scales).to(DEV), torch.from_numpy(zeros).to(DEV))
    assert torch.equal(mul2, mul)
    # the layer is re-quantised IN PLACE: the remembered repack must not be used
    codes2 = (maxq - codes).astype(np.uint8)
    mat.copy_(ops.pack(torch.from_numpy(codes2).to(DEV), bits, ops.LAYOUT_CANONICAL))
    mul3 = torch.from_numpy(bias.copy()).to(DEV)
    ops.vecquantmatmul(bits, torch.from_numpy(vec).to(DEV), mat, mul3, torch.from_numpy(scales).to(DEV), torch.from_numpy(zeros).to(DEV))
    want3 = bias.astype(np.float64) + (scales.astype(np.float64) * codes2 - zeros.astype(np.float64)) @ vec.astype(np.float64)
    assert np.linalg.norm(mul3.cpu().numpy() - want3) / np.linalg.norm(want3) <= 1e-4


def test_vecquant_c_entry_is_stateless_unless_prepared(ops, O):
    """the C entry point called with the SAME workspace / mat addresses after the contents of `mat` changed must use the new
    contents (it used to remember the repack per address).  With quipamd_vecquant_prepare the caller opts into the remembered repack --
    and then a silent change of `mat` is the caller's to announce (prepare again / invalidate)."""
    import ctypes
    from quip_amd import _lib
    bits, m, d = 4, 64, 512
    rng = np.random.default_rng(11)
    scales = torch.full((m,), 0.01, device=DEV)
    zeros = torch.full((m,), 0.08, device=DEV)                          # zero * scale, zero = 8
    vec = torch.from_numpy(rng.standard_normal(d).astype(np.float32)).to(DEV)
    c1 = rng.integers(0, 16, (m, d)).astype(np.uint8)
    c2 = (15 - c1).astype(np.uint8)
    mat = ops.pack(torch.from_numpy(c1).to(DEV), bits, ops.LAYOUT_CANONICAL)
    lib = _lib.load()
    nbytes = int(lib.quipamd_vecquant_workspace_bytes(bits, m, d))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    lib.quipamd_vecquant_invalidate(ctypes.c_void_p(ws.data_ptr()))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())

    def run():
        mul = torch.zeros(m, device=DEV)
        _lib.call("quipamd_vecquant4matmul", vp(vec), vp(mat), vp(mul), vp(scales), vp(zeros), m, d, vp(ws), nbytes, st)
        return mul.cpu().numpy().astype(np.float64)

    def want(c):
        return (0.01 * c.astype(np.float64) - 0.08) @ vec.cpu().numpy().astype(np.float64)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(run(), want(c1)) <= 1e-4
    mat.data.copy_(ops.pack(torch.from_numpy(c2).to(DEV), bits, ops.LAYOUT_CANONICAL))      # through .data: no version bump, same address
    assert rel(run(), want(c2)) <= 1e-4                                  # stateless: the new contents
    _lib.call("quipamd_vecquant_prepare", bits, vp(mat), m, d, vp(ws), nbytes, st)          # opt in
    assert rel(run(), want(c2)) <= 1e-4
    mat.data.copy_(ops.pack(torch.from_numpy(c1).to(DEV), bits, ops.LAYOUT_CANONICAL))
    assert rel(run(), want(c2)) <= 1e-4                                  # prepared: the repack of the prepare call is what runs (documented)
    lib.quipamd_vecquant_invalidate(vp(ws))
    assert rel(run(), want(c1)) <= 1e-4
    # the Python wrapper's guard is torch's version counter; a write through .data needs ops.vecquant_forget
    mul = torch.zeros(m, device=DEV)
    ops.vecquantmatmul(bits, vec, mat, mul, scales, zeros)
    mat.data.copy_(ops.pack(torch.from_numpy(c2).to(DEV), bits, ops.LAYOUT_CANONICAL))
    ops.vecquant_forget(mat)
    mul = torch.zeros(m, device=DEV)
    ops.vecquantmatmul(bits, vec, mat, mul, scales, zeros)
    assert rel(mul.cpu().numpy().astype(np.float64), want(c2)) <= 1e-4


def test_reference_quant3linear_checkpoint_loads(ops, O):
    """state dict with the reference Quant3Linear's buffers (quant.py:176-197) -> quip_amd.quant.Quant3Linear, via make_quant3 +
    load_state_dict like opt.py:350-381 load_quant3 does."""
    from quip_amd import quant as Q
    m, d = 64, 1024
    rng = np.random.default_rng(3)
    W = (0.02 * rng.standard_normal((m, d))).astype(np.float32)
    scale, zero = O.find_params_qfna(W, 3)
    Wq = O.quantize_qfna(W, scale, zero, 7)
    codes = np.clip(np.round(W / scale) + zero, 0, 7).astype(np.uint8)
    bias = rng.standard_normal(m).astype(np.float32)
    ref_state = {"0.qweight": torch.from_numpy(O.pack3(codes)), "0.scales": torch.from_numpy(scale.reshape(m, 1).astype(np.float32)),
                 "0.zeros": torch.from_numpy((zero.reshape(m, 1) * scale.reshape(m, 1)).astype(np.float32)), "0.bias": torch.from_numpy(bias)}
    holder = torch.nn.Sequential(torch.nn.Linear(d, m)).to(DEV)
    Q.make_quant3(holder, ["0"])
    holder = holder.to(DEV)
    holder.load_state_dict(ref_state)
    np.testing.assert_array_equal(ops.unpack(holder[0].qweight, 3, ops.LAYOUT_STREAM, m, d).cpu().numpy(), codes)
    x = torch.from_numpy(rng.standard_normal((3, d)).astype(np.float32)).to(DEV).half()
    got = holder[0](x).double().cpu().numpy()
    want = x.double().cpu().numpy() @ Wq.astype(np.float64).T + bias
    assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 2e-3
    # the reference's own order (opt.py:350-381 load_quant3): make_quant3 and load_state_dict on the CPU, THEN .to(dev) -- the
    # canonical words wait on the module and are repacked by the move; forward before the move raises (no CPU fallback)
    cpu_holder = torch.nn.Sequential(torch.nn.Linear(d, m))
    Q.make_quant3(cpu_holder, ["0"])
    res = cpu_holder.load_state_dict({k: v.clone() for k, v in ref_state.items()})
    assert not res.missing_keys and not res.unexpected_keys
    with pytest.raises(RuntimeError):
        cpu_holder[0](x.cpu())
    cpu_holder = cpu_holder.to(DEV)
    assert torch.equal(cpu_holder[0].qweight, holder[0].qweight) and torch.equal(cpu_holder[0].zeros, holder[0].zeros)
    assert torch.equal(cpu_holder[0](x), holder[0](x))
    # a record of the wrong shape is refused BEFORE the module is touched
    bad = {k: v.clone() for k, v in ref_state.items()}
    bad["0.scales"] = bad["0.scales"][:-1]
    fresh = torch.nn.Sequential(torch.nn.Linear(d, m)).to(DEV)
    Q.make_quant3(fresh, ["0"])
    fresh = fresh.to(DEV)
    with pytest.raises(RuntimeError):
        fresh.load_state_dict(bad)
    assert fresh[0].bias is None
    # and the 4-bit format of zeroShot/models/quant.py
    sc4, z4 = O.find_params_qfna(W, 4)
    c4 = np.clip(np.round(W / sc4) + z4, 0, 15).astype(np.uint8)
    q4 = Q.from_reference_packed(torch.from_numpy(O.pack_canonical(c4, 4)), torch.from_numpy(sc4.reshape(m, 1).astype(np.float32)),
                                 torch.from_numpy((z4.reshape(m, 1) * sc4.reshape(m, 1)).astype(np.float32)), torch.from_numpy(bias), 4)
    got4 = q4(x).double().cpu().numpy()
    want4 = x.double().cpu().numpy() @ O.quantize_qfna(W, sc4, z4, 15).astype(np.float64).T + bias
    assert np.linalg.norm(got4 - want4) / np.linalg.norm(want4) <= 2e-3
