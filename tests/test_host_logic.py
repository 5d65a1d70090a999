"""CPU: host-side logic of the surface modules that needs no kernel (factorisation, RNG parity of the operator
generators, module structure)."""
import numpy as np
import torch

from conftest import load_golden


def test_butterfly_factors_match_reference():
    from quip_amd.method import butterfly_factors
    g = load_golden("butterfly")
    for n in (2, 6, 40, 64, 192, 768, 2048, 3072, 4096, 7168, 8192, 11008, 28672):
        assert butterfly_factors(n) == tuple(g[f"factors_{n}"])


def test_generators_consume_rng_like_the_reference():
    """seeding numpy + torch as the golden script did must reproduce the reference's operators bit for bit."""
    from quip_amd import method as M
    g = load_golden("butterfly")
    gens = {"blocked": M.gen_rand_ortho_butterfly, "noblock": M.gen_rand_ortho_butterfly_noblock,
            "nopermute": M.gen_rand_ortho_butterfly_nopermute}
    for n in (6, 40, 64, 192):
        for name, gen in gens.items():
            np.random.seed(100 + n)
            torch.manual_seed(100 + n)
            B, p_in, p_out = gen(n)
            k = f"n{n}_{name}"
            assert B[0].dtype == torch.float32
            np.testing.assert_array_equal(B[0].numpy(), g[k + "_B0"])
            np.testing.assert_array_equal(B[1].numpy(), g[k + "_B1"])
            np.testing.assert_array_equal(p_in.numpy(), g[k + "_pin"])
            np.testing.assert_array_equal(p_out.numpy(), g[k + "_pout"])


def test_gpu_sampler_is_scipys_algorithm_on_the_same_stream():
    """method._special_ortho_group_gpu (run here on the CPU device) restates scipy.stats.special_ortho_group.rvs:
    same numpy draws in the same order, same matrices up to fp64 summation order, same shape rule for size 1."""
    import scipy.stats
    from quip_amd import method as M
    for p, m in [(3, 1), (5, 4), (43, 2), (64, 8)]:
        np.random.seed(7)
        want = scipy.stats.special_ortho_group.rvs(p, size=m)
        after_ref = np.random.normal()
        np.random.seed(7)
        got = M._special_ortho_group_gpu(p, m, torch.device("cpu")).numpy()
        after = np.random.normal()
        assert got.shape == want.shape and after == after_ref
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-13)


def test_surface_names_exist():
    import quip_amd.quant as q, quip_amd.method as m, quip_amd.vector_balance as vb
    import quip_amd.bal as bal, quip_amd.gptq as gptq, quip_amd.near as near, quip_amd.modelutils as mu
    for name in ("quantize_qfna", "quantize_qfnb", "quantize_qfnc", "Quantizer", "QuantLinear", "make_quant", "Quant3Linear",
                 "make_quant3"):
        assert hasattr(q, name)
    for name in ("butterfly_factors", "gen_rand_orthos", "gen_rand_ortho_butterfly", "gen_rand_ortho_butterfly_noblock",
                 "gen_rand_ortho_butterfly_nopermute", "mul_ortho_butterfly", "rand_ortho_butterfly", "QuantMethod"):
        assert hasattr(m, name)
    for name in ("round_ldl", "round_ldl_block", "round_ldl_gptqequiv", "round_sorted_ldlqRG", "round_sorted_ldlqRG_block", "quantize_weight_vecbal",
                 "check_nbits"):
        assert hasattr(vb, name)
    holder = torch.nn.Sequential(torch.nn.Linear(512, 32))
    q.make_quant3(holder, ["0"])                                  # module swap needs no GPU (quant.py:236-246)
    assert isinstance(holder[0], q.Quant3Linear) and holder[0].qweight.numel() == 512 * 32 * 4 // 32
    assert issubclass(bal.Balance, m.QuantMethod) and issubclass(gptq.GPTQ, m.QuantMethod)
    assert issubclass(near.Nearest, m.QuantMethod)
    assert str(mu.DEV) == "cuda:0"
    layer = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Sequential(torch.nn.ReLU(), torch.nn.Linear(4, 2)))
    assert list(mu.find_layers(layer)) == ["0", "1.1"]


def test_quantizer_configure_and_cpu_find_params():
    """find_params_qfna is O(m) torch glue and runs wherever the tensor lives; the grid kernels do not."""
    from quip_amd.quant import Quantizer
    g = load_golden("grids")
    q = Quantizer()
    q.configure(4, perchannel=True, sym=False, qfn='a', mse=False)
    assert int(q.maxq) == 15
    q.find_params(torch.from_numpy(g["W32"]), weight=True)
    np.testing.assert_array_equal(q.scale.numpy(), g["a4_f32_scale"])
    np.testing.assert_array_equal(q.zero.numpy(), g["a4_f32_zero"])


def test_share_hessian_between_layers_with_the_same_input():
    """q/k/v see one input: a follower takes the leader's Hessian instead of accumulating its own copy."""
    from quip_amd.method import QuantMethod
    torch.manual_seed(0)
    lq, lk, lv = (torch.nn.Linear(24, 8) for _ in range(3))
    X = torch.randn(4, 10, 24)
    ref = QuantMethod(lk)
    lead, f1, f2 = QuantMethod(lq), QuantMethod(lk), QuantMethod(lv)
    f1.share_hessian_from(lead)
    f2.share_hessian_from(lead)
    for j in range(4):
        for qm in (ref, lead, f1, f2):
            qm.add_batch(X[j:j + 1], None)
    ref.post_batch()
    f1.post_batch()                                   # before the leader: finishes from the fp64 accumulator
    lead.post_batch()
    f2.post_batch()                                   # after the leader: copies the finished matrix
    for qm in (lead, f1, f2):
        assert qm.nsamples == 4 and qm.H.dtype == torch.float32
        assert torch.equal(qm.H, ref.H)
    assert f1.H.data_ptr() != lead.H.data_ptr() and f2.H.data_ptr() != lead.H.data_ptr()


# ---- quip_amd.decode: the host side of the decode engine (no kernel runs here) -------------------------------------------------
def _tiny_hf(arch, **kw):
    if arch == "opt":
        from transformers import OPTConfig, OPTForCausalLM
        cfg = dict(hidden_size=64, ffn_dim=128, num_hidden_layers=2, num_attention_heads=4, word_embed_proj_dim=64, vocab_size=96,
                   max_position_embeddings=16)
        cfg.update(kw)
        return OPTForCausalLM(OPTConfig(**cfg))
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = dict(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4, vocab_size=96,
               max_position_embeddings=16, tie_word_embeddings=False)
    cfg.update(kw)
    return LlamaForCausalLM(LlamaConfig(**cfg))


def test_decoder_from_hf_shares_the_models_modules():
    """from_hf builds its block records over the Hugging Face model's own modules (no copies): what make_quant swapped in is
    what the engine runs (reference: benchmark() drives the model object it was handed, opt.py:431-482)"""
    from quip_amd import decode
    m = _tiny_hf("opt")
    d = decode.decoder_from_hf(m)
    assert d.arch == "opt" and d.layers_n == 2 and d.heads == 4 and d.h == 64
    assert d.tok is m.model.decoder.embed_tokens and d.blocks[1].fc2 is m.model.decoder.layers[1].fc2
    assert d.head_weight.data_ptr() == m.model.decoder.embed_tokens.weight.data_ptr()        # tied head
    assert not d.packed() and decode.best_mode(d, 1, torch.float16) == "plain"
    m = _tiny_hf("llama")
    d = decode.decoder_from_hf(m, max_len=16)
    assert d.arch == "llama" and d.blocks[0].down_proj is m.model.layers[0].mlp.down_proj and d.head_weight is m.lm_head.weight
    assert d.cos.shape == (16, 16) and d.cos.dtype == torch.float32
    inv = m.model.rotary_emb.inv_freq.float()
    np.testing.assert_allclose(d.sin[3, :8].numpy(), torch.sin(3 * inv).numpy(), rtol=1e-6)
    d.half()                                                  # a dtype change must not narrow the rotary tables
    assert d.cos.dtype == torch.float32


def test_decode_engine_refuses_what_it_cannot_serve():
    import pytest
    from quip_amd import decode
    with pytest.raises(NotImplementedError):
        decode.decoder_from_hf(_tiny_hf("opt", word_embed_proj_dim=32))            # projected embeddings (opt-350m's form)
    with pytest.raises(NotImplementedError):
        decode.decoder_from_hf(_tiny_hf("opt", do_layer_norm_before=False))
    with pytest.raises(NotImplementedError):
        decode.decoder_from_hf(_tiny_hf("llama", num_key_value_heads=2))           # grouped-query attention
    with pytest.raises(RuntimeError):
        decode.DecodeEngine.from_hf(_tiny_hf("opt").half(), device="cpu")          # no CPU path


def test_collect_packed_patches_free_only_while_active():
    from quip_amd import decode, method
    orig = method.QuantMethod.free
    lin = torch.nn.Linear(8, 8)
    with decode.collect_packed() as packed:
        assert method.QuantMethod.free is not orig
        m = method.QuantMethod(lin)
        m.free()                                              # no integer codes on this method: skipped, free() still runs
        assert m.H is None and packed.layers == []
    assert method.QuantMethod.free is orig
    assert packed.named(torch.nn.Sequential(lin)) == {}


def test_lt_queue_keeps_its_hessians_alive():
    """a queued LT is matched by (address, shape) of its H; the queue must hold H, or a freed H's address can be handed to a
    different Hessian of the same shape (ldlqRG's permuted copy) that then matches the stale entry"""
    import gc
    import weakref
    from quip_amd import shard
    h = shard.ShardedLDLQ()
    H1 = torch.randn(64, 64)
    ref = weakref.ref(H1)
    key = shard.h_key(H1)
    h.queue_LTs([(H1, torch.zeros(64, 64))])
    del H1
    gc.collect()
    assert ref() is not None                                   # still alive: held by the queue
    others = [torch.randn(64, 64) for _ in range(32)]          # none of them can sit at the queued address
    assert all(shard.h_key(t) != key for t in others)
    assert h.queued(key) and not h.queued(shard.h_key(others[0]))


def _draw_pairs(M, shapes, extra):
    gen = M._GENERATORS[extra]
    return [(gen(r), gen(c)) for r, c in shapes]


def _same_op(a, b):
    (Ba, pia, poa), (Bb, pib, pob) = a, b
    return all(torch.equal(x, y) for x, y in zip(Ba, Bb)) and torch.equal(pia, pib) and torch.equal(poa, pob)


def test_operator_prefetcher_draws_the_same_operators_from_the_same_streams():
    """method.OPERATOR_PREFETCH: the host thread makes exactly the draws preproc would make, in the same order, on numpy's and torch's
    global generators -- a seeded run gives the same operators with and without it, and leaves both streams at the same position"""
    from quip_amd import method as M
    shapes = [(48, 48), (48, 40), (40, 192), (6, 48)]
    for extra in (0, 1):
        np.random.seed(7)
        torch.manual_seed(7)
        want = _draw_pairs(M, shapes, extra)
        tail_np, tail_t = np.random.normal(size=3), torch.randperm(11)
        np.random.seed(7)
        torch.manual_seed(7)
        pf = M._OperatorPrefetcher()
        pf.note_flags(True, extra)
        for i, (r, c) in enumerate(shapes):
            pf.request(i, r, c)
        got = [pf.take(i, extra) for i in range(len(shapes))]
        assert all(g is not None for g in got) and pf.stats["prefetched"] == len(shapes)
        for (gu, gv), (wu, wv) in zip(got, want):
            assert _same_op(gu, wu) and _same_op(gv, wv)
        pf.shutdown()
        np.testing.assert_array_equal(np.random.normal(size=3), tail_np)
        assert torch.equal(torch.randperm(11), tail_t)


def test_operator_prefetcher_rewinds_when_its_guess_was_wrong():
    """a preproc that does not find its pair at the head of the queue (another generator, another call order, no projection at all) gets the
    streams back exactly where a run without the prefetcher would have them"""
    from quip_amd import method as M
    shapes = [(48, 48), (40, 48), (48, 40)]
    np.random.seed(3)
    torch.manual_seed(3)
    first = _draw_pairs(M, shapes[:1], 0)
    rest_as_kron = _draw_pairs(M, shapes[1:], 1)                 # the run we must reproduce: generator 0 once, then generator 1
    np.random.seed(3)
    torch.manual_seed(3)
    pf = M._OperatorPrefetcher()
    pf.note_flags(True, 0)
    for i, (r, c) in enumerate(shapes):
        pf.request(i, r, c)
    u, v = pf.take(0, 0)
    assert _same_op(u, first[0][0]) and _same_op(v, first[0][1])
    assert pf.take(1, 1) is None                                 # drawn (or being drawn) with generator 0: rewind, caller samples
    got = _draw_pairs(M, shapes[1:], 1)
    for (gu, gv), (wu, wv) in zip(got, rest_as_kron):
        assert _same_op(gu, wu) and _same_op(gv, wv)
    assert pf.stats["sync"] == 1
    # out-of-order take and a drain behave the same way
    np.random.seed(5)
    torch.manual_seed(5)
    want = _draw_pairs(M, shapes[:2], 0)
    np.random.seed(5)
    torch.manual_seed(5)
    pf.note_flags(True, 0)
    for i, (r, c) in enumerate(shapes):
        pf.request(10 + i, r, c)
    assert pf.take(11, 0) is None                                # 10 is at the head, not 11
    got = _draw_pairs(M, shapes[:2], 0)
    for (gu, gv), (wu, wv) in zip(got, want):
        assert _same_op(gu, wu) and _same_op(gv, wv)
    pf.shutdown()


def test_identical_inputs_are_accumulated_once():
    """method.SHARE_IDENTICAL_INPUTS: Linears that are handed the SAME tensor (q / k / v, gate / up -- the drivers hook each of them,
    opt.py:131-140) accumulate its X^T X once; the others take a copy of that Hessian at post_batch -- bit-identical to accumulating it
    again, whatever the order of the post_batch / preproc calls (opt.py finishes k and v before q; llama.py finishes all, then preprocesses)"""
    from quip_amd import method as M
    from quip_amd.method import QuantMethod
    torch.manual_seed(1)
    lq, lk, lv, lo = (torch.nn.Linear(24, 8) for _ in range(4))
    X = [torch.randn(1, 10, 24) for _ in range(5)]
    Y = [torch.randn(1, 10, 24) for _ in range(5)]          # out_proj's input: another tensor of the same shape

    def run(share):
        M.SHARE_IDENTICAL_INPUTS = share
        try:
            ms = [QuantMethod(l) for l in (lq, lk, lv, lo)]
            calls = 0
            orig = torch.Tensor.addmm_
            def counting(self, *a, **k):
                nonlocal calls
                calls += 1
                return orig(self, *a, **k)
            torch.Tensor.addmm_ = counting
            try:
                for j in range(5):
                    x = X[j].clone()                         # a fresh activation per sample, like a block forward produces
                    for m in ms[:3]:
                        m.add_batch(x.data, None)            # the drivers pass inp[0].data: a new tensor object over the same memory
                    ms[3].add_batch(Y[j].clone().data, None)
                    del x
            finally:
                torch.Tensor.addmm_ = orig
            # opt.py's order: k, v finish (and are preprocessed) before q, the leader
            ms[1].post_batch(); ms[1].preproc(preproc_gptqH=True)
            ms[2].post_batch()
            ms[0].post_batch(); ms[0].preproc(preproc_gptqH=True)
            ms[3].post_batch()
            return ms, calls
        finally:
            M.SHARE_IDENTICAL_INPUTS = True
    shared, n_shared = run(True)
    plain, n_plain = run(False)
    assert n_plain == 20 and n_shared == 10                  # q (leader) and out_proj only
    assert [m.nsamples for m in shared] == [5, 5, 5, 5]
    for a, b in zip(shared, plain):
        assert a.H.dtype == torch.float32 and torch.equal(a.H, b.H)
    assert shared[2].H.data_ptr() != shared[0].H.data_ptr()
    assert not M._last_inputs                                # nothing is kept alive past post_batch


def test_a_shared_input_that_stops_being_shared_is_an_error_not_a_wrong_hessian():
    import pytest
    from quip_amd.method import QuantMethod
    a, b = QuantMethod(torch.nn.Linear(16, 4)), QuantMethod(torch.nn.Linear(16, 4))
    x = torch.randn(1, 6, 16)
    a.add_batch(x, None)
    b.add_batch(x, None)                                     # b now relies on a
    y = torch.randn(1, 6, 16)
    a.add_batch(x, None)
    with pytest.raises(RuntimeError, match="SHARE_IDENTICAL_INPUTS"):
        b.add_batch(y, None)
    a.free(); b.free()


def test_two_launch_policy_of_a_decode_step():
    """quant.two_launch_from: the row count from which a packed layer's decode launches split into [prologue-only launch] + [dequant-GEMM]
    -- by the layer's size unless quant.TWO_LAUNCH_ROWS pins it (profiles/r05s_two_launch_*.jsonl), always inside 2 .. FUSED_MAX_ROWS + 1"""
    from types import SimpleNamespace as NS
    from quip_amd import quant, ops
    keep = quant.TWO_LAUNCH_ROWS
    try:
        quant.TWO_LAUNCH_ROWS = None
        assert quant.two_launch_from(NS(infeatures=2048, outfeatures=2048)) == 3
        assert quant.two_launch_from(NS(infeatures=8192, outfeatures=2048)) == 3          # OPT-1.3B fc2
        assert quant.two_launch_from(NS(infeatures=2048, outfeatures=8192)) == 3
        assert quant.two_launch_from(NS(infeatures=4096, outfeatures=4096)) == ops.FUSED_MAX_ROWS + 1
        assert quant.two_launch_from(NS(infeatures=11008, outfeatures=4096)) == ops.FUSED_MAX_ROWS + 1
        for pinned, want in ((1, 2), (2, 2), (4, 4), (5, 5), (99, ops.FUSED_MAX_ROWS + 1)):
            quant.TWO_LAUNCH_ROWS = pinned
            assert quant.two_launch_from(NS(infeatures=4096, outfeatures=4096)) == want
    finally:
        quant.TWO_LAUNCH_ROWS = keep


def test_float_reciprocal_division_of_the_blocked_stage_is_exact():
    """csrc/ortho_blk.hip `bk_div`: int((float(a) + 0.5f) * (1.0f / d)) == a // d for 0 <= a < 2^22 -- the kernel replaced a dozen integer
    divisions by run-time divisors (~50 instructions each on a wave that issues one instruction every 4+ clocks) with this.  Checked here in
    float32 arithmetic as the device does it (round to nearest, no fused contraction: -ffp-contract=off), for every divisor the kernel can
    meet (n / 8 chunks per row, P / 8 chunks per factor row, P) and every dividend up to 2^22."""
    import numpy as np
    a = np.arange(0, 1 << 22, dtype=np.int64)
    af = a.astype(np.float32) + np.float32(0.5)
    for d in list(range(1, 130)) + [160, 192, 256, 320, 344, 512, 688, 768, 1024, 1376, 2048, 4095, 4096]:
        rcp = np.float32(1.0) / np.float32(d)
        q = (af * rcp).astype(np.int64)                                  # truncation of a non-negative float
        assert np.array_equal(q, a // d), d


def test_dpp_group_sums_of_the_blocked_stage():
    """csrc/ortho_blk.hip sums the C chunk products of one first-stage dot product -- they sit in C adjacent lanes, C = 2 / 4 / 8 / 16 -- on the
    DPP network: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror, as many steps as log2 C.  Lane-level emulation: after
    the steps EVERY lane of a group holds the group's sum (the kernel lets the lane with chunk 0 write it), other groups untouched; and
    csrc/wavered.h's wave sum (the four steps + four v_readlane) is the sum of all 64 lanes."""
    import numpy as np
    rng = np.random.default_rng(0)
    lanes = np.arange(64)
    src = [lanes ^ 1, lanes ^ 2, (lanes & ~7) | (7 - (lanes & 7)), (lanes & ~15) | (15 - (lanes & 15))]
    for C, steps in ((2, 1), (4, 2), (8, 3), (16, 4)):
        v = rng.integers(-1000, 1000, 64).astype(np.int64)              # integers: the check is about WHICH lanes meet, not rounding
        want = v.reshape(64 // C, C).sum(1).repeat(C)
        for s in range(steps):
            v = v + v[src[s]]
        assert np.array_equal(v, want), C
    v = rng.integers(-1000, 1000, 64).astype(np.int64)
    total = v.sum()
    for s in range(4):
        v = v + v[src[s]]
    assert (v[0] + v[16]) + (v[32] + v[48]) == total
