#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/*.npz by running the REFERENCE
(Cornell-RelaxML/QuIP, mounted read-only at /root/reference) on CPU.

Run only in the authoring container (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

Nothing in tests/, bench.py or __graft_entry__.py imports the reference at run
time; they read the .npz files this script writes.  Every fixture records the
reference symbol (file:line) it came from in its `__doc__` entry.

The reference needs one shim: `primefac` (method.py:8) is not installed here, so
tests/golden/_shims/primefac.py supplies ascending trial division.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, REF)

import quant as ref_quant            # noqa: E402
import method as ref_method          # noqa: E402
import vector_balance as ref_vb      # noqa: E402
import bal as ref_bal                # noqa: E402
import near as ref_near              # noqa: E402
import optq_counter as ref_counter   # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "ref_zs_quant", os.path.join(REF, "zeroShot/models/quant.py"))
ref_zs_quant = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_zs_quant)

torch.set_num_threads(1)   # deterministic summation order for the fp32 fixtures


def save(name, doc, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    out["__doc__"] = np.asarray(doc)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def correlated_H(d, seed):
    """SURVEY.md section 4 / 8(d) parity fixture: strongly correlated Hessian."""
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(d, d, generator=g) / d ** 0.5
    sv = torch.arange(1, d + 1, dtype=torch.float32) ** -0.75
    X = (torch.randn(2 * d, d, generator=g) * sv) @ A
    H = X.T @ X / (2 * d)
    H = H + 0.01 * torch.diag(H).mean() * torch.eye(d)
    return H.float().contiguous()


# ---------------------------------------------------------------- A. grid functions
def gen_grids():
    g = torch.Generator().manual_seed(1)
    W32 = (0.02 * torch.randn(24, 64, generator=g)).float()
    W32[3, :] = 0.0                       # all-zero row: xmin=xmax=0 branch (quant.py:87-89)
    W32[5, :] = W32[5, :].abs()           # strictly non-negative row (xmin clamps to 0)
    W16 = W32.half()
    arrs = {"W32": W32, "W16": W16.view(torch.int16)}
    for bits in (2, 3, 4):
        for tag, W in (("f32", W32), ("f16", W16)):
            # qfn a: quant.py:57-136 (find_params_qfna), quant.py:6-8
            q = ref_quant.Quantizer()
            q.configure(bits, perchannel=True, sym=False, qfn='a', mse=False)
            q.find_params(W, weight=True)
            out = q.quantize(W)
            arrs[f"a{bits}_{tag}_scale"] = q.scale.float()
            arrs[f"a{bits}_{tag}_zero"] = q.zero.float()
            arrs[f"a{bits}_{tag}_out"] = out.float()
            arrs[f"a{bits}_{tag}_outdtype"] = str(out.dtype)
            # qfn c: quant.py:17-21
            q = ref_quant.Quantizer()
            q.configure(bits, perchannel=True, sym=False, qfn='c', mse=False)
            q.find_params(W, weight=True)
            arrs[f"c{bits}_{tag}_out"] = q.quantize(W).float()
            # qfn b: quant.py:10-15, 148-151
            q = ref_quant.Quantizer()
            q.configure(bits, perchannel=True, sym=False, qfn='b', mse=False)
            q.find_params(W, weight=True)
            out = q.quantize(W)
            arrs[f"b{bits}_{tag}_scale"] = q.scale.float().reshape(1)
            arrs[f"b{bits}_{tag}_out"] = out.float()
            arrs[f"b{bits}_{tag}_outdtype"] = str(out.dtype)
    # sym=True / perchannel=False variants of find_params_qfna (quant.py:82-86,121-126)
    q = ref_quant.Quantizer()
    q.configure(4, perchannel=False, sym=True, qfn='a', mse=False)
    q.find_params(W32, weight=True)
    arrs["a4_sym_tensor_scale"] = q.scale.float()
    arrs["a4_sym_tensor_zero"] = q.zero.float()
    arrs["a4_sym_tensor_out"] = q.quantize(W32)
    save("grids", "quant.py:6-21 quantize_qfn{a,b,c}; quant.py:23-163 Quantizer "
         "(configure/find_params/quantize), bits 2/3/4, fp32 and fp16 inputs", **arrs)


# ---------------------------------------------------------------- B. packers
def gen_pack():
    g = torch.Generator().manual_seed(2)
    arrs = {}
    # 4-bit: zeroShot/models/quant.py:183-199
    m, d = 24, 64
    lin = torch.nn.Linear(d, m)
    lin.weight.data = 0.05 * torch.randn(m, d, generator=g)
    lin.bias.data = torch.randn(m, generator=g)
    q = ref_quant.Quantizer()
    q.configure(4, perchannel=True, sym=False, qfn='a', mse=False)
    q.find_params(lin.weight.data, weight=True)
    lin.weight.data = q.quantize(lin.weight.data)       # on-grid weights
    ql = ref_zs_quant.Quant4Linear(lin, q.scale, q.zero)
    codes = torch.round((lin.weight.data + ql.zeros) / ql.scales).to(torch.int)
    arrs.update(p4_W=lin.weight.data, p4_scale=q.scale, p4_zero=q.zero, p4_bias=lin.bias.data,
                p4_codes=codes.to(torch.uint8), p4_qweight=ql.qweight, p4_zeros=ql.zeros,
                p4_scales=ql.scales)
    # 3-bit: quant.py:173-220
    m, d = 8, 1024
    lin = torch.nn.Linear(d, m)
    lin.weight.data = 0.05 * torch.randn(m, d, generator=g)
    q = ref_quant.Quantizer()
    q.configure(3, perchannel=True, sym=False, qfn='a', mse=False)
    q.find_params(lin.weight.data, weight=True)
    lin.weight.data = q.quantize(lin.weight.data)
    ql = ref_quant.Quant3Linear(d, m)
    ql.pack(lin, q.scale, q.zero)
    codes = torch.round((lin.weight.data + ql.zeros) / ql.scales).to(torch.int)
    arrs.update(p3_codes=codes.to(torch.uint8), p3_qweight=ql.qweight)
    save("pack", "zeroShot/models/quant.py:183-199 Quant4Linear.__init__ (int4 rule); "
         "quant.py:185-220 Quant3Linear.pack (3-bit rule)", **arrs)


# ---------------------------------------------------------------- C. butterfly
def gen_butterfly():
    arrs = {}
    gens = {"blocked": ref_method.gen_rand_ortho_butterfly,
            "noblock": ref_method.gen_rand_ortho_butterfly_noblock,
            "nopermute": ref_method.gen_rand_ortho_butterfly_nopermute}
    for n in (6, 40, 64, 192, 768):
        arrs[f"n{n}_factors"] = np.asarray(ref_method.butterfly_factors(n))
        for gname, gen in gens.items():
            if n == 768 and gname != "blocked":
                continue
            np.random.seed(100 + n)
            torch.manual_seed(100 + n)
            (B, p_in, p_out) = gen(n)
            X = torch.randn(n, 5)
            Y = ref_method.mul_ortho_butterfly((B, p_in, p_out), X)
            y1 = ref_method.mul_ortho_butterfly((B, p_in, p_out), X[:, 0].clone())
            key = f"n{n}_{gname}"
            arrs[key + "_B0"] = B[0]
            arrs[key + "_B1"] = B[1]
            arrs[key + "_pin"] = p_in
            arrs[key + "_pout"] = p_out
            arrs[key + "_X"] = X
            arrs[key + "_Y"] = Y
            arrs[key + "_y1"] = y1
            if n <= 64:
                arrs[key + "_dense"] = ref_method.mul_ortho_butterfly((B, p_in, p_out), torch.eye(n))
    for n in (2, 6, 40, 64, 192, 768, 2048, 3072, 4096, 7168, 8192, 11008, 28672):
        arrs[f"factors_{n}"] = np.asarray(ref_method.butterfly_factors(n))
    save("butterfly", "method.py:16-78 butterfly_factors / gen_rand_orthos / "
         "gen_rand_ortho_butterfly{,_noblock,_nopermute} / mul_ortho_butterfly; seeds "
         "np.random.seed(100+n); torch.manual_seed(100+n) before each generator call", **arrs)


# ---------------------------------------------------------------- D/E. LDLQ
def gen_ldlq():
    arrs = {}
    d, m = 192, 40
    H = correlated_H(d, 3)
    arrs["H"] = H
    for bits in (2, 4):
        g = torch.Generator().manual_seed(10 + bits)
        maxq = 2 ** bits - 1
        W = (torch.rand(m, d, generator=g) * (maxq + 0.6) - 0.3).clamp(0, maxq).float()
        arrs[f"W{bits}"] = W
        # vector_balance.py:155-199
        arrs[f"ldl{bits}"] = ref_vb.round_ldl(W, H, bits, n_greedy_passes=0)
        # vector_balance.py:218-291
        arrs[f"ldlblock{bits}"] = ref_vb.round_ldl_block(W, H, bits, n_greedy_passes=0)
        # vector_balance.py:381-422
        arrs[f"gptqequiv{bits}"] = ref_vb.round_ldl_gptqequiv(W, H, bits)
        # unbiased: eta = torch.rand(w.shape) drawn inside (vector_balance.py:174-175)
        torch.manual_seed(77 + bits)
        eta = torch.rand(W.shape)
        torch.manual_seed(77 + bits)
        arrs[f"ldl{bits}_unbiased"] = ref_vb.round_ldl(W, H, bits, n_greedy_passes=0, unbiased=True)
        arrs[f"eta{bits}"] = eta
        near = torch.clamp(torch.floor(W + 0.5), 0, maxq)
        arrs[f"proxy_ldl{bits}"] = ref_vb.hessian_loss(arrs[f"ldl{bits}"] - W, H).item()
        arrs[f"proxy_near{bits}"] = ref_vb.hessian_loss(near - W, H).item()
    # E. quantize_weight_vecbal (vector_balance.py:500-532), fp16 and fp32 inputs
    g = torch.Generator().manual_seed(5)
    Wf = (0.02 * torch.randn(m, d, generator=g)).float()
    arrs["Wf32"] = Wf
    arrs["Wf16"] = Wf.half().view(torch.int16)
    for bits in (2, 4):
        for tag, W in (("f32", Wf), ("f16", Wf.half())):
            q = ref_quant.Quantizer()
            q.configure(bits, perchannel=True, sym=False, qfn='a', mse=False)
            q.find_params(W, weight=True)
            for lazy in (False, True):
                out = ref_vb.quantize_weight_vecbal(
                    w=W, H=H, nbits=bits, npasses=0, scale=q.scale, zero=q.zero, maxq=q.maxq,
                    unbiased=False, qfn='a', qmethod='ldlq', lazy_batch=lazy)
                arrs[f"vecbal_a{bits}_{tag}_lazy{int(lazy)}"] = out.float()
                out = ref_vb.quantize_weight_vecbal(
                    w=W, H=H, nbits=bits, npasses=0, scale=None, zero=None, maxq=q.maxq,
                    unbiased=False, qfn='b', qmethod='ldlq', lazy_batch=lazy)
                arrs[f"vecbal_b{bits}_{tag}_lazy{int(lazy)}"] = out.float()
    save("ldlq", "vector_balance.py:155-199 round_ldl; :218-291 round_ldl_block; :381-422 "
         "round_ldl_gptqequiv; :500-532 quantize_weight_vecbal.  H = correlated fixture "
         "(SURVEY.md 8(d)), d=192 (one full + one ragged 128-block), m=40", **arrs)


def gen_ldl_blocksizes():
    """round 6: the reference's round_ldl_block at block sizes other than its default (vector_balance.py:218-257, `blocksize`), same
    fixture as gen_ldlq: d = 192, m = 40, correlated H"""
    arrs = {}
    d, m = 192, 40
    H = correlated_H(d, 3)
    for bits in (2, 4):
        g = torch.Generator().manual_seed(10 + bits)
        maxq = 2 ** bits - 1
        W = (torch.rand(m, d, generator=g) * (maxq + 0.6) - 0.3).clamp(0, maxq).float()
        for bsz in (32, 64, 1000):
            arrs[f"ldlblock{bits}_bs{bsz}"] = ref_vb.round_ldl_block(W, H, bits, blocksize=bsz, n_greedy_passes=0)
    save("ldl_blocksizes", "vector_balance.py:218-257 round_ldl_block(blocksize = 32, 64, 1000), n_greedy_passes=0, on ldlq.npz's H / W2 / W4", **arrs)


# ---------------------------------------------------------------- F/G. QuantMethod end to end
def gen_method():
    arrs = {}
    m, d = 24, 96

    def fresh_layer():
        torch.manual_seed(1234)
        np.random.seed(1234)
        lin = torch.nn.Linear(d, m).half()
        lin.weight.data = (0.02 * torch.randn(m, d)).half()
        A = torch.randn(d, d) / d ** 0.5
        sv = torch.arange(1, d + 1, dtype=torch.float32) ** -0.5
        X = ((torch.randn(6, 64, d) * sv) @ A).half()          # 6 "samples" x 64 tokens
        return lin, X

    for case, (extra, lazy, bits, qfn) in {
            "incoh_w2": (0, False, 2, 'b'),
            "incoh_w4_noblock_lazy": (1, True, 4, 'b'),
            "plain_w4_qfna": (None, False, 4, 'a')}.items():
        lin, X = fresh_layer()
        W0 = lin.weight.data.clone()
        meth = ref_bal.Balance(lin)
        meth.configure('ldlq', bits, 0, unbiased=False)
        meth.quantizer = ref_quant.Quantizer()
        meth.quantizer.configure(bits, perchannel=True, sym=False, qfn=qfn, mse=False)
        for j in range(X.shape[0]):
            meth.add_batch(X[j].unsqueeze(0), None)             # method.py:98-120
        H64 = meth.H.clone()
        meth.post_batch()                                       # method.py:122-123
        Hraw = meth.H.clone()
        arrs.update(X=X.view(torch.int16), W0=W0.view(torch.int16), H64=H64, Hraw=Hraw)
        log = []
        names = ("gen_rand_ortho_butterfly", "gen_rand_ortho_butterfly_noblock")
        origs = {g: getattr(ref_method, g) for g in names}
        for gname in names:
            def rec(n, _orig=origs[gname]):
                r = _orig(n)
                log.append(r)
                return r
            setattr(ref_method, gname, rec)
        np.random.seed(4321)
        torch.manual_seed(4321)
        if extra is None:
            meth.preproc(preproc_gptqH=True, percdamp=.01)
        else:
            meth.preproc(preproc_gptqH=True, percdamp=.01, preproc_rescale=True,
                         preproc_proj=True, preproc_proj_extra=extra)      # method.py:125-193
        for gname in names:
            setattr(ref_method, gname, origs[gname])
        arrs[case + "_Wpre"] = lin.weight.data.clone().view(torch.int16)
        arrs[case + "_Hpre"] = meth.H.clone()
        if extra is not None:
            arrs[case + "_scaleWH"] = meth.scaleWH
            arrs[case + "_projU"] = meth.projU
            for side, (B, p_in, p_out) in zip("UV", log):
                arrs[f"{case}_{side}_B0"] = B[0]
                arrs[f"{case}_{side}_B1"] = B[1]
                arrs[f"{case}_{side}_pin"] = p_in
                arrs[f"{case}_{side}_pout"] = p_out
        meth.fasterquant(lazy_batch=lazy)                        # bal.py:21-48
        arrs[case + "_Wq"] = lin.weight.data.clone().view(torch.int16)
        if case == "incoh_w2":
            arrs[case + "_Hpost"] = meth.H.clone()
        arrs[case + "_error"] = meth.error
        arrs[case + "_Hmag"] = meth.Hmag
    # Nearest (near.py:7-20) on the same layer, qfn a w4 with gptqH only
    lin, X = fresh_layer()
    meth = ref_near.Nearest(lin)
    meth.quantizer = ref_quant.Quantizer()
    meth.quantizer.configure(4, perchannel=True, sym=False, qfn='a', mse=False)
    meth.H = arrs["Hraw"].clone()
    meth.preproc(preproc_gptqH=True, percdamp=.01)
    meth.fasterquant()
    arrs["nearest_w4_Wq"] = lin.weight.data.clone().view(torch.int16)
    arrs["nearest_w4_error"] = meth.error
    save("method", "method.py:80-233 QuantMethod (add_batch/post_batch/preproc/postproc/"
         "error_compute) driven through bal.py:13-48 Balance and near.py:5-20 Nearest on an "
         "fp16 nn.Linear(96->24). Seeds: manual_seed(1234) for data, np/torch seed 4321 "
         "immediately before preproc (U drawn first, then V: method.py:162-163)", **arrs)


# ---------------------------------------------------------------- H. optq_counter
def gen_counter():
    import io
    import contextlib
    arrs = {}
    for n in (64, 256):
        torch.manual_seed(0)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref_counter.counter(n, n, 0.01)                     # optq_counter.py:7-31
        vals = {ln.split(":")[0]: float(ln.split(":")[1]) for ln in buf.getvalue().strip().splitlines()}
        arrs[f"n{n}_ldl_loss"] = vals["ldl_loss"]
        arrs[f"n{n}_near_loss"] = vals["near_loss"]
    save("counter", "optq_counter.py:7-31 counter(n, n, 0.01): deterministic ldl_loss / near_loss "
         "(round_ldl_gptqequiv vs nearest on the constructed (W,H))", **arrs)


# ---------------------------------------------------------------- I. greedy passes, LDLQ-RG, GPTQ
def gen_rounders():
    import gptq as ref_gptq
    arrs = {}
    d, m = 192, 40
    H = correlated_H(d, 3)
    arrs["H"] = H
    g = torch.Generator().manual_seed(12)
    W = (torch.rand(m, d, generator=g) * 3.6 - 0.3).clamp(0, 3).float()
    arrs["W2"] = W
    arrs["ldl2_greedy3"] = ref_vb.round_ldl(W, H, 2, n_greedy_passes=3)                   # vector_balance.py:155-199
    arrs["ldlblock2_greedy3"] = ref_vb.round_ldl_block(W, H, 2, n_greedy_passes=3)        # :218-291
    arrs["ldlqRG2_greedy2"] = ref_vb.round_sorted_ldlqRG(W, H, 2, n_greedy_passes=2)      # :139-153
    # GPTQ.fasterquant (gptq.py:19-115) on CPU: its only CUDA call is a synchronize
    torch.manual_seed(3)
    lin = torch.nn.Linear(d, m, bias=False)
    lin.weight.data = (0.02 * torch.randn(m, d)).float()
    arrs["gptq_W0"] = lin.weight.data.clone()
    meth = ref_gptq.GPTQ(lin)
    meth.quantizer = ref_quant.Quantizer()
    meth.quantizer.configure(4, perchannel=True, sym=False, qfn='a', mse=False)
    meth.H = H.clone()
    meth.preproc(preproc_gptqH=True, percdamp=.01)
    arrs["gptq_Hdamped"] = meth.H.clone()
    sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        meth.fasterquant(copy_H=True)
    finally:
        torch.cuda.synchronize = sync
    arrs["gptq_w4_Q"] = lin.weight.data.clone()
    arrs["gptq_w4_scale"] = meth.quantizer.scale.clone()
    arrs["gptq_w4_zero"] = meth.quantizer.zero.clone()
    arrs["gptq_w4_error"] = meth.error
    save("rounders", "vector_balance.py:155-199 / 218-291 round_ldl(_block) with n_greedy_passes=3; :139-153 "
         "round_sorted_ldlqRG with 2 passes; gptq.py:19-115 GPTQ.fasterquant (w4, qfn a, groupsize -1, preproc_gptqH) run on CPU "
         "with torch.cuda.synchronize stubbed.  H = correlated fixture, d=192, m=40", **arrs)


# ---------------------------------------------------------------- I2. GPTQ with groups, qfn b / c, Conv1D
GPTQ_CASES = [  # name, d, m, bits, groupsize, qfn, sym, layer kind
    ("w3_g64_a", 384, 40, 3, 64, 'a', False, "linear"),
    ("w4_g32_c_sym", 384, 40, 4, 32, 'c', True, "linear"),
    ("w2_g128_a", 384, 24, 2, 128, 'a', False, "linear"),
    ("w4_g64_a_ragged", 320, 24, 4, 64, 'a', False, "linear"),     # 320 = 2.5 blocks of 128
    ("w3_g16_a_sym", 256, 24, 3, 16, 'a', True, "linear"),
    ("w4_c", 256, 24, 4, -1, 'c', False, "linear"),
    ("w4_b", 128, 24, 4, -1, 'b', False, "linear"),
    ("w4_g64_b", 128, 24, 4, 64, 'b', False, "linear"),
]


def gen_gptq_groups():
    """gptq.py:19-115 GPTQ.fasterquant on CPU for the configurations of GPTQ_CASES (groupsize, qfn b/c, sym, Conv1D)."""
    import gptq as ref_gptq
    import transformers
    arrs = {}
    sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        for name, d, m, bits, gs, qfn, sym, kind in GPTQ_CASES:
            H = correlated_H(d, 7 + d)
            torch.manual_seed(d + m + bits)
            W0 = (0.02 * torch.randn(m, d)).float()
            if kind == "conv1d":
                lin = transformers.Conv1D(m, d)                 # weight [d, m]
                lin.weight.data = W0.t().contiguous()
            else:
                lin = torch.nn.Linear(d, m, bias=False)
                lin.weight.data = W0.clone()
            meth = ref_gptq.GPTQ(lin)
            meth.quantizer = ref_quant.Quantizer()
            meth.quantizer.configure(bits, perchannel=True, sym=sym, qfn=qfn, mse=False)
            meth.H = H.clone()
            meth.preproc(preproc_gptqH=True, percdamp=.01)
            meth.fasterquant(groupsize=gs, copy_H=True)
            arrs[f"{name}_W0"] = W0
            arrs[f"H{d}"] = H                                  # one Hessian per width (seed 7 + d)
            arrs[f"{name}_Q"] = lin.weight.data.clone()
            arrs[f"{name}_error"] = meth.error
            arrs[f"{name}_scale"] = meth.quantizer.scale.clone()
            if meth.quantizer.zero is not None:
                arrs[f"{name}_zero"] = meth.quantizer.zero.clone()
    finally:
        torch.cuda.synchronize = sync
    save("gptq_groups", "gptq.py:19-115 GPTQ.fasterquant run on CPU (torch.cuda.synchronize stubbed) after preproc(preproc_gptqH, "
         "percdamp .01), perchannel, mse off; cases (name: d, m, bits, groupsize, qfn, sym, layer) = " + repr(GPTQ_CASES) +
         "; H<d> = the Hessian of width d before damping; per case W0 [m,d], Q = layer.weight after, error, and the quantiser left behind", **arrs)


# ---------------------------------------------------------------- H. the reference DRIVER, end to end
def gen_driver():
    """/root/reference/opt.py:29-190 `opt_sequential`, unmodified, on the tiny random-init fp16 OPT of tiny_model.py, CPU,
    for `nearest` (w4 qfn a), `ldlq` (w4 qfn a) and `ldlq --incoh_processing` (w2 qfn b; opt.py:594-600 turns that flag into
    pre_gptqH + pre_rescale + pre_proj with the blocked butterfly, pre_proj_extra = 0).  Records, per quantised Linear in
    driver order: error, Hmag (method.py:228-233), the first 8 rows of the final weights and a SHA-256 of all of them;
    plus the quantised model's logits on two probe sequences.  numpy / torch RNGs are seeded right before each run
    (the reference itself never seeds them, datautils.py:5-7)."""
    import hashlib
    import types
    import tiny_model as TM
    import opt as ref_opt                      # the reference driver module itself
    arrs = {}
    rec = []
    orig_free = ref_method.QuantMethod.free

    def recording_free(self):                   # error / Hmag live on the method object until free() (opt.py:165-168)
        rec.append((float(self.error), float(self.Hmag)))
        return orig_free(self)
    ref_method.QuantMethod.free = recording_free
    for cname, cfg in TM.CONFIGS.items():
        model = TM.build_tiny_opt()
        args = types.SimpleNamespace(nsamples=TM.NSAMPLES, **cfg)
        del rec[:]
        np.random.seed(0)
        torch.manual_seed(0)
        quantizers, errors = ref_opt.opt_sequential(model, TM.calibration_batches(), torch.device("cpu"), args)
        assert len(rec) == len(errors) == 12
        arrs[f"{cname}_error"] = np.asarray([r[0] for r in rec], np.float64)
        arrs[f"{cname}_Hmag"] = np.asarray([r[1] for r in rec], np.float64)
        names = sorted(quantizers.keys(), key=lambda k: list(quantizers.keys()).index(k))
        arrs[f"{cname}_names"] = np.asarray(names)
        for k in names:
            w = dict(model.named_parameters())[k + ".weight"].detach()
            assert w.dtype == torch.float16
            arrs[f"{cname}_{k}_rows8"] = w[:8].view(torch.int16).numpy().copy()
            arrs[f"{cname}_{k}_sha256"] = np.asarray(hashlib.sha256(w.contiguous().view(torch.int16).numpy().tobytes()).hexdigest())
        with torch.no_grad():
            arrs[f"{cname}_logits"] = model(TM.probe_tokens()).logits.float().numpy().astype(np.float16)
    ref_method.QuantMethod.free = orig_free
    with torch.no_grad():
        arrs["fp16_logits"] = TM.build_tiny_opt()(TM.probe_tokens()).logits.float().numpy().astype(np.float16)
    save("driver", "opt.py:29-190 opt_sequential (the reference driver, unmodified) on tests/golden/tiny_model.py: per-Linear "
         "error / Hmag, final fp16 weights (first 8 rows + SHA-256), logits; configs nearest_w4, ldlq_w4, ldlq_w2_incoh", **arrs)


# ---------------------------------------------------------------- H1b. reference-vs-reference spread of the LDLQ driver runs
def gen_driver_spread():
    """How far do two runs of the REFERENCE's own opt_sequential land from each other when only the fp summation order changes?
    The LDLQ driver configs of gen_driver() re-run with torch.set_num_threads in {1, 2, 8} and oneDNN on / off (different
    blocking of the fp16 / fp32 CPU matmuls -- same code, same seeds, same model).  Block 0 sees identical Hessians and moves by
    <= 1.5e-2; block 1's Hessians are downstream of block 0's near-tie code flips and its per-Linear proxy errors move by
    percent.  tests/test_gpu_driver.py derives its block-1 gates from the maxima recorded here instead of hand-picked numbers.
    (Thread-count dependent: the values reproduce on a machine with the same core count and MKL / oneDNN build only; the test
    uses the per-block maxima, not the individual entries.)"""
    import types
    import tiny_model as TM
    import opt as ref_opt
    rec = []
    orig_free = ref_method.QuantMethod.free

    def recording_free(self):
        rec.append(float(self.error))
        return orig_free(self)
    ref_method.QuantMethod.free = recording_free
    variants = [(1, True), (2, True), (8, True), (1, False), (8, False)]
    arrs = {"variants_threads_onednn": np.asarray([[t, int(m)] for t, m in variants])}
    try:
        for cname in ("ldlq_w4", "ldlq_w2_incoh"):
            rows = []
            for threads, onednn in variants:
                torch.set_num_threads(threads)
                torch.backends.mkldnn.enabled = onednn
                model = TM.build_tiny_opt()
                del rec[:]
                np.random.seed(0)
                torch.manual_seed(0)
                ref_opt.opt_sequential(model, TM.calibration_batches(), torch.device("cpu"),
                                       types.SimpleNamespace(nsamples=TM.NSAMPLES, **TM.CONFIGS[cname]))
                rows.append(np.asarray(rec, np.float64))
            e = np.stack(rows)
            arrs[f"{cname}_errors"] = e
            rel = np.abs(e[1:] - e[0]) / e[0]
            arrs[f"{cname}_rel_spread_block0"] = np.asarray(rel[:, :6].max())
            arrs[f"{cname}_rel_spread_block1"] = np.asarray(rel[:, 6:].max())
            arrs[f"{cname}_rel_spread_sum"] = np.asarray((np.abs(e[1:].sum(1) - e[0].sum()) / e[0].sum()).max())
            print(cname, "block0", rel[:, :6].max(), "block1", rel[:, 6:].max(), "sum", arrs[f"{cname}_rel_spread_sum"])
        # the Llama twin (llama.py:36-171, Balance branch with the qbits shim): 14 Linears, block 0 = the first 7
        import llama as ref_llama
        from transformers.models.llama import modeling_llama as ML
        orig_fwd, rot, sync = ML.LlamaDecoderLayer.forward, {}, torch.cuda.synchronize

        def fwd(self, hidden_states, *a, position_embeddings=None, position_ids=None, **kw):
            if position_embeddings is None:
                position_embeddings = rot["m"](hidden_states, position_ids=position_ids)
            return orig_fwd(self, hidden_states, *a, position_embeddings=position_embeddings, position_ids=position_ids, **kw)
        ML.LlamaDecoderLayer.forward = fwd
        torch.cuda.synchronize = lambda *a, **k: None
        try:
            with TM.balance_configure_shim(ref_bal.Balance):
                rows = []
                for threads, onednn in variants:
                    torch.set_num_threads(threads)
                    torch.backends.mkldnn.enabled = onednn
                    model = TM.build_tiny_llama()
                    rot["m"] = model.model.rotary_emb
                    ref_llama.args = types.SimpleNamespace(nsamples=TM.NSAMPLES, **TM.LLAMA_CONFIGS["ldlq_w2_incoh"])
                    del rec[:]
                    np.random.seed(0)
                    torch.manual_seed(0)
                    ref_llama.llama_sequential(model, TM.calibration_batches(), torch.device("cpu"))
                    rows.append(np.asarray(rec, np.float64))
            e = np.stack(rows)
            rel = np.abs(e[1:] - e[0]) / e[0]
            arrs["llama_ldlq_w2_incoh_errors"] = e
            arrs["llama_ldlq_w2_incoh_rel_spread_block0"] = np.asarray(rel[:, :7].max())
            arrs["llama_ldlq_w2_incoh_rel_spread_block1"] = np.asarray(rel[:, 7:].max())
            arrs["llama_ldlq_w2_incoh_rel_spread_sum"] = np.asarray((np.abs(e[1:].sum(1) - e[0].sum()) / e[0].sum()).max())
            print("llama ldlq_w2_incoh block0", rel[:, :7].max(), "block1", rel[:, 7:].max(), "sum", arrs["llama_ldlq_w2_incoh_rel_spread_sum"])
        finally:
            ML.LlamaDecoderLayer.forward = orig_fwd
            torch.cuda.synchronize = sync
    finally:
        ref_method.QuantMethod.free = orig_free
        torch.set_num_threads(1)
        torch.backends.mkldnn.enabled = True
    save("driver_spread", "opt.py:29-190 opt_sequential and llama.py:36-171 llama_sequential (reference, CPU) re-run under (threads, oneDNN) "
         "variants [(1,on) = driver.npz / driver_llama.npz, (2,on), (8,on), (1,off), (8,off)]: per-Linear proxy errors [variant, 12 | 14] and "
         "the largest relative deviation from variant 0 per block -- the reference's own run-to-run noise under fp re-ordering", **arrs)


# ---------------------------------------------------------------- H2. the reference LLAMA driver
def gen_llama_driver():
    """/root/reference/llama.py:36-171 `llama_sequential`, unmodified, on the tiny random-init fp16 Llama of tiny_model.py, CPU,
    for the branches that run as shipped: `nearest` and `gptq` (groupsize -1 and 64), plus `ldlq` w2 qfn b with incoherence
    processing (BASELINE configs[3]) through the Balance branch with tiny_model.balance_configure_shim dropping the stray
    args.qbits of llama.py:110-115.  Two things outside the reference are
    adapted so that its code runs at all under transformers 5: `llama.args` (the driver reads a module
    global) is set from tiny_model.LLAMA_CONFIGS, and transformers' LlamaDecoderLayer.forward is wrapped so that a call
    WITHOUT position_embeddings (the reference forwards only attention_mask / position_ids, llama.py:134,160) computes them
    from position_ids with the model's own rotary module -- the values HF's LlamaModel.forward would have passed."""
    import hashlib
    import types
    import tiny_model as TM
    import llama as ref_llama
    from transformers.models.llama import modeling_llama as ML
    arrs = {}
    rec = []
    orig_free = ref_method.QuantMethod.free
    orig_fwd = ML.LlamaDecoderLayer.forward
    rot = {}

    def recording_free(self):
        rec.append((float(self.error), float(self.Hmag)))
        return orig_free(self)

    def fwd(self, hidden_states, *a, position_embeddings=None, position_ids=None, **kw):
        if position_embeddings is None:
            position_embeddings = rot["m"](hidden_states, position_ids=position_ids)
        return orig_fwd(self, hidden_states, *a, position_embeddings=position_embeddings, position_ids=position_ids, **kw)
    ref_method.QuantMethod.free = recording_free
    ML.LlamaDecoderLayer.forward = fwd
    sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        shim = TM.balance_configure_shim(ref_bal.Balance)    # llama.py:110-115 passes a stray args.qbits (see tiny_model.py)
        shim.__enter__()
        for cname, cfg in TM.LLAMA_CONFIGS.items():
            model = TM.build_tiny_llama()
            rot["m"] = model.model.rotary_emb
            ref_llama.args = types.SimpleNamespace(nsamples=TM.NSAMPLES, **cfg)
            del rec[:]
            np.random.seed(0)
            torch.manual_seed(0)
            quantizers = ref_llama.llama_sequential(model, TM.calibration_batches(), torch.device("cpu"))
            assert len(rec) == len(quantizers) == 14
            arrs[f"{cname}_error"] = np.asarray([r[0] for r in rec], np.float64)
            arrs[f"{cname}_Hmag"] = np.asarray([r[1] for r in rec], np.float64)
            names = [k.replace("model.decoder.layers.", "model.layers.") for k in quantizers.keys()]   # llama.py:153 keeps OPT's prefix
            arrs[f"{cname}_names"] = np.asarray(names)
            params = dict(model.named_parameters())
            for k in names:
                w = params[k + ".weight"].detach()
                assert w.dtype == torch.float16
                arrs[f"{cname}_{k}_rows8"] = w[:8].view(torch.int16).numpy().copy()
                arrs[f"{cname}_{k}_sha256"] = np.asarray(hashlib.sha256(w.contiguous().view(torch.int16).numpy().tobytes()).hexdigest())
            with torch.no_grad():
                arrs[f"{cname}_logits"] = model(TM.probe_tokens()).logits.float().numpy().astype(np.float16)
        with torch.no_grad():
            arrs["fp16_logits"] = TM.build_tiny_llama()(TM.probe_tokens()).logits.float().numpy().astype(np.float16)
    finally:
        shim.__exit__()
        ref_method.QuantMethod.free = orig_free
        ML.LlamaDecoderLayer.forward = orig_fwd
        torch.cuda.synchronize = sync
    save("driver_llama", "llama.py:36-171 llama_sequential (the reference driver, unmodified; args injected as the module global it reads, "
         "HF's LlamaDecoderLayer.forward wrapped to derive position_embeddings from position_ids) on tests/golden/tiny_model.py "
         "build_tiny_llama: per-Linear error / Hmag in driver order, final fp16 weights (first 8 rows + SHA-256), logits; "
         "configs nearest_w4, gptq_w4, gptq_w3_g64, ldlq_w2_incoh (Balance branch, llama.py:107-115, with the stray args.qbits "
         "argument dropped by tiny_model.balance_configure_shim)", **arrs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rounders":
        gen_rounders()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ldl_blocksizes":
        gen_ldl_blocksizes()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gptq_groups":
        gen_gptq_groups()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "driver_llama":
        gen_llama_driver()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "driver":
        gen_driver()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "driver_spread":
        gen_driver_spread()
        sys.exit(0)
    gen_grids()
    gen_pack()
    gen_butterfly()
    gen_ldlq()
    gen_ldl_blocksizes()
    gen_method()
    gen_counter()
    gen_rounders()
    gen_gptq_groups()
    gen_driver()
    gen_driver_spread()
    gen_llama_driver()
