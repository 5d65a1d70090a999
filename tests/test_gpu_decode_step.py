"""-m gpu: the assembled decode step (scripts/decode_opt.py: OPT block on packed w2 layers with incoherence operators,
LayerNorm / bias / residual / ReLU folded into the operator launches, single-launch attention, chained hand-overs) against
its dense fp16 twin -- the whole-step logits check that used to live only in the script.
The twin holds U^T What V / s folded into dense fp16 weights, so the difference is activation rounding (the packed path runs
its GEMM activations in the 16-bit dtype of the model) plus fp16 rounding of the folded weights: gated at 1e-2 of the logits'
norm (measured 3.8e-3)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _mod():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "decode_opt.py")
    spec = importlib.util.spec_from_file_location("decode_opt", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_packed_decode_step_matches_its_dense_twin():
    e_plain, e_fused, e_fused_attn, chained_equal = _mod().decode_check(layers=2, bits=2)
    assert e_plain <= 1e-2 and e_fused <= 1e-2 and e_fused_attn <= 1e-2, (e_plain, e_fused, e_fused_attn)
    assert chained_equal, "chained hand-over launches must reproduce the unchained step bit for bit"
