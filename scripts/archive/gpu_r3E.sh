#!/bin/bash
# round 3E: multi-exponent dequantisation in the fused decode launches -- tests, then both decode loops
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_decode_bigp.py tests/test_gpu_decode_hf.py tests/test_gpu_decode_head.py tests/test_gpu_decode_step.py tests/test_gpu_decode_attn.py -q > $O/r3E_tests.log 2>&1
grep -E "passed|failed|FAILED" $O/r3E_tests.log
timeout 900 python scripts/decode_opt.py --only-chained --v3-only > $O/r3E_opt.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r3E_opt.json')); print({k: round(v['tok_per_s'],1) for k,v in d.items() if isinstance(v,dict) and 'tok_per_s' in v})"
timeout 900 python scripts/decode_llama.py --no-dense > $O/r3E_llama.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r3E_llama.json')); print({k: round(v['tok_per_s'],1) for k,v in d.items() if isinstance(v,dict) and 'tok_per_s' in v})"
