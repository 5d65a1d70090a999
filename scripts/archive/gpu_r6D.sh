#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest decode (attention forms)"; timeout 1200 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_decode_attn.py tests/test_gpu_decode_hf.py tests/test_gpu_decode_step.py tests/test_gpu_decode_e2e.py -q -x 2>&1 | tail -3
timeout 900 python scripts/bench_attn_forms.py --arch opt 2>/dev/null | tee $O/r06E_attn_forms.jsonl
timeout 1200 python scripts/bench_attn_forms.py --arch llama --reps 1 2>/dev/null | tee -a $O/r06E_attn_forms.jsonl
