#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
S=tests/test_gpu_feedback_stress.py
i=0
for pre in "tests/test_gpu_ortho_ldlq.py" "tests/test_gpu_shard_rccl.py" "tests/test_gpu_gptq.py" "tests/test_gpu_dqgemm_v2.py" "tests/test_gpu_decode_fused.py tests/test_gpu_decode_attn.py tests/test_gpu_decode_step.py tests/test_gpu_decode_hf.py tests/test_gpu_decode_e2e.py" \
           "tests/test_gpu_decode_fused.py tests/test_gpu_decode_attn.py tests/test_gpu_decode_step.py tests/test_gpu_decode_hf.py tests/test_gpu_decode_e2e.py tests/test_gpu_dqgemm_v2.py tests/test_gpu_gptq.py tests/test_gpu_shard_rccl.py tests/test_gpu_ortho_ldlq.py"; do
  i=$((i+1))
  QUIP_FEEDBACK_REPS=40 timeout 900 python -m pytest -q -x -m gpu $pre $S > $O/stress_pre_$i.log 2>&1; echo "combo $i ($pre) rc=$?"; grep -E "AssertionError|passed|failed|FAILED" $O/stress_pre_$i.log | cut -c1-700 | head -6
done
