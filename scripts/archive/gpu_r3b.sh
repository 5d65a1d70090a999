#!/bin/bash
# round 3, GPU call B: the fused decode launches (csrc/decode_fused.hip): unit tests, decode parity tests, decode timing + kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_decode_step.py tests/test_gpu_decode_hf.py tests/test_gpu_driver.py -m gpu -q -rP -x -p no:cacheprovider > $O/pytest_gpu_r03b.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $O/pytest_gpu_r03b.log | tail -3
grep -E "decode vs HF" $O/pytest_gpu_r03b.log
grep -E "^E  " $O/pytest_gpu_r03b.log | head -20
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_dec_r03b -o trace -- python $R/scripts/decode_opt.py --only-chained --v3-only --layers 24 --prompt 8 --tokens 96 > $O/decode_r03b.json 2> $O/decode_r03b.err
echo "rc=$?"; cat $O/decode_r03b.json; tail -3 $O/decode_r03b.err
cd $R; python scripts/rocpd_summary.py $O/prof_dec_r03b/trace_results.db | awk 'NR<=2 || $0 ~ /fused|ortho|dq|decode_attn|Cijk|layer_norm|argmax/' | cut -c1-170 | head -30 > $O/decode_trace_r03b.txt
cat $O/decode_trace_r03b.txt; rm -rf $O/prof_dec_r03b
