#!/bin/bash
# round 4, call I: ortho_blk with early operand requests; the tests' independent dense operators
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ortho_blk.py tests/test_gpu_decode_fused.py tests/test_gpu_decode_bigp.py tests/test_gpu_decode_head.py tests/test_gpu_decode_e2e.py -x -q -m gpu > gpurun_out/r04i_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r04i_pytest.log
rm -f gpurun_out/r04i_decode_engine.jsonl
for cfg in "--arch opt --blocked" "--arch llama --blocked"; do
  timeout 600 python scripts/decode_engine_bench.py $cfg 2>/dev/null | tail -1 >> gpurun_out/r04i_decode_engine.jsonl; echo "decode $cfg rc=$?"
done
cut -c1-330 gpurun_out/r04i_decode_engine.jsonl
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04i_opt -o trace -- python $GRAFT_REPO_ROOT/scripts/decode_engine_bench.py --arch opt --blocked --layers 4 --prompt 8 --tokens 24 > $GRAFT_REPO_ROOT/gpurun_out/r04i_prof_opt.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python scripts/rocpd_summary.py gpurun_out/prof_r04i_opt/trace_results.db > gpurun_out/r04i_decode_opt_blocked_kernel_trace.txt 2>&1; grep -E "blk_stage|dqgemm|dq_h|decode_attn" gpurun_out/r04i_decode_opt_blocked_kernel_trace.txt | cut -c1-200
rm -rf gpurun_out/prof_r04i_opt
