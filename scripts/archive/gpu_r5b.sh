#!/bin/bash
# round 5, call b: flaky-NaN hunt, determinism tests, multi-row fused stage, decode at 8 / 16 sequences, sharded legs of bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== debug feedback 2080"; timeout 600 python scripts/debug_feedback_2080.py > $O/r05b_debug_feedback.log 2>&1; echo "rc=$?"; tail -12 $O/r05b_debug_feedback.log
echo "== pytest new"; timeout 1200 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_decode_bigp.py tests/test_gpu_decode_hf.py tests/test_gpu_decode_e2e.py tests/test_gpu_opt30b.py tests/test_gpu_shard_rccl.py -x -q -s > $O/r05b_pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed|opt30b|engine|Error" $O/r05b_pytest.log | tail -15
echo "== decode batch sweep"; timeout 600 python scripts/decode_engine_bench.py --arch opt --prompt 32 --tokens 32 --sweep 1:-1,4:-1,5:-1,8:-1,16:-1,32:-1 > $O/r05b_decode_batch.jsonl 2> $O/r05b_decode_batch.err; echo "rc=$?"; cut -c1-420 $O/r05b_decode_batch.jsonl
