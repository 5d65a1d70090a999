#!/bin/bash
# hunt for the flaky NaN of test_gptq_feedback_ragged_widths[2080]: the whole file, repeatedly, in one process each
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
fails=0
for i in $(seq 1 40); do
  timeout 300 python -m pytest tests/test_gpu_edge_round2.py -x -q > $O/r05n_edge_$i.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "run $i FAILED"; grep -E "FAILED|assert" $O/r05n_edge_$i.log | head -5; else rm -f $O/r05n_edge_$i.log; fi
done
echo "edge_round2 x 40: $fails failures"
# and in the order of the full suite up to that file, three times
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_cholesky_sched.py tests/test_gpu_dqgemm.py tests/test_gpu_driver.py tests/test_gpu_edge_round2.py -x -q > $O/r05n_prefix_$i.log 2>&1; echo "prefix run $i rc=$?"; tail -2 $O/r05n_prefix_$i.log
done
