#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  QUIP_FEEDBACK_REPS=40 timeout 600 python -m pytest -q -x -m gpu tests/test_gpu_feedback_stress.py > $O/stress_$i.log 2>&1; echo "run $i rc=$?"; grep -E "AssertionError|passed|failed" $O/stress_$i.log | cut -c1-600 | head -5
done
