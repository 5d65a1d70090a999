#!/bin/bash
# round 6, call y: the feedback stress tests at 500 repetitions on the final tree (the one NaN of round 5 never reproduced)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
QUIP_FEEDBACK_REPS=500 timeout 2400 python -m pytest tests/test_gpu_feedback_stress.py tests/test_gpu_edge_round2.py tests/test_gpu_gptq.py -q 2>&1 | tail -6 | tee $O/r06y_feedback_stress_500.txt
