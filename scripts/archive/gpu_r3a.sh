#!/bin/bash
# round 3, GPU call A: the whole -m gpu suite (new: reference drivers staged in oracle/_ref, HF decode parity, ADVICE fixes),
# then a kernel trace of the current decode loop as the baseline of the round.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
ls oracle/_ref
timeout 1200 python -m pytest tests -m gpu -q -rP -p no:cacheprovider > $O/pytest_gpu_r03a.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $O/pytest_gpu_r03a.log | tail -5
grep -E "drivers executed|decode vs HF" $O/pytest_gpu_r03a.log
grep -E "^(FAILED|ERROR)" $O/pytest_gpu_r03a.log | head -30
bash scripts/prof_decode.sh r03a > $O/prof_decode_r03a.log 2>&1; tail -25 $O/prof_decode_r03a.log
