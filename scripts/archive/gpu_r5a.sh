#!/bin/bash
# round 5, call a: the d = 28672 side of configs[4] -- new parity tests, then one OPT-30B-geometry block through the reference driver + kernels alone
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest opt30b"; timeout 900 python -m pytest tests/test_gpu_opt30b.py -x -q -s > $O/r05a_pytest_opt30b.log 2>&1; echo "rc=$?"; tail -15 $O/r05a_pytest_opt30b.log
echo "== opt30b block"; timeout 1200 python scripts/opt30b_block.py --out $O/r05a_opt30b_block.json > $O/r05a_opt30b_block.log 2>&1; echo "rc=$?"; tail -5 $O/r05a_opt30b_block.log | cut -c1-3000
echo "== pytest gpu (rest)"; timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_opt30b.py > $O/r05a_pytest_gpu.log 2>&1; echo "rc=$?"; tail -5 $O/r05a_pytest_gpu.log
