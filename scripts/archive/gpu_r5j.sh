#!/bin/bash
# round 5, call j: MB with 4 x 8 accumulator tiles per wave (one compute wave per SIMD), v2 tests, MFMA counters for the winner
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do
echo "### mb 28672x7168 bs256 rep $rep"; timeout 200 build_gpu/k2lab mb 28672 7168 256 2 bf16 2>&1 | grep -E "mb<|old"
echo "### mb 4096x4096 bs2048 rep $rep"; timeout 200 build_gpu/k2lab mb 4096 4096 2048 2 bf16 2>&1 | grep -E "mb<|old"
done
echo "### mb 8192x8192 bs1024"; timeout 200 build_gpu/k2lab mb 8192 8192 1024 2 bf16 2>&1 | grep -E "mb<|old"
echo "### mb 28672x7168 bs64"; timeout 200 build_gpu/k2lab mb 28672 7168 64 2 bf16 2>&1 | grep -E "mb<|old"
echo "### mb f16 4096 bs2048"; timeout 200 build_gpu/k2lab mb 4096 4096 2048 2 f16 2>&1 | grep -E "mb<|old"
} > $O/r05j_k2lab_mb.txt 2>&1
cat $O/r05j_k2lab_mb.txt | cut -c1-150
echo "== pytest v2"; timeout 900 python -m pytest tests/test_gpu_dqgemm_v2.py -x -q > $O/r05j_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/r05j_pytest.log
