#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest two ranks"; timeout 1500 python -m pytest tests/test_gpu_shard_two_ranks.py tests/test_gpu_shard_rccl.py -x -q > $O/r05p_pytest.log 2>&1; echo "rc=$?"; tail -15 $O/r05p_pytest.log | cut -c1-300
{
echo "### h 8192x2048 bs16 f16"; timeout 150 build_gpu/k2lab h 8192 2048 16 2 f16 "h<" 2>&1 | grep -E "h<"
echo "### h 11008x4096 bs16 f16"; timeout 150 build_gpu/k2lab h 11008 4096 16 2 f16 "h<" 2>&1 | grep -E "h<"
echo "### h 4096x4096 bs16 bf16"; timeout 150 build_gpu/k2lab h 4096 4096 16 2 bf16 "h<" 2>&1 | grep -E "h<"
echo "### h 12288x4096 bs16 bf16"; timeout 150 build_gpu/k2lab h 12288 4096 16 2 bf16 "h<" 2>&1 | grep -E "h<"
} > $O/r05p_k2lab_rt.txt 2>&1
cat $O/r05p_k2lab_rt.txt | cut -c1-150
