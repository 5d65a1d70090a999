#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do
echo "### mb 28672x7168 bs256 rep $rep"; timeout 200 build_gpu/k2lab mb 28672 7168 256 2 bf16 "4x4" 2>&1 | grep -E "mb<"; timeout 200 build_gpu/k2lab mb 28672 7168 256 2 bf16 "4x8" 2>&1 | grep -E "mb<"
echo "### mb 4096x4096 bs2048 rep $rep"; timeout 200 build_gpu/k2lab mb 4096 4096 2048 2 bf16 "4x4" 2>&1 | grep -E "mb<"; timeout 200 build_gpu/k2lab mb 4096 4096 2048 2 bf16 "4x8" 2>&1 | grep -E "mb<"
done
} > $O/r05k_k2lab_mb_loaders.txt 2>&1
cat $O/r05k_k2lab_mb_loaders.txt | cut -c1-150
