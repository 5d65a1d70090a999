#!/bin/bash
# round 5, call h: the role-split headline kernel (loader waves for the weights) in the lab
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2 3; do echo "### rep $rep bf16 bs16"; timeout 150 build_gpu/k2lab h 4096 4096 16 2 bf16 2>&1 | grep -E "h<|hl<"; done
echo "### f16"; timeout 150 build_gpu/k2lab h 4096 4096 16 2 f16 2>&1 | grep -E "h<|hl<"
echo "### bs8"; timeout 150 build_gpu/k2lab h 4096 4096 8 2 bf16 2>&1 | grep -E "h<|hl<"
echo "### bs1"; timeout 150 build_gpu/k2lab h 4096 4096 1 2 bf16 2>&1 | grep -E "h<|hl<"
echo "### 2048"; timeout 150 build_gpu/k2lab h 2048 2048 16 2 bf16 2>&1 | grep -E "h<|hl<"
echo "### 8192x2048"; timeout 150 build_gpu/k2lab h 8192 2048 16 2 bf16 2>&1 | grep -E "h<|hl<"
echo "### 11008x4096"; timeout 150 build_gpu/k2lab h 11008 4096 16 2 bf16 2>&1 | grep -E "h<|hl<"
echo "### w4"; timeout 150 build_gpu/k2lab h 4096 4096 16 4 bf16 2>&1 | grep -E "h<|hl<"
echo "### w4 2048"; timeout 150 build_gpu/k2lab h 2048 2048 16 4 bf16 2>&1 | grep -E "h<|hl<"
} > $O/r05h_k2lab_hl.txt 2>&1
cat $O/r05h_k2lab_hl.txt | cut -c1-150
