#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do echo "### s 28672x7168 bs16 rep $rep"; timeout 200 build_gpu/k2lab s 28672 7168 16 2 bf16 "s<" 2>&1 | grep -E "s<"; done
echo "### s 28672x7168 bs1"; timeout 200 build_gpu/k2lab s 28672 7168 1 2 bf16 "s<" 2>&1 | grep -E "s<"
echo "### s 16384x8192 bs16"; timeout 200 build_gpu/k2lab s 16384 8192 16 2 bf16 "s<" 2>&1 | grep -E "s<"
echo "### s 2048x8192 bs16 f16"; timeout 200 build_gpu/k2lab s 2048 8192 16 2 f16 "s<" 2>&1 | grep -E "s<"
echo "### s w4 28672x7168 bs16"; timeout 200 build_gpu/k2lab s 28672 7168 16 4 bf16 "s<" 2>&1 | grep -E "s<"
} > $O/r05m_k2lab_s_loaders.txt 2>&1
cat $O/r05m_k2lab_s_loaders.txt | cut -c1-150
