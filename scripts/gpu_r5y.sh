#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for shape in "28672 7168 256" "4096 4096 2048"; do
  echo "### mb $shape bf16"; timeout 200 build_gpu/k2lab mb $shape 2 bf16 "4x4,nl4" 2>&1 | grep -E "mb"
  echo "### nosums"; timeout 200 build_gpu/k2lab_nosums mb $shape 2 bf16 "mb32<2,4x2,4x4,nl4" 2>&1 | grep -E "mb"
done
} > $O/r05y_k2lab_mb32_nosums.txt 2>&1
cut -c1-170 $O/r05y_k2lab_mb32_nosums.txt
