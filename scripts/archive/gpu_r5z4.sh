#!/bin/bash
# both row sums of a 32-k step from ONE product (A rows: ones, OFF pattern, zeros) against two products
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do for shape in "28672 7168 256" "4096 4096 2048" "8192 8192 1024"; do
  echo "### $shape rep $rep"
  echo -n "two products: "; timeout 200 build_gpu/k2lab mb $shape 2 bf16 "mb<2,4x2,4x4,nl4" 2>&1 | grep -E "^mb<" | cut -c42-140
  echo -n "one product:  "; timeout 200 build_gpu/k2lab_new mb $shape 2 bf16 "mb<2,4x2,4x4,nl4" 2>&1 | grep -E "^mb<" | cut -c42-140
done; done
echo "### others (one product)"
timeout 200 build_gpu/k2lab_new mb 4096 4096 512 2 f16 "nl4" 2>&1 | grep -E "mb"
timeout 200 build_gpu/k2lab_new mb 4112 2048 200 2 bf16 "nl4" 2>&1 | grep -E "mb"
timeout 200 build_gpu/k2lab_new mb 4096 2048 256 4 bf16 "nl4" 2>&1 | grep -E "mb"
} > $O/r05z_k2lab_mb_onesum.txt 2>&1
cut -c1-150 $O/r05z_k2lab_mb_onesum.txt
