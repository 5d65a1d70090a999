#!/bin/bash
# round 5: dq_mb_kernel on v_mfma_f32_32x32x16 (T32) against the 16x16x32 form, same box, alternating
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do
for shape in "28672 7168 256" "4096 4096 2048" "8192 8192 1024"; do
  echo "### mb $shape bf16 rep $rep"; timeout 200 build_gpu/k2lab mb $shape 2 bf16 "4x4,nl4" 2>&1 | grep -E "mb"
done
done
echo "### f16, ragged"; timeout 200 build_gpu/k2lab mb 4096 4096 512 2 f16 "4x4,nl4" 2>&1 | grep -E "mb"
timeout 200 build_gpu/k2lab mb 4112 2048 200 2 bf16 "4x4,nl4" 2>&1 | grep -E "mb"
timeout 200 build_gpu/k2lab mb 4096 2048 256 4 bf16 "2x4,nl4" 2>&1 | grep -E "mb"
} > $O/r05x_k2lab_mb32.txt 2>&1
cut -c1-170 $O/r05x_k2lab_mb32.txt
