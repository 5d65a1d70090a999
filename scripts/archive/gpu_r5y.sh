#!/bin/bash
# lab binaries: for v in <switches>; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $(for s in $v; do echo -DK2_MB_$s; done) -I include -I quip_amd/csrc scripts/k2lab.hip -o build_gpu/k2lab_<name>; done
#   k2lab_nosums = NOSUMS (the switches were named K2_T32_* when this ran; they are K2_MB_* in dqgemm_v2.h now and act on both forms of the kernel)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for shape in "28672 7168 256" "4096 4096 2048"; do
  echo "### mb $shape bf16"; timeout 200 build_gpu/k2lab mb $shape 2 bf16 "4x4,nl4" 2>&1 | grep -E "mb"
  echo "### nosums"; timeout 200 build_gpu/k2lab_nosums mb $shape 2 bf16 "mb32<2,4x2,4x4,nl4" 2>&1 | grep -E "mb"
done
} > $O/r05y_k2lab_mb32_nosums.txt 2>&1
cut -c1-170 $O/r05y_k2lab_mb32_nosums.txt
