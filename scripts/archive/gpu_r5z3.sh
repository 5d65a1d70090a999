#!/bin/bash
# lab binaries: for v in <switches>; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $(for s in $v; do echo -DK2_MB_$s; done) -I include -I quip_amd/csrc scripts/k2lab.hip -o build_gpu/k2lab_<name>; done
#   k2lab_UNIFORM = -DK2_MB_UNIFORM
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do for shape in "28672 7168 256" "4096 4096 2048"; do
  echo "### $shape rep $rep"
  echo -n "multi-exponent: "; timeout 200 build_gpu/k2lab mb $shape 2 bf16 "mb<2,4x2,4x4,nl4" 2>&1 | grep -E "^mb<" | cut -c42-140
  echo -n "uniform:        "; timeout 200 build_gpu/k2lab_UNIFORM mb $shape 2 bf16 "mb<2,4x2,4x4,nl4" 2>&1 | grep -E "^mb<" | cut -c42-140
done; done
} > $O/r05z_k2lab_mb_uniform.txt 2>&1
cat $O/r05z_k2lab_mb_uniform.txt
