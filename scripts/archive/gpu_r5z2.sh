#!/bin/bash
# lab binaries: for v in <switches>; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $(for s in $v; do echo -DK2_MB_$s; done) -I include -I quip_amd/csrc scripts/k2lab.hip -o build_gpu/k2lab_<name>; done
#   k2lab_<A>_<B> = -DK2_MB_<A> -DK2_MB_<B>
# ablations of the 16x16x32 stage loop of dq_mb_kernel (results wrong by construction, only the time matters), both headline shapes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for shape in "28672 7168 256" "4096 4096 2048"; do
  echo "### $shape"
  echo -n "full: "; timeout 200 build_gpu/k2lab mb $shape 2 bf16 "mb<2,4x2,4x4,nl4" 2>&1 | grep -E "^mb<" | cut -c42-140
  for v in NOSUMS NODEQ NOX NODMA NOSUMS_NODEQ NOSUMS_NODEQ_NOX NOSUMS_NODEQ_NOX_NODMA; do
    echo -n "$v: "; timeout 200 build_gpu/k2lab_$v mb $shape 2 bf16 "mb<2,4x2,4x4,nl4" 2>&1 | grep -E "^mb<" | cut -c42-140
  done
done
} > $O/r05z_k2lab_mb16_ablations.txt 2>&1
cat $O/r05z_k2lab_mb16_ablations.txt
