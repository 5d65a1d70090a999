#!/bin/bash
# lab binaries: for v in <switches>; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $(for s in $v; do echo -DK2_MB_$s; done) -I include -I quip_amd/csrc scripts/k2lab.hip -o build_gpu/k2lab_<name>; done
#   every binary here = NOSUMS + the switches in its name (K2_T32_* when this ran, K2_MB_* now)
# ablations of the T32 stage loop (all without the row sums; results are wrong by construction, only the time matters)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for shape in "28672 7168 256"; do
  for v in nosums NODEQ NOX NODEQ_NOX NODMA NODMA_NODEQ_NOX; do
    echo -n "$v: "; timeout 200 build_gpu/k2lab_$v mb $shape 2 bf16 "mb32<2,4x2,4x4,nl4" 2>&1 | grep -E "mb32" | cut -c42-140
  done
done
} > $O/r05z_k2lab_mb32_ablations.txt 2>&1
cat $O/r05z_k2lab_mb32_ablations.txt
