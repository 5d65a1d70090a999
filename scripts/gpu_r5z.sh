#!/bin/bash
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
