#!/bin/bash
# GPU box: build + run the fused-decode lab with phase stamps -> gpurun_out/fusedlab_<tag>.log
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-x}; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DFG_PROBE -DFG_PROBE_WG=${2:-5} -I include -I quip_amd/csrc scripts/fusedlab.hip -o /tmp/fusedlab 2>&1 | grep -E "error" 
# regimes: 0 rounds 3-5 form, 1 HBM-cold (every operand its own copy, > 320 MiB), 2 Infinity-Cache-warm, 3 L2-warm
: > $O/fusedlab_$TAG.log
for reg in ${3:-0 1 2 3}; do
  echo "==== regime $reg" >> $O/fusedlab_$TAG.log
  timeout 600 /tmp/fusedlab $reg >> $O/fusedlab_$TAG.log 2>&1; echo "regime $reg rc=$?"
done
cat $O/fusedlab_$TAG.log
