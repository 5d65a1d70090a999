#!/bin/bash
# GPU box: build + run the fused-decode lab with phase stamps -> gpurun_out/fusedlab_<tag>.log
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-x}; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DFG_PROBE -DFG_PROBE_WG=${2:-5} -I include -I quip_amd/csrc scripts/fusedlab.hip -o /tmp/fusedlab 2>&1 | grep -E "error" 
timeout 300 /tmp/fusedlab > $O/fusedlab_$TAG.log 2>&1; echo "rc=$?"; cat $O/fusedlab_$TAG.log
