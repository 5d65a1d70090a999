#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O /tmp/b; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/b/pdl_lab scripts/pdl_lab.hip > /tmp/b/cc.log 2>&1; echo "cc rc=$?"; tail -3 /tmp/b/cc.log
timeout 120 /tmp/b/pdl_lab 2>&1 | tee $O/r06z_pdl_lab.txt
