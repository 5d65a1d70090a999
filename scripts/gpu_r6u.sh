#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python scripts/decode_wglog.py --blocked 2>&1 | grep -v amdgpu.ids | tee $O/r06u_decode_wglog_blocked.txt | cut -c1-200
