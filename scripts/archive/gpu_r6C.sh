#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python scripts/decode_wglog.py --bs 16 --tokens 1 2>&1 | grep -v amdgpu.ids | tee $O/r06C_decode_wglog_kron_bs16.txt | tail -30 | cut -c1-200
