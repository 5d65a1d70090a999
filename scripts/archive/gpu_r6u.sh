#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python scripts/decode_wglog.py 2>&1 | grep -v amdgpu.ids | tee $O/r06w_decode_wglog_kron.txt | cut -c1-200
timeout 900 python scripts/decode_wglog.py --arch llama 2>&1 | grep -v amdgpu.ids | tee $O/r06w_decode_wglog_llama.txt | tail -3 | cut -c1-300
