#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python scripts/bench_k2_s_cfgs.py --shapes 28672x7168,16384x8192 2>&1 | grep -v amdgpu.ids | tee $O/r06G_k2_s_cfgs.jsonl | cut -c1-260
