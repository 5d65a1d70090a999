#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for s in "28672 7168 256" "4096 4096 2048"; do timeout 300 python scripts/bench_k2_mb_modes.py $s; done > $O/r05u_k2_mb_modes.jsonl 2>&1
cat $O/r05u_k2_mb_modes.jsonl
