#!/bin/bash
# round 6, call A: row tiles per workgroup of the grouped one-pass GEMM (q / k / v) in the blocked decode step: QUIP_HG_RT 1 / 2 / default
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2; do
for v in unset 1 2; do
  if [ "$v" = "unset" ]; then env -u QUIP_HG_RT timeout 600 python scripts/decode_engine_bench.py --arch opt --blocked 2>/dev/null | tail -1 | sed "s/^{/{\"QUIP_HG_RT\": \"$v\", /" | tee -a $O/r06A_hg_rt_blocked.jsonl | cut -c1-40,300-400
  else QUIP_HG_RT=$v timeout 600 python scripts/decode_engine_bench.py --arch opt --blocked 2>/dev/null | tail -1 | sed "s/^{/{\"QUIP_HG_RT\": \"$v\", /" | tee -a $O/r06A_hg_rt_blocked.jsonl | cut -c1-40,300-400; fi
done
done
