#!/bin/bash
# GPU box: A / B of headline-kernel configurations on one box, alternating runs (QUIP_K2_CFG = family,p1,p2 through quipamd_dequant_gemm_cfg)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
CFGS=${CFGS:-"2,8,2 2,16,1"}
for rep in 1 2 3; do
  for v in $CFGS; do
    QUIP_K2_CFG=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-decode --no-llama --no-ldlq --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg $v', 'K=20 cold us', d['roofline']['us_per_launch'], 'warm', d['warm']['us_per_launch'])"
  done
done
for v in $CFGS; do
  QUIP_K2_CFG=$v timeout 300 python bench.py --steps 2000 --warmup 200 --no-decode --no-llama --no-ldlq --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg $v', 'K=2000 cold us', d['roofline']['us_per_launch'], 'warm', d['warm']['us_per_launch'])"
done
