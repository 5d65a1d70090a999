#!/bin/bash
# GPU box: A / B of the headline kernel with / without the one-round-trip kernarg fetch, same box, alternating runs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2 3; do
  for v in touch notouch; do
    L=$R/quip_amd/csrc/libquip_amd.so; [ $v = notouch ] && L=$R/scripts/dbg/libquip_amd_notouch.so
    QUIP_AMD_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-decode --no-llama --no-ldlq --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'K=20 cold us', d['roofline']['us_per_launch'], 'warm', d['warm']['us_per_launch'])"
  done
done
for v in touch notouch; do
  L=$R/quip_amd/csrc/libquip_amd.so; [ $v = notouch ] && L=$R/scripts/dbg/libquip_amd_notouch.so
  QUIP_AMD_LIB=$L timeout 300 python bench.py --steps 2000 --warmup 200 --no-decode --no-llama --no-ldlq --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'K=2000 cold us', d['roofline']['us_per_launch'], 'warm', d['warm']['us_per_launch'])"
done
