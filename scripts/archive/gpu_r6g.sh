#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
TAG=r06g
echo "== pytest register-x kernel"; timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_dqgemm_v2.py -k "register_x or h_kernel or one_hot" > $O/pytest_$TAG.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_$TAG.log | cut -c1-300
for rep in 1 2 3; do
  for cfg in "2,4,4" "2,44,4"; do
    echo "== bench headline, QUIP_K2_CFG=$cfg (rep $rep)"
    QUIP_K2_CFG=$cfg timeout 600 python bench.py --steps 20 --warmup 5 --no-ldlq --no-decode --no-llama --no-cpu-baseline 2> $O/bench_${TAG}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'cfg': '$cfg', 'us_cold': d['roofline']['us_per_launch'], 'frac': d['roofline']['frac'], 'from_idle': d.get('from_idle', {}).get('us_per_launch'), 'warm': d['warm']['us_per_launch'], 'fp32_y': d.get('fp32_y', {}).get('us_per_launch'), 'acc': d['accumulate_contract']['us_per_launch_cold'], 'rel': d['parity_rel_err']}))" | tee -a $O/headline_ab_$TAG.jsonl
  done
done
for cfg in "2,4,4" "2,44,4"; do
  echo "== bench headline K = 2000, QUIP_K2_CFG=$cfg"
  QUIP_K2_CFG=$cfg timeout 600 python bench.py --steps 2000 --warmup 200 --no-ldlq --no-decode --no-llama --no-cpu-baseline 2>> $O/bench_${TAG}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'cfg': '$cfg', 'K': 2000, 'us_cold': d['roofline']['us_per_launch'], 'warm': d['warm']['us_per_launch']}))" | tee -a $O/headline_ab_$TAG.jsonl
done
echo "== stamps"; timeout 600 python scripts/k2h_stamps.py > $O/k2h_stamps_$TAG.txt 2>&1; cat $O/k2h_stamps_$TAG.txt
