#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2 3; do for ph in 0 50 200; do
  echo -n "preheat $ph: "; timeout 300 python bench.py --steps 20 --warmup 5 --no-ldlq --no-decode --no-llama --no-cpu-baseline --preheat-ms $ph 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['us_per_launch'], d['warm']['us_per_launch'])"
done; done > $O/r05v_headline_preheat.txt 2>&1
cat $O/r05v_headline_preheat.txt
