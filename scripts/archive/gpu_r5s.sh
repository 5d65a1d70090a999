#!/bin/bash
# round 5: from how many rows on does the two-launch form (prologue-only launch + dequant-GEMM) beat the single fused launch?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for arch in opt llama; do
  timeout 900 python scripts/decode_engine_bench.py --arch $arch --prompt 32 --tokens 32 --sweep 1:-1,2:-1,3:-1,4:-1 --two-launch-rows 5,3,2 > $O/r05s_two_launch_$arch.jsonl 2> $O/r05s_two_launch_$arch.err
  echo "$arch rc=$?"
  python - $O/r05s_two_launch_$arch.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print(d.get('bs'), d.get('two_launch_rows'), d.get('engine_mode'), round(d.get('ms_per_step_median', 0), 4), round(d.get('tok_per_s', 0), 1), d.get('error', ''))
PY
done
