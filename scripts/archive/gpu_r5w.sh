#!/bin/bash
# round 5: 8 row tiles per workgroup in the grouped h kernel (Llama gate / up at 5..16 rows: 2 x 688 tiles)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_dqgemm_v2.py -q -x -k "fused_stage or grouped" > $O/r05w_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r05w_pytest.log
for rt in 4 8 4 8; do
  echo -n "QUIP_HG_RT=$rt: "; QUIP_HG_RT=$rt timeout 600 python scripts/decode_engine_bench.py --arch llama --prompt 32 --tokens 32 --sweep 8:-1,16:-1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['bs'], round(d['ms_per_step_median'], 4), round(d['tok_per_s'], 1), end='   ')
print()"
done > $O/r05w_hg_rt8_llama.txt 2>&1
cat $O/r05w_hg_rt8_llama.txt
