#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest"; timeout 1200 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_decode_hf.py tests/test_gpu_dqgemm_v2.py -x -q > $O/r05o_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/r05o_pytest.log
for rt in 1 0; do
  for arch in llama opt; do
    if [ $rt = 1 ]; then export QUIP_HG_RT=1; else unset QUIP_HG_RT; fi
    timeout 900 python scripts/decode_engine_bench.py --arch $arch --prompt 32 --tokens 32 --sweep 8:-1,16:-1 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print('$arch', 'QUIP_HG_RT=${QUIP_HG_RT:-auto}', r.get('bs'), r.get('engine_mode'), round(r.get('ms_per_step_median',0),3), round(r.get('tok_per_s',0)), r.get('error'))"
  done
done
