#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest v2"; timeout 900 python -m pytest tests/test_gpu_dqgemm_v2.py tests/test_gpu_dqgemm.py -x -q > $O/r05l_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/r05l_pytest.log
{
echo "### mb w4 28672x7168 bs256"; timeout 200 build_gpu/k2lab mb 28672 7168 256 4 bf16 2>&1 | grep -E "mb<|old"
echo "### mb 2048x2048 bs64 (cfg 22/23 territory)"; timeout 200 build_gpu/k2lab mb 2048 2048 64 2 f16 2>&1 | grep -E "mb<|old"
echo "### mb 8192x2048 bs32"; timeout 200 build_gpu/k2lab mb 8192 2048 32 2 f16 2>&1 | grep -E "mb<|old"
echo "### mb 8192x8192 bs1024"; timeout 200 build_gpu/k2lab mb 8192 8192 1024 2 bf16 "4x4" 2>&1 | grep -E "mb<|old"
} > $O/r05l_k2lab_mb.txt 2>&1
cat $O/r05l_k2lab_mb.txt | cut -c1-150
echo "== decode sweep opt 32/64"; timeout 600 python scripts/decode_engine_bench.py --arch opt --prompt 32 --tokens 32 --sweep 32:-1,64:-1 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print('opt', r.get('bs'), r.get('engine_mode'), round(r.get('ms_per_step_median',0),3), round(r.get('tok_per_s',0)), r.get('error'))"
