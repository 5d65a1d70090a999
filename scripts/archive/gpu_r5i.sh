#!/bin/bash
# round 5, call i: more wave x chunk shapes of the headline kernel, the bench headline with the new default, full gpu test suite, decode sweeps
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do echo "### rep $rep bf16 bs16"; timeout 150 build_gpu/k2lab h 4096 4096 16 2 bf16 "h<" 2>&1 | grep -E "h<"; done
echo "### w4"; timeout 150 build_gpu/k2lab h 4096 4096 16 4 bf16 "h<" 2>&1 | grep -E "h<"
echo "### w4 2048"; timeout 150 build_gpu/k2lab h 2048 2048 16 4 bf16 "h<" 2>&1 | grep -E "h<"
echo "### 11008x4096"; timeout 150 build_gpu/k2lab h 11008 4096 16 2 f16 "h<" 2>&1 | grep -E "h<"
} > $O/r05i_k2lab_shapes.txt 2>&1
cat $O/r05i_k2lab_shapes.txt | cut -c1-150
echo "== bench headline"; for rep in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-decode --no-llama --no-ldlq --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20 cold us', d['roofline']['us_per_launch'], 'frac', d['roofline']['frac'], 'warm', d['warm']['us_per_launch'], 'acc', d['accumulate_contract']['us_per_launch_cold'])"; done
timeout 300 python bench.py --steps 2000 --warmup 200 --no-decode --no-llama --no-ldlq --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=2000 cold us', d['roofline']['us_per_launch'], 'warm', d['warm']['us_per_launch'])"
echo "== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q > $O/r05i_pytest_gpu.log 2>&1; echo "rc=$?"; tail -5 $O/r05i_pytest_gpu.log
echo "== decode sweeps"; for arch in opt llama; do timeout 900 python scripts/decode_engine_bench.py --arch $arch --prompt 32 --tokens 32 --sweep 1:-1,16:-1 > $O/r05i_decode_batch_$arch.jsonl 2> $O/r05i_decode_batch_$arch.err; python -c "
import json,sys
for l in open('$O/r05i_decode_batch_$arch.jsonl'):
    r=json.loads(l); print('$arch', r.get('bs'), r.get('engine_mode'), round(r.get('ms_per_step_median',0),3), round(r.get('tok_per_s',0)), r.get('error'))"; done
