#!/bin/bash
# round 5, call g: weights-first A/B on the headline kernel, bigp tail two-launch form, tests, decode sweeps, bench headline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2 3 4; do
  for v in c0w0 c2w0 c2w1 c0w1 c3w1; do echo -n "$v rep $rep: "; timeout 120 build_gpu/k2lab_$v h 4096 4096 16 2 bf16 "h<2,rt1,nw8,nch2>" 2>&1 | grep "h<2" ; done
done
for v in c0w0 c2w0 c2w1; do echo -n "$v f16: "; timeout 120 build_gpu/k2lab_$v h 4096 4096 16 2 f16 "h<2,rt1,nw8,nch2>" 2>&1 | grep "h<2"; done
for v in c0w0 c2w0 c2w1; do echo -n "$v bs8: "; timeout 120 build_gpu/k2lab_$v h 4096 4096 8 2 bf16 "h<2,rt1,nw8,nch2>" 2>&1 | grep "h<2"; done
for v in c0w0 c2w0 c2w1; do echo -n "$v bs1: "; timeout 120 build_gpu/k2lab_$v h 4096 4096 1 2 bf16 "h<2,rt1,nw8,nch2>" 2>&1 | grep "h<2"; done
for v in c0w0 c2w0 c2w1; do echo -n "$v w4: "; timeout 120 build_gpu/k2lab_$v h 4096 4096 16 4 bf16 "h<4,rt1,nw8,nch4>" 2>&1 | grep "h<4"; done
for v in c0w0 c2w0 c2w1; do echo -n "$v 8192x2048: "; timeout 120 build_gpu/k2lab_$v h 8192 2048 16 2 bf16 "h<2,rt1,nw8,nch1>" 2>&1 | grep "h<2"; done
for v in c0w0 c2w0 c2w1; do echo -n "$v 11008x4096: "; timeout 120 build_gpu/k2lab_$v h 11008 4096 16 2 bf16 "h<2,rt1,nw8,nch2>" 2>&1 | grep "h<2"; done
} > $O/r05g_k2lab_wfirst_ab.txt 2>&1
cat $O/r05g_k2lab_wfirst_ab.txt | cut -c1-150
echo "== bench headline"; for rep in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-decode --no-llama --no-ldlq --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20 cold us', d['roofline']['us_per_launch'], 'frac', d['roofline']['frac'], 'warm', d['warm']['us_per_launch'], 'acc', d['accumulate_contract']['us_per_launch_cold'], 'fp16', d.get('k2_shapes'))"; done
timeout 300 python bench.py --steps 2000 --warmup 200 --no-decode --no-llama --no-ldlq --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=2000 cold us', d['roofline']['us_per_launch'], 'warm', d['warm']['us_per_launch'])"
echo "== bigp tail"; timeout 300 python scripts/bench_bigp_tail.py > $O/r05g_bigp_tail.jsonl 2>&1; cat $O/r05g_bigp_tail.jsonl | cut -c1-300
echo "== pytest"; timeout 1200 python -m pytest tests/test_gpu_dqgemm.py tests/test_gpu_dqgemm_v2.py tests/test_gpu_decode_bigp.py tests/test_gpu_decode_e2e.py tests/test_gpu_decode_hf.py tests/test_gpu_decode_fused.py tests/test_gpu_decode_step.py -x -q > $O/r05g_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/r05g_pytest.log
echo "== decode sweeps"; for arch in opt llama; do timeout 900 python scripts/decode_engine_bench.py --arch $arch --prompt 32 --tokens 32 --sweep 1:-1,8:-1,16:-1 > $O/r05g_decode_batch_$arch.jsonl 2> $O/r05g_decode_batch_$arch.err; python -c "
import json,sys
for l in open('$O/r05g_decode_batch_$arch.jsonl'):
    r=json.loads(l); print('$arch', r.get('bs'), r.get('engine_mode'), round(r.get('ms_per_step_median',0),3), round(r.get('tok_per_s',0)), r.get('error'))"; done
