#!/bin/bash
# round 5, call e: rotation A/B on the headline kernel (lab builds, alternating), stamps, more ingest patterns, bigp tail by rows, dqgemm tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2 3; do
  for v in norot rot; do echo "### $v rep $rep"; timeout 120 build_gpu/k2lab_$v h 4096 4096 16 2 bf16 "h<2,rt1,nw8,nch2>" 2>&1 | grep -v amdgpu.ids; done
done
for v in norot rot; do echo "### $v f16"; timeout 120 build_gpu/k2lab_$v h 4096 4096 16 2 f16 "h<2,rt1,nw8,nch2>" 2>&1 | grep -v amdgpu.ids; done
for v in norot rot; do echo "### $v 2048 (nch1)"; timeout 120 build_gpu/k2lab_$v h 2048 2048 16 2 bf16 2>&1 | grep -v amdgpu.ids; done
for v in norot rot; do echo "### $v 11008x4096"; timeout 120 build_gpu/k2lab_$v h 11008 4096 16 2 bf16 "h<2,rt1,nw8,nch2>" 2>&1 | grep -v amdgpu.ids; done
for v in norot rot; do echo "### $v w4"; timeout 120 build_gpu/k2lab_$v h 4096 4096 16 4 bf16 2>&1 | grep -v amdgpu.ids; done
for v in norot rot; do echo "### probe $v"; timeout 120 build_gpu/k2lab_probe_$v probe_h 4096 4096 16 2 bf16 2>&1 | grep -v amdgpu.ids | head -12; done
} > $O/r05e_k2lab_rotation_ab.txt 2>&1
cat $O/r05e_k2lab_rotation_ab.txt | cut -c1-170
echo "== xingest lab v2"; timeout 200 build_gpu/xingest_lab 2>&1 | grep -v amdgpu.ids > $O/r05e_xingest_lab.txt; tail -29 $O/r05e_xingest_lab.txt
echo "== bigp tail"; timeout 300 python scripts/bench_bigp_tail.py > $O/r05e_bigp_tail.jsonl 2>&1; cat $O/r05e_bigp_tail.jsonl | cut -c1-300
echo "== pytest dqgemm"; timeout 900 python -m pytest tests/test_gpu_dqgemm.py tests/test_gpu_dqgemm_v2.py tests/test_gpu_checkpoint.py -x -q > $O/r05e_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/r05e_pytest.log
echo "== bench headline"; for rep in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-decode --no-llama --no-ldlq --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20 cold us', d['roofline']['us_per_launch'], 'frac', d['roofline']['frac'], 'warm', d['warm']['us_per_launch'], 'acc', d['accumulate_contract']['us_per_launch_cold'])"; done
timeout 300 python bench.py --steps 2000 --warmup 200 --no-decode --no-llama --no-ldlq --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=2000 cold us', d['roofline']['us_per_launch'], 'warm', d['warm']['us_per_launch'])"
