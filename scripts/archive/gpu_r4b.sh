#!/bin/bash
# round 4, call B: tests touched by the ADVICE fixes + operator prefetch; whole-model walls with and without the opt-ins; bench with the CPU reference leg
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_gptq_qfnb.py tests/test_gpu_gptq.py tests/test_gpu_refformats.py tests/test_gpu_dqgemm_v2.py tests/test_gpu_decode_e2e.py tests/test_gpu_preproc_fused.py tests/test_gpu_method.py tests/test_gpu_driver.py tests/test_gpu_decode_fused.py tests/test_gpu_decode_bigp.py tests/test_gpu_shard_rccl.py -x -q -m gpu > gpurun_out/r04b_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r04b_pytest.log
timeout 900 python scripts/run_full_model.py --model llama-2-7b --prefetch-operators --out gpurun_out/r04b_llama7b_prefetch.json > gpurun_out/r04b_llama7b_prefetch.log 2>&1
echo "llama prefetch rc=$?"; python -c "import json;d=json.load(open('gpurun_out/r04b_llama7b_prefetch.json'));print(d['wall_s'],d['phases_s'])"
timeout 900 python scripts/run_full_model.py --model opt-1.3b --out gpurun_out/r04b_opt1p3b.json > gpurun_out/r04b_opt1p3b.log 2>&1
echo "opt1.3b rc=$?"; python -c "import json;d=json.load(open('gpurun_out/r04b_opt1p3b.json'));print(d['wall_s'],d['phases_s'])"
timeout 900 python scripts/run_full_model.py --model opt-1.3b --prefetch-operators --out gpurun_out/r04b_opt1p3b_prefetch.json > gpurun_out/r04b_opt1p3b_prefetch.log 2>&1
echo "opt1.3b prefetch rc=$?"; python -c "import json;d=json.load(open('gpurun_out/r04b_opt1p3b_prefetch.json'));print(d['wall_s'],d['phases_s'])"
timeout 900 python scripts/run_full_model.py --model llama-2-7b --prefetch-operators --fast-hessian --out gpurun_out/r04b_llama7b_prefetch_fasthessian.json > gpurun_out/r04b_llama7b_pf_fh.log 2>&1
echo "llama prefetch+fast hessian rc=$?"; python -c "import json;d=json.load(open('gpurun_out/r04b_llama7b_prefetch_fasthessian.json'));print(d['wall_s'],d['phases_s'])"
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err
echo "bench rc=$?"; python -c "
import json;d=json.load(open('gpurun_out/r04b_bench.json'))
print(d['value'],d['roofline']['frac'],d['ms_per_step'])
print(json.dumps(d.get('ldlq_cpu_reference'))[:1500]); print(d.get('ldlq_cpu_port')); print(d.get('cpu_baseline'))
print(d['decode']['value'], d['decode'].get('roofline')); print(d['decode_llama']['value'], d['decode_llama'].get('roofline'))"
