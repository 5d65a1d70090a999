#!/bin/bash
# round 4, call G: whole gpu suite with shared-input Hessians, then the whole-model walls again
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04g_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r04g_pytest_gpu.log
timeout 900 python scripts/run_full_model.py --model llama-2-7b --out gpurun_out/r04g_llama7b.json > gpurun_out/r04g_llama7b.log 2>&1
echo "llama rc=$?"; python -c "import json;d=json.load(open('gpurun_out/r04g_llama7b.json'));print(d['wall_s'],d['phases_s'],d['add_batch_calls'])"
timeout 900 python scripts/run_full_model.py --model llama-2-7b --prefetch-operators --out gpurun_out/r04g_llama7b_prefetch.json > gpurun_out/r04g_llama7b_prefetch.log 2>&1
echo "llama prefetch rc=$?"; python -c "import json;d=json.load(open('gpurun_out/r04g_llama7b_prefetch.json'));print(d['wall_s'],d['phases_s'])"
timeout 900 python scripts/run_full_model.py --model opt-1.3b --prefetch-operators --out gpurun_out/r04g_opt1p3b_prefetch.json > gpurun_out/r04g_opt1p3b_prefetch.log 2>&1
echo "opt1.3b prefetch rc=$?"; python -c "import json;d=json.load(open('gpurun_out/r04g_opt1p3b_prefetch.json'));print(d['wall_s'],d['phases_s'])"
timeout 900 python scripts/run_full_model.py --model opt-1.3b --out gpurun_out/r04g_opt1p3b.json > gpurun_out/r04g_opt1p3b.log 2>&1
echo "opt1.3b rc=$?"; python -c "import json;d=json.load(open('gpurun_out/r04g_opt1p3b.json'));print(d['wall_s'],d['phases_s'])"
