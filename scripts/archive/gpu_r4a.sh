#!/bin/bash
# round 4, call A: the refactored decode engine + end-to-end test, then the reference drivers at full model size
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode_e2e.py tests/test_gpu_decode_hf.py tests/test_gpu_decode_step.py -x -q -m gpu -s > gpurun_out/r04a_pytest_decode.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04a_pytest_decode.log
tail -5 gpurun_out/r04a_pytest_decode.log
timeout 600 python scripts/run_full_model.py --model opt-125m --out gpurun_out/r04a_opt125m_w4_ldlq_reference_driver.json > gpurun_out/r04a_opt125m.log 2>&1
echo "opt125m rc=$?"; tail -c 600 gpurun_out/r04a_opt125m.log
timeout 1200 python scripts/run_full_model.py --model llama-2-7b --out gpurun_out/r04a_llama7b_w2_ldlq_incoh_reference_driver.json > gpurun_out/r04a_llama7b.log 2>&1
echo "llama7b rc=$?"; tail -c 1500 gpurun_out/r04a_llama7b.log
