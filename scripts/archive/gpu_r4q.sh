#!/bin/bash
# round 4, call Q: K2's weight-stream kernel with two row tiles per compute wave (x fragments read from LDS once for both)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dqgemm_v2.py -x -q -m gpu -k "s_kernel" > gpurun_out/r04q_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" gpurun_out/r04q_pytest.log | tail -5
timeout 600 python scripts/bench_k2_s_cfgs.py > gpurun_out/r04q_k2_s_cfgs.jsonl 2> gpurun_out/r04q_k2_s_cfgs.err; echo "bench rc=$?"
cut -c1-200 gpurun_out/r04q_k2_s_cfgs.jsonl; tail -3 gpurun_out/r04q_k2_s_cfgs.err
