#!/bin/bash
# round 4, call R: K2's weight-stream kernel, reads of a ring step one barrier ahead of its use (the 10-wave configuration)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dqgemm_v2.py -x -q -m gpu > gpurun_out/r04r_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" gpurun_out/r04r_pytest.log | tail -5
timeout 600 python scripts/bench_k2_s_cfgs.py --shapes 16384x8192,28672x7168,16384x4096 > gpurun_out/r04r_k2_s_cfgs.jsonl 2> gpurun_out/r04r_k2_s_cfgs.err; echo "bench rc=$?"
cut -c1-200 gpurun_out/r04r_k2_s_cfgs.jsonl; tail -3 gpurun_out/r04r_k2_s_cfgs.err
