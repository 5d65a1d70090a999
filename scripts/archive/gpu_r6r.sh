#!/bin/bash
# round 6, call r: the pipelined 64-row chain of csrc/gptq_qfnb.hip against the barrier-per-phase form (rows 16 / 32 / 64 / 128 forced)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
echo "== pytest gptq"; timeout 1500 python -m pytest tests/test_gpu_gptq_qfnb.py tests/test_gpu_gptq.py -q -x 2>&1 | tail -8
echo "== rows A/B (R0 = default = pipelined)"; timeout 1200 python scripts/bench_gptq_qfnb_rows.py 2>&1 | tee $O/r06r_gptq_qfnb_rows.jsonl | cut -c1-600
