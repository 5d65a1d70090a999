#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dqgemm_v2.py -m gpu -q -x -p no:cacheprovider -k "prefill" > $O/pf_tests_r03t.log 2>&1; tail -15 $O/pf_tests_r03t.log | cut -c1-250
timeout 900 python scripts/bench_k2_prefill.py > $O/k2_prefill_r03t.jsonl 2> $O/k2_prefill_r03t.err; tail -3 $O/k2_prefill_r03t.err | cut -c1-300; cat $O/k2_prefill_r03t.jsonl
