#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_decode_hf.py tests/test_gpu_decode_step.py -q -x 2>&1 | tail -3
timeout 900 python scripts/decode_engine_bench.py --arch opt --sweep 16:2048,12:2048,16:2048 2>/dev/null | tee $O/r06K_decode_bs.jsonl | cut -c1-30,250-360
timeout 900 python scripts/decode_engine_bench.py --arch llama --sweep 16:2048 2>/dev/null | tee -a $O/r06K_decode_bs.jsonl | cut -c1-30,250-360
