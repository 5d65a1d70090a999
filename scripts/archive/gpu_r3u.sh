#!/bin/bash
# round 3u: the p x 16 decode tail (csrc/decode_bigp.hip): tests, then the Llama-2-7B-architecture decode under rocprofv3
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_decode_bigp.py tests/test_gpu_decode_fused.py tests/test_gpu_decode_hf.py -x -q > $O/r3u_tests.log 2>&1
echo "tests rc=$?"; tail -15 $O/r3u_tests.log
bash scripts/prof_llama_v3.sh r03u 2>&1 | tail -30
