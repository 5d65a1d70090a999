#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1200 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_decode_attn.py -q 2>&1 | tail -3
