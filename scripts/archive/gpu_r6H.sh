#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1200 python -m pytest tests/test_gpu_gptq_qfnb.py -q 2>&1 | tail -4
