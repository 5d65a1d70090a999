#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python scripts/debug_gptq_forms.py 2>&1 | grep -v amdgpu.ids
