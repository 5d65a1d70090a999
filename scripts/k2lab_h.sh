#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out; mkdir -p $O
L=${GRAFT_REPO_ROOT:-.}/build_gpu/k2lab
run() { echo "### $*"; timeout 150 $L "$@" 2>&1 | grep -v amdgpu.ids; echo "rc=$?"; }
{
run h 4096 4096 16 2 bf16
run h 4096 4096 16 2 f16
run h 4096 4096 8 2 bf16
run h 4096 4096 1 2 bf16
run h 2048 2048 1 2 bf16
run h 8192 2048 16 2 bf16
run h 4096 4096 16 4 bf16
} > $O/k2lab_$1.log 2>&1
tail -5 $O/k2lab_$1.log
